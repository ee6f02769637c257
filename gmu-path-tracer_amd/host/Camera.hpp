// Host camera: same public surface as the reference's Camera (Include/Camera.hpp:3-46) without DirectXMath.
#pragma once
#include <cstddef>
#include <cstdint>
#include "../../include/gmupt.h"

class Camera
{
public:
	// field order / size of the reference's Camera::CameraBuffer (Include/Camera.hpp:8-22), 112 bytes
	using CameraBuffer = gmupt_camera_buffer;

	Camera();
	Camera(size_t width, size_t height);

	void updateResolution(size_t width, size_t height);
	void update(float dt);
	void setRotation(float pitch, float yaw);
	void setPosition(float x, float y, float z);
	CameraBuffer* getBuffer();

	// input hooks (the reference reads a Win32 Input singleton, Source/Camera.cpp:28,50-57,72)
	void addMouseDelta(float dx, float dy) { mDeltaX += dx; mDeltaY += dy; }
	void setKeys(bool w, bool s, bool a, bool d) { mKeyW = w; mKeyS = s; mKeyA = a; mKeyD = d; }
	// MSVC rand() stream used for CameraBuffer::randomSeed (Source/Camera.cpp:85-87); the reference never calls srand
	void seedRandom(uint32_t state) { mRandState = state; }

private:
	int nextRand();

	CameraBuffer mCBuffer;
	float mFront[3] = { 0.f, 0.f, 1.f };
	float mUp[3] = { 0.f, 1.f, 0.f };
	float mLeft[3] = { 0.f, 0.f, 0.f };
	float mHalfWidth = 0.f;
	float mHalfHeight = 0.f;
	float mPitch = 0.f;
	float mYaw = 270.f;
	bool moveHysteresis = false;
	float mDeltaX = 0.f, mDeltaY = 0.f;
	bool mKeyW = false, mKeyS = false, mKeyA = false, mKeyD = false;
	uint32_t mRandState = 1;
};
