// Host camera.  Behaviour follows the reference's Source/Camera.cpp:13-99 (cited per statement);
// DirectXMath vector ops are written out as scalar fp32 arithmetic.
#include "Camera.hpp"
#include "Constants.hpp"
#include <cmath>
#include <cstring>

namespace {
inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const float* a, const float* b, float* o)
{
	o[0] = a[1] * b[2] - a[2] * b[1];
	o[1] = a[2] * b[0] - a[0] * b[2];
	o[2] = a[0] * b[1] - a[1] * b[0];
}
inline void normalize3(float* v)
{
	const float len = std::sqrt(dot3(v, v));
	v[0] /= len; v[1] /= len; v[2] /= len;
}
inline float radians(float deg) { return deg * (3.141592654f / 180.0f); } // XMConvertToRadians
}

Camera::Camera() : Camera(WIDTH, HEIGHT) {}

Camera::Camera(size_t width, size_t height)
{
	std::memset(&mCBuffer, 0, sizeof(mCBuffer));
	mCBuffer.position[0] = 1.f; mCBuffer.position[1] = 3.f; mCBuffer.position[2] = 8.0f; // Camera.hpp:10
	mCBuffer.envColor[0] = 0.0f; mCBuffer.envColor[1] = 0.0001f; mCBuffer.envColor[2] = 0.0001f; // Camera.hpp:18
	mCBuffer.iterationCounter = -1;
	mCBuffer.lightCount = 2;      // Camera.hpp:20 (never set from the .params file)
	mCBuffer.sampleLights = 0;
	updateResolution(width, height);
}

void Camera::updateResolution(size_t width, size_t height)
{
	const auto theta = 60 * 3.14f / 180; // Camera.cpp:15 -- 3.14, not pi
	const auto aspect = width / static_cast<float>(height);

	mHalfHeight = std::tan(theta / 2.f);
	mHalfWidth = aspect * mHalfHeight;

	mCBuffer.pixelSize[0] = 1.f / width;
	mCBuffer.pixelSize[1] = 1.f / height;
	mCBuffer.iterationCounter = -1;
}

void Camera::update(float dt)
{
	// mouse (Camera.cpp:28-33)
	const float dx = mDeltaX, dy = mDeltaY;
	mDeltaX = mDeltaY = 0.f;
	mYaw += dx;
	mPitch -= dy;
	if (mPitch > 89.0f) mPitch = 89.0f;
	if (mPitch < -89.0f) mPitch = -89.0f;

	// Camera.cpp:35-43
	mFront[0] = std::cos(radians(mYaw)) * std::cos(radians(mPitch));
	mFront[1] = std::sin(radians(mPitch));
	mFront[2] = std::sin(radians(mYaw)) * std::cos(radians(mPitch));
	normalize3(mFront);
	const float yAxis[3] = { 0.f, 1.f, 0.f };
	cross3(yAxis, mFront, mLeft); normalize3(mLeft);
	cross3(mFront, mLeft, mUp); normalize3(mUp);

	// keyboard (Camera.cpp:47-57)
	constexpr auto speed = 5.f;
	const auto velocity = speed * dt;
	for (int i = 0; i < 3; i++) {
		if (mKeyW) mCBuffer.position[i] += mFront[i] * velocity;
		if (mKeyS) mCBuffer.position[i] -= mFront[i] * velocity;
		if (mKeyA) mCBuffer.position[i] += mLeft[i] * velocity;
		if (mKeyD) mCBuffer.position[i] -= mLeft[i] * velocity;
	}

	// Camera.cpp:61-65: rows of transpose(XMMatrixLookAtRH(pos, front + pos, up)) are (axis, -dot(axis, pos))
	float focus[3], zAxis[3], xAxis[3], yAx[3], negEye[3];
	for (int i = 0; i < 3; i++) focus[i] = mFront[i] + mCBuffer.position[i];
	for (int i = 0; i < 3; i++) zAxis[i] = mCBuffer.position[i] - focus[i];
	normalize3(zAxis);
	cross3(mUp, zAxis, xAxis); normalize3(xAxis);
	cross3(zAxis, xAxis, yAx);
	for (int i = 0; i < 3; i++) negEye[i] = -mCBuffer.position[i];
	const float left[4] = { xAxis[0], xAxis[1], xAxis[2], dot3(xAxis, negEye) };
	const float up[4] = { yAx[0], yAx[1], yAx[2], dot3(yAx, negEye) };
	const float w[4] = { zAxis[0], zAxis[1], zAxis[2], dot3(zAxis, negEye) };

	for (int i = 0; i < 4; i++) {
		mCBuffer.upperLeftCorner[i] = -mHalfWidth * left[i] + mHalfHeight * up[i] - w[i]; // Camera.cpp:67
		mCBuffer.horizontal[i] = 2 * mHalfWidth * left[i];                               // :68
		mCBuffer.vertical[i] = 2 * mHalfHeight * up[i];                                  // :69
	}
	mCBuffer.iterationCounter++; // :70

	// accumulation reset hysteresis (Camera.cpp:72-83)
	const bool anyActive = mKeyW || mKeyS || mKeyA || mKeyD;
	if (dx != 0.f || dy != 0.f || anyActive)
	{
		if (mCBuffer.iterationCounter > 4)
			mCBuffer.iterationCounter = 0;
		moveHysteresis = true;
	}
	else if (moveHysteresis && mCBuffer.iterationCounter > 4)
	{
		mCBuffer.iterationCounter = 0;
		moveHysteresis = false;
	}

	// Camera.cpp:85-87
	const float a = nextRand() / static_cast<float>(32767);
	const float b = nextRand() / static_cast<float>(32767);
	mCBuffer.randomSeed[0] = a;
	mCBuffer.randomSeed[1] = b;
}

void Camera::setRotation(float pitch, float yaw)
{
	mPitch = pitch;
	mYaw = yaw;
}

void Camera::setPosition(float x, float y, float z)
{
	mCBuffer.position[0] = x; mCBuffer.position[1] = y; mCBuffer.position[2] = z; mCBuffer.position[3] = 0.f;
}

Camera::CameraBuffer* Camera::getBuffer()
{
	return &mCBuffer;
}

int Camera::nextRand()
{
	// MSVC CRT rand(): 15-bit LCG, RAND_MAX = 32767
	mRandState = mRandState * 214013u + 2531011u;
	return static_cast<int>((mRandState >> 16) & 0x7FFFu);
}
