// Same public surface as the reference's Scene / SceneParams / Light / MaterialProperty (Include/Scene.hpp:13-120), with the
// ID3D11Device replaced by the C-ABI device handle and uni::Buffer members by gmupt_buffer handles (move-only RAII).
#pragma once
#include <array>
#include <memory>
#include <string>
#include <vector>
#include "Camera.hpp"
#include "BVHWrapper.hpp"
#include "Constants.hpp"
#include "TextureLoader.hpp"

using Light = gmupt_light;                   // Include/Scene.hpp:13-19
using MaterialProperty = gmupt_material;     // Include/Scene.hpp:43-68

struct SceneParams
{
	struct CameraParam { float position[3]; float pitch; float yaw; };
	struct Entry { std::vector<Light> lights; CameraParam camera; };
	// row 0 = camera (x,y,z,pitch,yaw), rows 1.. = lights (x,y,z,falloff,r,g,b,radius); defaults if the file is missing
	// (Source/Scene.cpp:34-62)
	static Entry load(const std::string& paramsPath);
};

struct BufferDeleter { void operator()(gmupt_buffer* b) const { gmupt_buffer_destroy(b); } };
using Buffer = std::unique_ptr<gmupt_buffer, BufferDeleter>;

class Scene
{
public:
	Scene() = default;
	// path: "<dir>/<name>.gmesh" with an optional sibling "<name>.params", or the built-in "cornell"
	Scene(gmupt_device* device, const std::string& path);

	Scene(Scene&) = delete;
	Scene& operator=(const Scene&) = delete;
	Scene(Scene&& scene) = default;
	Scene& operator=(Scene&& scene) = default;

	void update(float dt);
	size_t lightCount() const { return mLightCount; }
	void setLights(const std::vector<Light>& lights); // GUI light editor: re-upload (Source/GUI.cpp:125-130)

private:
	void loadScene(const std::string& path);
	void createBVH();
	void loadTextures();
	void createTextures(struct gmupt::TextureSet set, Buffer& resource);
	void createPropertyBuffer(const std::vector<MaterialProperty>& data);
	void createLights(const std::vector<Light>& lights);

	gmupt_device* mDevice = nullptr;
	Camera mCamera;
	MeshData mScene;
	std::string mSceneName;

	Buffer mBVHBuffer;
	Buffer mIndexBuffer;
	Buffer mVertexBuffer;
	Buffer mTriangleProperties;
	Buffer mLightBuffer;
	Buffer mMaterialPropertyBuffer;
	Buffer mDiffuse, mMetallicRoughness, mNormal; // Texture2DArray t5..t7 (Include/Scene.hpp:108-110); null when the scene has none

	std::array<Light, MAX_LIGHTS> mLights{};
	size_t mLightCount = 0;

	friend class Renderer;
};
