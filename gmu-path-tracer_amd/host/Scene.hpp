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

// The reference's registry of scenes (Include/Scene.hpp:21-41, Source/Scene.cpp:22-80): every *.gltf below the models directory with the
// camera pose and lights of its sibling .params file (or the defaults).  Differences: the models directory is a settable path (the
// reference hard-codes "Assets\\Models\\" relative to the working directory), names use the platform's separator, .glb files are listed
// too, and the registry is filled on first use instead of during static initialisation.
struct SceneParams
{
	struct CameraParam { float position[3]; float pitch; float yaw; };
	struct Entry { std::vector<Light> lights; CameraParam camera; };

	void loadScenes();                                   // scans modelsRoot recursively (Scene.cpp:22-73); replaces the current lists
	void loadScenes(const std::string& root);            // the same after modelsRoot = root
	size_t getSceneIndex(const std::string& name);       // name relative to modelsRoot; throws std::runtime_error("Non existing scene <name>") (Scene.cpp:75-80)
	bool contains(const std::string& name);

	std::vector<std::string> pathNames;                  // scene names relative to modelsRoot, in directory-walk order
	std::vector<const char*> pathsReference;             // the same as C strings (GUI.cpp:72 hands them to its combo box)
	std::vector<std::vector<Light>> lights;
	std::vector<CameraParam> cameraParams;
	std::string modelsRoot = "Assets/Models/";

	static SceneParams instance;

	// row 0 = camera (x,y,z,pitch,yaw), rows 1.. = lights (x,y,z,falloff,r,g,b,radius); defaults if the file is missing
	// (Source/Scene.cpp:34-62)
	static Entry load(const std::string& paramsPath);

private:
	bool mLoaded = false;
};

struct BufferDeleter { void operator()(gmupt_buffer* b) const { gmupt_buffer_destroy(b); } };
using Buffer = std::unique_ptr<gmupt_buffer, BufferDeleter>;

class Scene
{
public:
	Scene() = default;
	// path: a .gltf / .glb file.  Below SceneParams::instance.modelsRoot its camera / lights come from the registry, as in the reference
	// (Scene.cpp:95,315); elsewhere (and for a ".gmesh" dump or the built-in "cornell") from the sibling "<name>.params" or the defaults.
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
