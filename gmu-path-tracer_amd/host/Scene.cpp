#include "Scene.hpp"
#include <stdexcept>
#include <thread>

namespace {
void check(int rc) { if (rc != GMUPT_OK) throw std::runtime_error(gmupt_last_error()); }
}

Scene::Scene(gmupt_device* device, const std::string& path)
	: mDevice(device)
	, mSceneName(path)
{
	// Scene.cpp:85: the scene's name is its path below the models directory
	SceneParams& registry = SceneParams::instance;
	const bool belowRoot = !registry.modelsRoot.empty() && path.compare(0, registry.modelsRoot.size(), registry.modelsRoot) == 0;
	if (belowRoot) mSceneName = path.substr(registry.modelsRoot.size());

	loadScene(path);

	std::exception_ptr bvhError;                 // as Source/Scene.cpp:89: BVH build + upload on a worker thread
	std::thread worker([&]() { try { createBVH(); } catch (...) { bvhError = std::current_exception(); } });
	try
	{
		loadTextures(); // decodes / resizes / uploads the three texture arrays on three threads, then uploads the material table

		SceneParams::Entry params;
		if (belowRoot)
		{
			const size_t index = registry.getSceneIndex(mSceneName);   // Scene.cpp:95,315; throws "Non existing scene" like the reference
			params.camera = registry.cameraParams[index]; params.lights = registry.lights[index];
		}
		else
		{
			std::string paramsPath = path;
			const auto dot = paramsPath.find_last_of('.');
			paramsPath = (dot == std::string::npos ? paramsPath : paramsPath.substr(0, dot)) + ".params";
			params = SceneParams::load(paramsPath);
		}
		createLights(params.lights);

		mCamera.setPosition(params.camera.position[0], params.camera.position[1], params.camera.position[2]);
		mCamera.setRotation(params.camera.pitch, params.camera.yaw);
	}
	catch (...) { worker.join(); throw; }

	worker.join();
	if (bvhError) std::rethrow_exception(bvhError);
}

void Scene::update(float dt)
{
	mCamera.update(dt);
}

void Scene::loadScene(const std::string& path)
{
	const bool gltf = (path.size() > 5 && path.compare(path.size() - 5, 5, ".gltf") == 0) || (path.size() > 4 && path.compare(path.size() - 4, 4, ".glb") == 0);
	mScene = (path == "cornell") ? MeshData::cornell() : gltf ? MeshData::loadGltf(path) : MeshData::load(path);
	if (mScene.materials.size() > MAX_LIGHTS) throw std::runtime_error("More than 128 materials (logic.hlsl:8)");
}

void Scene::loadTextures()
{
	// Source/Scene.cpp:126-166: one worker per texture type fills its textureIndices column and creates its Texture2DArray
	std::vector<MaterialProperty>& materialProperties = mScene.materials;
	std::string errors[3];
	auto work = [&](int index, Buffer& resource)
	{
		try
		{
			if (mScene.textureFiles[index].empty()) return;
			createTextures(gmupt::loadSpecificTexture(mScene.textureFiles[index], materialProperties, index), resource);
		}
		catch (const std::exception& e) { errors[index] = e.what(); }
	};
	std::thread diffWorker(work, 0, std::ref(mDiffuse));
	std::thread mtrWorker(work, 1, std::ref(mMetallicRoughness));
	std::thread normWorker(work, 2, std::ref(mNormal));
	diffWorker.join();
	mtrWorker.join();
	normWorker.join();
	for (const std::string& e : errors) if (!e.empty()) throw std::runtime_error(e);
	createPropertyBuffer(materialProperties);
}

void Scene::createTextures(gmupt::TextureSet set, Buffer& resource)
{
	if (set.layers.empty()) return; // Scene.cpp:250-251
	std::vector<uint8_t> packed;
	packed.reserve(set.layers.size() * set.layers[0].size());
	for (const auto& layer : set.layers) packed.insert(packed.end(), layer.begin(), layer.end());
	gmupt_buffer* b = nullptr;
	check(gmupt_texture_array_create(mDevice, packed.data(), set.dimension, static_cast<uint32_t>(set.layers.size()), &b));
	resource.reset(b);
}

void Scene::createBVH()
{
	BVHWrapper bvh(mScene);
	gmupt_buffer* b = nullptr;
	check(gmupt_buffer_create(mDevice, GMUPT_BUFFER_BVH_NODES, bvh.mGPUTree.data(), bvh.mGPUTree.size() * sizeof(BVHWrapper::BVHNode), &b)); mBVHBuffer.reset(b);
	check(gmupt_buffer_create(mDevice, GMUPT_BUFFER_TRIANGLES, bvh.mIndices.data(), bvh.mIndices.size() * sizeof(BVHWrapper::Triangle), &b)); mIndexBuffer.reset(b);
	check(gmupt_buffer_create(mDevice, GMUPT_BUFFER_VERTICES, bvh.mVertices.data(), bvh.mVertices.size() * sizeof(float), &b)); mVertexBuffer.reset(b);
	check(gmupt_buffer_create(mDevice, GMUPT_BUFFER_TRI_PROPS, bvh.mTriangleProperties.data(), bvh.mTriangleProperties.size() * sizeof(BVHWrapper::TriangleProperties), &b)); mTriangleProperties.reset(b);
}

void Scene::createPropertyBuffer(const std::vector<MaterialProperty>& data)
{
	gmupt_buffer* b = nullptr;
	check(gmupt_buffer_create(mDevice, GMUPT_BUFFER_MATERIALS, data.data(), data.size() * sizeof(MaterialProperty), &b));
	mMaterialPropertyBuffer.reset(b);
}

void Scene::createLights(const std::vector<Light>& lights)
{
	if (lights.size() > MAX_LIGHTS) throw std::runtime_error("More than 128 lights");
	mLights = {};
	for (size_t i = 0; i < lights.size(); ++i) mLights[i] = lights[i];
	mLightCount = lights.size();
	gmupt_buffer* b = nullptr;
	check(gmupt_buffer_create(mDevice, GMUPT_BUFFER_LIGHTS, mLights.data(), mLights.size() * sizeof(Light), &b));
	mLightBuffer.reset(b);
}

void Scene::setLights(const std::vector<Light>& lights)
{
	if (lights.size() > MAX_LIGHTS) throw std::runtime_error("More than 128 lights");
	mLights = {};
	for (size_t i = 0; i < lights.size(); ++i) mLights[i] = lights[i];
	mLightCount = lights.size();
	check(gmupt_buffer_update(mLightBuffer.get(), mLights.data(), mLights.size() * sizeof(Light)));
	mCamera.getBuffer()->lightCount = static_cast<uint32_t>(lights.size());
	mCamera.getBuffer()->iterationCounter = -1; // GUI.cpp: light edits reset the accumulation
}
