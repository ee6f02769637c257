#include "Scene.hpp"
#include "CsvParser.hpp"
#include <algorithm>
#include <filesystem>
#include <fstream>
#include <stdexcept>
#include <thread>

namespace {
void check(int rc) { if (rc != GMUPT_OK) throw std::runtime_error(gmupt_last_error()); }
}

SceneParams SceneParams::instance = SceneParams();

void SceneParams::loadScenes(const std::string& root)
{
	modelsRoot = root;
	if (!modelsRoot.empty() && modelsRoot.back() != '/' && modelsRoot.back() != '\\') modelsRoot += '/';
	loadScenes();
}

void SceneParams::loadScenes()
{
	namespace fs = std::filesystem;
	pathNames.clear(); pathsReference.clear(); lights.clear(); cameraParams.clear();
	mLoaded = true;
	std::error_code ec;
	if (fs::is_directory(modelsRoot, ec)) {
		std::vector<std::string> found;
		for (const auto& f : fs::recursive_directory_iterator(modelsRoot, ec)) {
			if (!f.is_regular_file()) continue;
			const std::string ext = f.path().extension().string();
			if (ext == ".gltf" || ext == ".glb") found.push_back(f.path().string());
		}
		std::sort(found.begin(), found.end());   // the reference lists in directory-walk order (unspecified); sorted here so that indices are stable
		for (const std::string& full : found) {
			pathNames.emplace_back(full.substr(modelsRoot.size()));                       // Scene.cpp:30: name relative to the models directory
			const Entry e = load(full.substr(0, full.find_last_of('.')) + ".params");     // :34-62
			lights.push_back(e.lights);
			cameraParams.push_back(e.camera);
		}
	}
	for (const auto& p : pathNames) pathsReference.emplace_back(p.c_str());              // :67-68
}

bool SceneParams::contains(const std::string& name)
{
	if (!mLoaded) loadScenes();
	return std::find(pathNames.begin(), pathNames.end(), name) != pathNames.end();
}

size_t SceneParams::getSceneIndex(const std::string& name)
{
	if (!mLoaded) loadScenes();
	for (size_t i = 0; i < pathNames.size(); ++i)
		if (name == pathNames[i]) return i;
	throw std::runtime_error("Non existing scene " + name);                              // Scene.cpp:79
}

SceneParams::Entry SceneParams::load(const std::string& paramsPath)
{
	Entry e;
	std::ifstream file(paramsPath);
	if (file.is_open())
	{
		CSVRow row;
		if (!row.readNextRow(file) || row.size() < 5) throw std::runtime_error("Malformed params file " + paramsPath);
		for (size_t i = 0; i < 3; i++) e.camera.position[i] = std::stof(row[i]);
		e.camera.pitch = std::stof(row[3]);
		e.camera.yaw = std::stof(row[4]);
		while (row.readNextRow(file))
		{
			if (row.size() < 8) throw std::runtime_error("Malformed light row in " + paramsPath);
			Light light;
			for (size_t i = 0; i < sizeof(Light) / 4; i++) reinterpret_cast<float*>(&light)[i] = std::stof(row[i]);
			e.lights.push_back(light);
		}
	}
	else
	{
		// default params (Source/Scene.cpp:59-61)
		e.camera = { {1.0f, 3.0f, 8.0f}, 0.f, 270.f };
		e.lights.push_back(Light{ {13.0f, 4.5f, 4.5f}, 100.0f, {80.0f, 80.0f, 40.0f}, 0.5f });
		e.lights.push_back(Light{ {0.0f, 4.5f, 2.0f}, 100.0f, {80.0f, 80.0f, 40.0f}, 0.5f });
	}
	return e;
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
