// SceneParams: the reference's registry of scenes and the per-scene .params CSV (Include/Scene.hpp:21-41, Source/Scene.cpp:22-80).
// Host-only (no device): kept apart from Scene.cpp so that tools can use it without the C-ABI library.
#include "Scene.hpp"
#include "CsvParser.hpp"
#include <algorithm>
#include <filesystem>
#include <fstream>
#include <stdexcept>

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

