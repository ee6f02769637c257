// Headless driver of the C++ host classes: the reference's Window::loop (Source/Window.cpp:60-90: update(dt); draw();)
// without a window.  Used by tests/test_host_cpp_gpu.py.
//   gmupt_render --scene cornell|file.gmesh|file.gltf|file.glb --size WxH --frames N --pool P --live L [--capture] [--dump out.f32] [--pfm out.pfm]
//                [--build-only] [--dump-mesh out.gmesh]   (what the loader produced, for the tests that feed it to the oracle)
//                [--models-root DIR] [--list-scenes]      (SceneParams registry of the reference, Source/Scene.cpp:22-80)
//                [--ranks N --rank R --rendezvous FILE [--device D] [--no-gather]]   one process per GPU: this process renders row band R of N on
//                                                         device D (default R) and the bands are gathered on rank 0 over RCCL (TileGather.hpp);
//                                                         rank 0 writes --dump / --pfm of the whole frame; --no-gather: every rank dumps its band
//                [--print-bands H N]                      the row split, as JSON
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <exception>
#include <stdexcept>
#include <string>
#include "Renderer.hpp"
#include "TileGather.hpp"
#include <hip/hip_runtime_api.h>
#include <memory>
#include <vector>

namespace {
void writeFloats(const std::string& path, const std::vector<float>& v)
{
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f || std::fwrite(v.data(), 4, v.size(), f) != v.size()) throw std::runtime_error("cannot write " + path);
	std::fclose(f);
}
void writePfmFile(const std::string& path, const std::vector<float>& rgba, unsigned w, unsigned h)   // "PF", little-endian, bottom row first
{
	std::FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) throw std::runtime_error("Failed to write " + path);
	std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
	std::vector<float> row(static_cast<size_t>(w) * 3);
	for (unsigned y = h; y-- > 0;)
	{
		for (unsigned x = 0; x < w; x++) for (int c = 0; c < 3; c++) row[static_cast<size_t>(x) * 3 + static_cast<size_t>(c)] = rgba[(static_cast<size_t>(y) * w + x) * 4 + static_cast<size_t>(c)];
		std::fwrite(row.data(), sizeof(float), row.size(), f);
	}
	std::fclose(f);
}
}

int main(int argc, char** argv)
{
	std::string scene = "cornell", dump, pfm, dumpMesh;
	bool listScenes = false;
	unsigned w = WIDTH, h = HEIGHT, frames = 16, pool = PATHCOUNT, live = REFERENCE_LIVE_PATHS;
	bool capture = false, buildOnly = false, noGather = false;
	unsigned ranks = 1, rank = 0; int device = -1;
	std::string paramsOnly, rendezvous;
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
		if (a == "--scene") scene = next();
		else if (a == "--size") { if (std::sscanf(next(), "%ux%u", &w, &h) != 2) return 2; }
		else if (a == "--frames") frames = std::strtoul(next(), nullptr, 10);
		else if (a == "--pool") pool = std::strtoul(next(), nullptr, 10);
		else if (a == "--live") live = std::strtoul(next(), nullptr, 10);
		else if (a == "--dump") dump = next();
		else if (a == "--pfm") pfm = next();
		else if (a == "--capture") capture = true;
		else if (a == "--build-only") buildOnly = true;
		else if (a == "--params") paramsOnly = next();
		else if (a == "--dump-mesh") dumpMesh = next();
		else if (a == "--models-root") SceneParams::instance.loadScenes(next());
		else if (a == "--list-scenes") listScenes = true;
		else if (a == "--ranks") ranks = std::strtoul(next(), nullptr, 10);
		else if (a == "--rank") rank = std::strtoul(next(), nullptr, 10);
		else if (a == "--rendezvous") rendezvous = next();
		else if (a == "--device") device = std::atoi(next());
		else if (a == "--no-gather") noGather = true;
		else if (a == "--print-bands") { // H N: the row bands of an H-row frame over N ranks, as JSON (the CPU tests compare them with tiles.py)
			const unsigned H = std::strtoul(next(), nullptr, 10), N = std::strtoul(next(), nullptr, 10);
			std::printf("[");
			for (unsigned r = 0; r < N; r++) { const auto b = gmupt::rowBand(H, N, r); std::printf("%s[%u, %u]", r ? ", " : "", b.first, b.second); }
			std::printf("]\n");
			return 0;
		}
		else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
	}
	try
	{
		if (!paramsOnly.empty()) { // SceneParams: the per-scene CSV of the reference (Source/Scene.cpp:34-62)
			const auto e = SceneParams::load(paramsOnly);
			std::printf("{\"camera\": [%.9g, %.9g, %.9g, %.9g, %.9g], \"lights\": [", e.camera.position[0], e.camera.position[1], e.camera.position[2], e.camera.pitch, e.camera.yaw);
			for (size_t i = 0; i < e.lights.size(); i++) {
				const Light& l = e.lights[i];
				std::printf("%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", l.position[0], l.position[1], l.position[2], l.falloff, l.emission[0], l.emission[1], l.emission[2], l.radius);
			}
			std::printf("]}\n");
			return 0;
		}
		if (listScenes) { // the registry: name, camera, number of lights per scene below the models directory
			SceneParams& reg = SceneParams::instance;
			reg.contains("");
			std::printf("[");
			for (size_t i = 0; i < reg.pathNames.size(); i++) {
				const auto& c = reg.cameraParams[i];
				std::printf("%s{\"name\": \"%s\", \"index\": %zu, \"camera\": [%.9g, %.9g, %.9g, %.9g, %.9g], \"lights\": %zu}", i ? ", " : "", reg.pathsReference[i], reg.getSceneIndex(reg.pathNames[i]),
				            c.position[0], c.position[1], c.position[2], c.pitch, c.yaw, reg.lights[i].size());
			}
			std::printf("]\n");
			return 0;
		}
		if (buildOnly) { // host-only leg (BASELINE config 1): scene load + SBVH build + flatten, no GPU
			const bool gltf = (scene.size() > 5 && scene.compare(scene.size() - 5, 5, ".gltf") == 0) || (scene.size() > 4 && scene.compare(scene.size() - 4, 4, ".glb") == 0);
			{ // like Scene::Scene: a scene below the models directory must be one the registry knows (Source/Scene.cpp:75-80,95)
				SceneParams& reg = SceneParams::instance;
				if (!reg.modelsRoot.empty() && scene.compare(0, reg.modelsRoot.size(), reg.modelsRoot) == 0) reg.getSceneIndex(scene.substr(reg.modelsRoot.size()));
			}
			MeshData mesh = (scene == "cornell") ? MeshData::cornell() : gltf ? MeshData::loadGltf(scene) : MeshData::load(scene);
			if (!dumpMesh.empty()) mesh.save(dumpMesh);
			// texture ingestion without the upload: layers, common size and checksum per texture type (Scene.cpp:209-244,268-285)
			std::string texInfo = "[";
			for (int t = 0; t < 3; t++) {
				gmupt::TextureSet set;
				if (!mesh.textureFiles[t].empty()) set = gmupt::loadSpecificTexture(mesh.textureFiles[t], mesh.materials, t);
				unsigned long long sum = 0; for (const auto& layer : set.layers) for (uint8_t v : layer) sum = sum * 31ull + v;
				char buf[128]; std::snprintf(buf, sizeof(buf), "%s{\"layers\": %zu, \"size\": %u, \"checksum\": %llu}", t ? ", " : "", set.layers.size(), set.dimension, sum);
				texInfo += buf;
			}
			texInfo += "], \"texture_indices\": [";
			for (size_t i = 0; i < mesh.materials.size(); i++) { char buf[96]; std::snprintf(buf, sizeof(buf), "%s[%d, %d, %d]", i ? ", " : "", mesh.materials[i].textureIndices[0], mesh.materials[i].textureIndices[1], mesh.materials[i].textureIndices[2]); texInfo += buf; }
			texInfo += "]";
			const auto t0 = std::chrono::steady_clock::now();
			BVHWrapper bvh(mesh);
			const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
			for (size_t i = 0; i < mesh.numVertices(); i++) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], mesh.vertices[3 * i + k]); hi[k] = std::max(hi[k], mesh.vertices[3 * i + k]); }
			size_t glass = 0; for (const auto& m : mesh.materials) glass += m.materialType == GMUPT_MATERIAL_GLASS;
			std::printf("{\"triangles\": %zu, \"vertices\": %zu, \"materials\": %zu, \"glass_materials\": %zu, \"nodes\": %zu, \"references\": %zu, \"sah\": %.6f, \"build_s\": %.4f, \"bbox\": [%.6f, %.6f, %.6f, %.6f, %.6f, %.6f], \"textures\": %s}\n",
			            mesh.numTriangles(), mesh.numVertices(), mesh.materials.size(), glass, bvh.tree().size(), bvh.indices().size(), bvh.sah(), s, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], texInfo.c_str());
			return 0;
		}
		if (ranks > 1 || !rendezvous.empty())
		{
			// one process per GPU: row band `rank` of `ranks`, the camera of the whole frame; the bands meet on rank 0 (RCCL send / receive)
			if (ranks == 0 || rank >= ranks) throw std::invalid_argument("--rank must be below --ranks");
			if (rendezvous.empty() && !noGather) throw std::invalid_argument("--ranks needs --rendezvous FILE (or --no-gather)");
			if (device < 0) device = static_cast<int>(rank);
			const auto band = gmupt::rowBand(h, ranks, rank);
			if (band.second == 0) throw std::invalid_argument("more ranks than rows");
			std::unique_ptr<gmupt::TileGather> gather;
			if (!noGather) gather.reset(new gmupt::TileGather(rank, ranks, device, rendezvous));   // first: every rank reaches the rendezvous before the long part
			Renderer renderer(nullptr, { w, h }, scene, device, pool, live, Renderer::RowBand{ band.first, band.second });
			for (unsigned f = 0; f < frames; f++) { renderer.update(0.f); renderer.draw(); }
			if (noGather) { if (!dump.empty()) writeFloats(dump, renderer.readFramebuffer()); }
			else
			{
				void* bandBuffer = nullptr;
				const size_t bytes = static_cast<size_t>(w) * band.second * 4 * sizeof(float);
				if (hipSetDevice(device) != hipSuccess || hipMalloc(&bandBuffer, bytes) != hipSuccess) throw std::runtime_error("cannot allocate the band buffer");
				renderer.copyFramebufferToDevice(bandBuffer);
				const std::vector<float> frame = gather->gatherToRoot(bandBuffer, w, h);
				(void)hipFree(bandBuffer);
				if (rank == 0) { if (!dump.empty()) writeFloats(dump, frame); if (!pfm.empty()) writePfmFile(pfm, frame, w, h); }
			}
			std::printf("rank %u of %u: rendered %llu iterations of rows %u..%u of %ux%u\n", rank, ranks, renderer.iterations(), band.first, band.first + band.second, w, h);
			return 0;
		}
		Renderer renderer(nullptr, { w, h }, scene, 0, pool, live);
		for (unsigned f = 0; f < frames; f++) { renderer.update(0.f); renderer.draw(); }
		if (capture) { renderer.requestCapture(); renderer.update(0.f); std::printf("capture %s\n", renderer.lastCapturePath().c_str()); }
		if (!pfm.empty()) renderer.writePfm(pfm);
		if (!dump.empty()) {
			const auto fb = renderer.readFramebuffer();
			FILE* f = std::fopen(dump.c_str(), "wb");
			if (!f || std::fwrite(fb.data(), 4, fb.size(), f) != fb.size()) throw std::runtime_error("cannot write " + dump);
			std::fclose(f);
		}
		std::printf("rendered %llu iterations of %ux%u\n", renderer.iterations(), w, h);
	}
	catch (const std::exception& e) // main.cpp:19-23
	{
		std::fprintf(stderr, "error: %s\n", e.what());
		return -1;
	}
	return 0;
}
