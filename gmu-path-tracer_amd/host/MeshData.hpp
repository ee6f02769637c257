// In-memory triangle scene: what the reference gets from assimp after its post-process flags
// (Source/Scene.cpp:113-121: triangulated, world-space pre-transformed vertices, per-vertex smooth normals, flipped UVs),
// reduced to the fields BVHWrapper reads (Source/BVHWrapper.cpp:17-47) plus the glTF material factors of Scene.cpp:130-146.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/gmupt.h"

struct MeshData
{
	std::vector<float> vertices;          // 3 per vertex
	std::vector<float> normals;           // 3 per vertex
	std::vector<float> texCoords;         // 2 per vertex (may be empty)
	std::vector<uint32_t> vertexMaterial; // material index of the mesh each vertex came from
	std::vector<int32_t> indices;         // 3 per triangle
	std::vector<gmupt_material> materials;
	// encoded PNG file of every material per texture type (0 base colour, 1 metallic-roughness, 2 normal; empty = none):
	// what aiMaterial::GetTexture(type, 0, &path) + the file behind `path` are to the reference (Source/Scene.cpp:218-226)
	std::vector<std::vector<uint8_t>> textureFiles[3];

	size_t numVertices() const { return vertices.size() / 3; }
	size_t numTriangles() const { return indices.size() / 3; }

	// ".gmesh": little-endian dump written by gmu-path-tracer_amd/scenes.py (save_gmesh); throws std::runtime_error
	static MeshData load(const std::string& path);
	// writes the same ".gmesh" format (what a loader produced, for inspection and for the tests that feed it to the oracle)
	void save(const std::string& path) const;
	// glTF 2.0 (.gltf + external .bin or base64 buffers, or a .glb container): triangulated, pre-transformed, smooth normals, flipped UVs -- the effect of the
	// reference's assimp flags (Source/Scene.cpp:113-121); throws std::runtime_error
	static MeshData loadGltf(const std::string& path);
	// Config-2 Cornell box (34 triangles), identical to scenes.cornell_mesh()
	static MeshData cornell();
};
