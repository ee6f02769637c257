#include "MeshData.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace {
struct File { FILE* f; ~File() { if (f) std::fclose(f); } };
template <class T> void readVec(FILE* f, std::vector<T>& v, size_t n, const std::string& path)
{
	v.resize(n);
	if (n && std::fread(v.data(), sizeof(T), n, f) != n) throw std::runtime_error("Truncated mesh file " + path);
}
gmupt_material makeMaterial(float r, float g, float b, float metallic, float roughness, uint32_t type)
{
	gmupt_material m{};
	m.color[0] = r; m.color[1] = g; m.color[2] = b; m.color[3] = 1.f;
	m.metallic = metallic; m.roughness = roughness; m.refractIndex = 1.458f; m.transmittance = 0.f;
	m.textureIndices[0] = m.textureIndices[1] = m.textureIndices[2] = -1; // Scene.hpp:65
	m.materialType = type;
	return m;
}
}

MeshData MeshData::load(const std::string& path)
{
	File file{ std::fopen(path.c_str(), "rb") };
	if (!file.f) throw std::runtime_error("Non existing scene " + path); // wording of Scene.cpp:79
	char magic[8]; uint32_t hdr[4];
	if (std::fread(magic, 1, 8, file.f) != 8 || std::memcmp(magic, "GMESH001", 8) != 0 || std::fread(hdr, 4, 4, file.f) != 4)
		throw std::runtime_error("Not a gmesh file: " + path);
	MeshData m;
	const size_t nv = hdr[0], nt = hdr[1], nm = hdr[2], hasUV = hdr[3];
	{ // the header must fit the file (a corrupt count must not become a multi-gigabyte allocation)
		const long at = std::ftell(file.f); std::fseek(file.f, 0, SEEK_END); const long end = std::ftell(file.f); std::fseek(file.f, at, SEEK_SET);
		const unsigned long long need = 4ull * (nv * 3 + nv * 3 + (hasUV ? nv * 2 : 0) + nv + nt * 3) + sizeof(gmupt_material) * (unsigned long long)nm;
		if (at < 0 || end < at || need > (unsigned long long)(end - at)) throw std::runtime_error("Truncated mesh file " + path);
	}
	readVec(file.f, m.vertices, nv * 3, path);
	readVec(file.f, m.normals, nv * 3, path);
	if (hasUV) readVec(file.f, m.texCoords, nv * 2, path);
	readVec(file.f, m.vertexMaterial, nv, path);
	readVec(file.f, m.indices, nt * 3, path);
	readVec(file.f, m.materials, nm, path);
	// a dump is input like any other file: nothing in it may index outside its own arrays
	for (int32_t v : m.indices) if (v < 0 || static_cast<size_t>(v) >= nv) throw std::runtime_error("Corrupt mesh file (vertex index out of range): " + path);
	for (uint32_t mat : m.vertexMaterial) if (mat >= (nm ? nm : 1)) throw std::runtime_error("Corrupt mesh file (material index out of range): " + path);
	if (nm == 0) throw std::runtime_error("Corrupt mesh file (no materials): " + path);
	return m;
}

void MeshData::save(const std::string& path) const
{
	File file{ std::fopen(path.c_str(), "wb") };
	if (!file.f) throw std::runtime_error("Cannot write " + path);
	const uint32_t hdr[4] = { static_cast<uint32_t>(numVertices()), static_cast<uint32_t>(numTriangles()), static_cast<uint32_t>(materials.size()), texCoords.empty() ? 0u : 1u };
	bool ok = std::fwrite("GMESH001", 1, 8, file.f) == 8 && std::fwrite(hdr, 4, 4, file.f) == 4;
	auto put = [&](const void* data, size_t bytes) { ok = ok && (bytes == 0 || std::fwrite(data, 1, bytes, file.f) == bytes); };
	put(vertices.data(), vertices.size() * 4); put(normals.data(), normals.size() * 4);
	if (!texCoords.empty()) put(texCoords.data(), texCoords.size() * 4);
	put(vertexMaterial.data(), vertexMaterial.size() * 4); put(indices.data(), indices.size() * 4); put(materials.data(), materials.size() * sizeof(gmupt_material));
	if (!ok) throw std::runtime_error("Cannot write " + path);
}

MeshData MeshData::cornell()
{
	MeshData m;
	auto quad = [&m](const double p[4][3], const double n[3], uint32_t mat) {
		const int32_t base = static_cast<int32_t>(m.numVertices());
		for (int k = 0; k < 4; k++) {
			for (int c = 0; c < 3; c++) { m.vertices.push_back(static_cast<float>(p[k][c])); m.normals.push_back(static_cast<float>(n[c])); }
			m.vertexMaterial.push_back(mat);
		}
		const int32_t t[6] = { base, base + 1, base + 2, base, base + 2, base + 3 };
		m.indices.insert(m.indices.end(), t, t + 6);
	};
	const double x0 = -5, y0 = 0, z0 = -5, x1 = 5, y1 = 10, z1 = 5;
	{ const double p[4][3] = { {x0,y0,z1},{x1,y0,z1},{x1,y0,z0},{x0,y0,z0} }, n[3] = { 0,1,0 }; quad(p, n, 0); }
	{ const double p[4][3] = { {x0,y1,z0},{x1,y1,z0},{x1,y1,z1},{x0,y1,z1} }, n[3] = { 0,-1,0 }; quad(p, n, 0); }
	{ const double p[4][3] = { {x0,y0,z0},{x1,y0,z0},{x1,y1,z0},{x0,y1,z0} }, n[3] = { 0,0,1 }; quad(p, n, 0); }
	{ const double p[4][3] = { {x0,y0,z1},{x0,y0,z0},{x0,y1,z0},{x0,y1,z1} }, n[3] = { 1,0,0 }; quad(p, n, 1); }
	{ const double p[4][3] = { {x1,y0,z0},{x1,y0,z1},{x1,y1,z1},{x1,y1,z0} }, n[3] = { -1,0,0 }; quad(p, n, 2); }
	auto box = [&](const double c[3], const double h[3], double yawDeg, uint32_t mat) {
		const double a = yawDeg * 3.14159265358979323846 / 180.0, ca = std::cos(a), sa = std::sin(a);
		auto rot = [&](const double v[3], double o[3]) { o[0] = ca * v[0] + sa * v[2]; o[1] = v[1]; o[2] = -sa * v[0] + ca * v[2]; };
		static const double faces[6][5][3] = {
			{ {1,0,0}, {1,-1,-1},{1,1,-1},{1,1,1},{1,-1,1} }, { {-1,0,0}, {-1,-1,1},{-1,1,1},{-1,1,-1},{-1,-1,-1} },
			{ {0,1,0}, {-1,1,-1},{-1,1,1},{1,1,1},{1,1,-1} }, { {0,-1,0}, {-1,-1,1},{-1,-1,-1},{1,-1,-1},{1,-1,1} },
			{ {0,0,1}, {-1,-1,1},{1,-1,1},{1,1,1},{-1,1,1} }, { {0,0,-1}, {1,-1,-1},{-1,-1,-1},{-1,1,-1},{1,1,-1} } };
		for (const auto& f : faces) {
			double n[3]; rot(f[0], n);
			double p[4][3];
			for (int k = 0; k < 4; k++) {
				const double local[3] = { h[0] * f[k + 1][0], h[1] * f[k + 1][1], h[2] * f[k + 1][2] };
				double r[3]; rot(local, r);
				for (int q = 0; q < 3; q++) p[k][q] = c[q] + r[q];
			}
			quad(p, n, mat);
		}
	};
	{ const double c[3] = { -1.8, 3.0, -1.5 }, h[3] = { 1.5, 3.0, 1.5 }; box(c, h, 18.0, 0); }
	{ const double c[3] = { 1.7, 1.5, 1.0 }, h[3] = { 1.5, 1.5, 1.5 }; box(c, h, -17.0, 0); }
	m.materials = { makeMaterial(0.73f, 0.73f, 0.73f, 0.f, 1.f, GMUPT_MATERIAL_UE4), makeMaterial(0.65f, 0.05f, 0.05f, 0.f, 1.f, GMUPT_MATERIAL_UE4),
	                makeMaterial(0.12f, 0.45f, 0.15f, 0.f, 1.f, GMUPT_MATERIAL_UE4) };
	return m;
}
