// Minimal glTF 2.0 reader producing what the reference gets from assimp with its post-process flags (Source/Scene.cpp:113-121), as the
// flags are documented in the assimp headers the reference vendors (Include/assimp/postprocess.h, config.h):
//   aiProcess_Triangulate + SortByPType  only TRIANGLES primitives are kept (like BVHWrapper.cpp:22 ignores the other meshes);
//   aiProcess_PreTransformVertices       node transforms baked into world-space vertices, the node graph flattened to one mesh per
//                                        material (and vertex format): postprocess.h:207-222;
//   aiProcess_GenSmoothNormals           only for meshes without authored normals: the face normals of all faces "at the same vertex
//                                        position" are smoothed together, default angle limit 175 degrees = always (postprocess.h:169-183,
//                                        config.h:163-173).  Weights are not documented; assimp's published step sums the UNIT face normals;
//   aiProcess_JoinIdenticalVertices      "identifies and joins identical vertex data sets within all imported meshes"; without it "no
//                                        vertices are referenced by more than one face" (postprocess.h:84-94): the pipeline works on one
//                                        vertex per face corner and this step re-indexes them -- vertices come out in the order of their
//                                        first use, duplicates (same position, normal, uv) are merged, also across the primitives that
//                                        PreTransformVertices merged into one mesh;
//   aiProcess_FlipUVs                    v -> 1 - v;
//   material factors as read in Scene.cpp:130-146: baseColorFactor, metallicFactor, roughnessFactor, alphaMode == "BLEND" -> GLASS.
// assimp itself is not available (headers only in the reference, its .lib is an LFS stub), so this restates the documented effect of
// those flags, not assimp's code: PARITY UNPINNED -- positions, triangles and authored normals are exact by
// construction; generated normals and the vertex numbering follow the documentation.  Image decoding / texture arrays: the loader only
// fetches the encoded image of every material's baseColorTexture / metallicRoughnessTexture / normalTexture (assimp's
// aiTextureType_DIFFUSE / _UNKNOWN / _NORMALS for glTF 2.0); decoding and layer assignment happen in TextureLoader.cpp.
#include "MeshData.hpp"
#include "json_min.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

namespace {
using gmupt::Json;

std::string readFile(const std::string& path, bool binary)
{
	std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
	if (!f.is_open()) throw std::runtime_error("Non existing scene " + path);
	std::stringstream ss; ss << f.rdbuf();
	return ss.str();
}

std::string decodeBase64(const std::string& in)
{
	std::string out; int val = 0, bits = -8;
	for (unsigned char c : in) {
		int d;
		if (c >= 'A' && c <= 'Z') d = c - 'A'; else if (c >= 'a' && c <= 'z') d = c - 'a' + 26; else if (c >= '0' && c <= '9') d = c - '0' + 52;
		else if (c == '+') d = 62; else if (c == '/') d = 63; else continue;
		val = (val << 6) | d; bits += 6;
		if (bits >= 0) { out.push_back(static_cast<char>((val >> bits) & 0xFF)); bits -= 8; }
	}
	return out;
}

struct Mat4 { double m[16]; }; // column-major like glTF

Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0; return r; }
Mat4 mul(const Mat4& a, const Mat4& b)
{
	Mat4 r{};
	for (int c = 0; c < 4; c++) for (int rr = 0; rr < 4; rr++) { double s = 0; for (int k = 0; k < 4; k++) s += a.m[k * 4 + rr] * b.m[c * 4 + k]; r.m[c * 4 + rr] = s; }
	return r;
}
Mat4 nodeMatrix(const Json& n)
{
	if (n.has("matrix")) { Mat4 r{}; for (int i = 0; i < 16; i++) r.m[i] = n["matrix"][i].number(); return r; }
	double t[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 1 }, s[3] = { 1, 1, 1 };
	if (n.has("translation")) for (int i = 0; i < 3; i++) t[i] = n["translation"][i].number();
	if (n.has("rotation")) for (int i = 0; i < 4; i++) q[i] = n["rotation"][i].number();
	if (n.has("scale")) for (int i = 0; i < 3; i++) s[i] = n["scale"][i].number();
	const double x = q[0], y = q[1], z = q[2], w = q[3];
	Mat4 r = identity();
	r.m[0] = (1 - 2 * (y * y + z * z)) * s[0]; r.m[1] = (2 * (x * y + z * w)) * s[0]; r.m[2] = (2 * (x * z - y * w)) * s[0];
	r.m[4] = (2 * (x * y - z * w)) * s[1]; r.m[5] = (1 - 2 * (x * x + z * z)) * s[1]; r.m[6] = (2 * (y * z + x * w)) * s[1];
	r.m[8] = (2 * (x * z + y * w)) * s[2]; r.m[9] = (2 * (y * z - x * w)) * s[2]; r.m[10] = (1 - 2 * (x * x + y * y)) * s[2];
	r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2];
	return r;
}

struct Gltf
{
	Json doc;
	std::vector<std::string> buffers;

	// reads accessor `index` as doubles, `comps` components per element (0 = scalar indices)
	std::vector<double> read(int index, int& count, int& comps) const
	{
		const Json& acc = doc["accessors"][static_cast<size_t>(index)];
		if (acc.type != Json::Object) throw std::runtime_error("glTF: missing accessor");
		const std::string& type = acc["type"].string();
		comps = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
		if (!comps) throw std::runtime_error("glTF: unsupported accessor type " + type);
		count = acc["count"].integer(0);
		const int ctype = acc["componentType"].integer();
		const size_t csize = (ctype == 5120 || ctype == 5121) ? 1 : (ctype == 5122 || ctype == 5123) ? 2 : 4;
		const Json& view = doc["bufferViews"][static_cast<size_t>(acc["bufferView"].integer(0))];
		const std::string& buf = buffers.at(static_cast<size_t>(view["buffer"].integer(0)));
		const size_t offset = static_cast<size_t>(view["byteOffset"].number(0)) + static_cast<size_t>(acc["byteOffset"].number(0));
		size_t stride = static_cast<size_t>(view["byteStride"].number(0));
		if (!stride) stride = csize * static_cast<size_t>(comps);
		if (offset + (count ? (static_cast<size_t>(count) - 1) * stride + csize * static_cast<size_t>(comps) : 0) > buf.size()) throw std::runtime_error("glTF: accessor outside its buffer");
		std::vector<double> out(static_cast<size_t>(count) * static_cast<size_t>(comps));
		for (int i = 0; i < count; i++)
			for (int c = 0; c < comps; c++) {
				const char* p = buf.data() + offset + static_cast<size_t>(i) * stride + static_cast<size_t>(c) * csize;
				double v = 0;
				switch (ctype) {
				case 5120: { int8_t x; std::memcpy(&x, p, 1); v = x; } break;
				case 5121: { uint8_t x; std::memcpy(&x, p, 1); v = x; } break;
				case 5122: { int16_t x; std::memcpy(&x, p, 2); v = x; } break;
				case 5123: { uint16_t x; std::memcpy(&x, p, 2); v = x; } break;
				case 5125: { uint32_t x; std::memcpy(&x, p, 4); v = x; } break;
				case 5126: { float x; std::memcpy(&x, p, 4); v = x; } break;
				default: throw std::runtime_error("glTF: unsupported component type");
				}
				out[static_cast<size_t>(i) * static_cast<size_t>(comps) + static_cast<size_t>(c)] = v;
			}
		return out;
	}
};

// one vertex per face corner: the "verbose" form the post-process steps work on
struct Corner { float p[3]; float n[3]; float uv[2]; };
struct Part { uint32_t material; bool hasNormals, hasUV; std::vector<Corner> corners; };   // one glTF primitive, 3 corners per triangle

void addPrimitive(const Gltf& g, const Json& prim, const Mat4& world, std::vector<Part>& parts)
{
	if (prim["mode"].integer(4) != 4) return; // only triangles are rendered (Source/BVHWrapper.cpp:21-23)
	const Json& attrs = prim["attributes"];
	if (!attrs.has("POSITION")) return;
	int n = 0, comps = 0;
	const std::vector<double> pos = g.read(attrs["POSITION"].integer(), n, comps);
	if (comps != 3) throw std::runtime_error("glTF: POSITION must be VEC3");
	std::vector<double> nrm, uv;
	int nn = 0, nc = 0, un = 0, uc = 0;
	if (attrs.has("NORMAL")) nrm = g.read(attrs["NORMAL"].integer(), nn, nc);
	if (attrs.has("TEXCOORD_0")) uv = g.read(attrs["TEXCOORD_0"].integer(), un, uc);
	std::vector<int32_t> idx;
	if (prim.has("indices")) {
		int ic = 0, icomp = 0;
		const std::vector<double> raw = g.read(prim["indices"].integer(), ic, icomp);
		idx.resize(raw.size());
		for (size_t i = 0; i < raw.size(); i++) idx[i] = static_cast<int32_t>(raw[i]);
	} else { idx.resize(static_cast<size_t>(n)); for (int i = 0; i < n; i++) idx[static_cast<size_t>(i)] = i; }
	idx.resize(idx.size() / 3 * 3);
	for (int32_t i : idx) if (i < 0 || i >= n) throw std::runtime_error("glTF: index outside the vertex range");

	// world-space positions (aiProcess_PreTransformVertices)
	std::vector<double> wp(static_cast<size_t>(n) * 3);
	for (int i = 0; i < n; i++)
		for (int r = 0; r < 3; r++)
			wp[static_cast<size_t>(i) * 3 + static_cast<size_t>(r)] = world.m[r] * pos[static_cast<size_t>(i) * 3] + world.m[4 + r] * pos[static_cast<size_t>(i) * 3 + 1] + world.m[8 + r] * pos[static_cast<size_t>(i) * 3 + 2] + world.m[12 + r];
	// authored normals: transformed by the upper 3x3 (uniform scale assumed) and renormalised; under an identity transform they are kept bit
	// for bit.  A primitive without normals gets them in finishMeshes (aiProcess_GenSmoothNormals works on the merged mesh).
	std::vector<double> wn(static_cast<size_t>(n) * 3, 0.0);
	if (nn == n && nc == 3) {
		bool isIdentity = true;
		{ const Mat4 id = identity(); for (int k = 0; k < 16; k++) isIdentity = isIdentity && world.m[k] == id.m[k]; }
		for (int i = 0; i < n; i++) {
			double* v = &wn[static_cast<size_t>(i) * 3];
			for (int r = 0; r < 3; r++)
				v[r] = world.m[r] * nrm[static_cast<size_t>(i) * 3] + world.m[4 + r] * nrm[static_cast<size_t>(i) * 3 + 1] + world.m[8 + r] * nrm[static_cast<size_t>(i) * 3 + 2];
			if (isIdentity) continue;
			const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
			if (len > 0) { v[0] /= len; v[1] /= len; v[2] /= len; }
		}
	}

	Part part;
	part.material = static_cast<uint32_t>(prim["material"].integer(0) < 0 ? 0 : prim["material"].integer(0));
	part.hasNormals = (nn == n && nc == 3);
	part.hasUV = (un == n && uc == 2);
	part.corners.reserve(idx.size());
	for (int32_t i : idx) {
		Corner c{};
		for (int r = 0; r < 3; r++) { c.p[r] = static_cast<float>(wp[static_cast<size_t>(i) * 3 + static_cast<size_t>(r)]); c.n[r] = part.hasNormals ? static_cast<float>(wn[static_cast<size_t>(i) * 3 + static_cast<size_t>(r)]) : 0.f; }
		if (part.hasUV) { c.uv[0] = static_cast<float>(uv[static_cast<size_t>(i) * 2]); c.uv[1] = static_cast<float>(1.0 - uv[static_cast<size_t>(i) * 2 + 1]); } // aiProcess_FlipUVs
		part.corners.push_back(c);
	}
	parts.push_back(std::move(part));
}

struct Bits3 { uint32_t v[3]; bool operator==(const Bits3& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; } };
struct Bits8 { uint32_t v[8]; bool operator==(const Bits8& o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; } };
template <typename T> struct BitsHash { size_t operator()(const T& k) const { size_t h = 1469598103934665603ull; for (uint32_t w : k.v) { h ^= w; h *= 1099511628211ull; } return h; } };
inline uint32_t floatKey(float f) { f += 0.0f; uint32_t u; std::memcpy(&u, &f, 4); return u; }   // -0 and +0 are the same coordinate

// aiProcess_GenSmoothNormals on one mesh in verbose form: every corner gets the normalised sum of the unit normals of all faces that have a
// corner at the same position
void smoothNormals(std::vector<Corner>& corners)
{
	std::unordered_map<Bits3, size_t, BitsHash<Bits3>> slot;
	std::vector<float> sum;
	std::vector<size_t> slotOf(corners.size());
	for (size_t i = 0; i < corners.size(); i++) {
		const Bits3 key = { { floatKey(corners[i].p[0]), floatKey(corners[i].p[1]), floatKey(corners[i].p[2]) } };
		auto it = slot.find(key);
		if (it == slot.end()) { it = slot.emplace(key, sum.size() / 3).first; sum.insert(sum.end(), { 0.f, 0.f, 0.f }); }
		slotOf[i] = it->second;
	}
	for (size_t t = 0; t + 2 < corners.size(); t += 3) {
		const float* a = corners[t].p; const float* b = corners[t + 1].p; const float* c = corners[t + 2].p;
		const float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
		float fn[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
		const float len = std::sqrt(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
		if (len > 0.f) { fn[0] /= len; fn[1] /= len; fn[2] /= len; }
		for (int k = 0; k < 3; k++) for (int r = 0; r < 3; r++) sum[slotOf[t + static_cast<size_t>(k)] * 3 + static_cast<size_t>(r)] += fn[r];
	}
	for (size_t i = 0; i < corners.size(); i++) {
		const float* v = &sum[slotOf[i] * 3];
		const float len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
		for (int r = 0; r < 3; r++) corners[i].n[r] = len > 0.f ? v[r] / len : v[r];
	}
}

// PreTransformVertices' mesh merge, GenSmoothNormals and JoinIdenticalVertices over the collected primitives -> the indexed MeshData
void finishMeshes(std::vector<Part>& parts, MeshData& out)
{
	// one mesh per (material, vertex format), materials in ascending order, formats in order of first appearance
	std::map<uint32_t, std::vector<size_t>> byMaterial;
	for (size_t i = 0; i < parts.size(); i++) byMaterial[parts[i].material].push_back(i);
	bool anyUV = false;
	for (const Part& p : parts) anyUV = anyUV || p.hasUV;
	for (auto& entry : byMaterial) {
		std::vector<std::pair<bool, bool>> formats;
		for (size_t i : entry.second) { const std::pair<bool, bool> f{ parts[i].hasNormals, parts[i].hasUV }; if (std::find(formats.begin(), formats.end(), f) == formats.end()) formats.push_back(f); }
		for (const auto& f : formats) {
			std::vector<Corner> corners;
			for (size_t i : entry.second) if (parts[i].hasNormals == f.first && parts[i].hasUV == f.second) corners.insert(corners.end(), parts[i].corners.begin(), parts[i].corners.end());
			if (!f.first) smoothNormals(corners);
			std::unordered_map<Bits8, int32_t, BitsHash<Bits8>> joined;
			for (const Corner& c : corners) {
				const Bits8 key = { { floatKey(c.p[0]), floatKey(c.p[1]), floatKey(c.p[2]), floatKey(c.n[0]), floatKey(c.n[1]), floatKey(c.n[2]), floatKey(c.uv[0]), floatKey(c.uv[1]) } };
				auto it = joined.find(key);
				if (it == joined.end()) {
					it = joined.emplace(key, static_cast<int32_t>(out.numVertices())).first;
					for (int r = 0; r < 3; r++) { out.vertices.push_back(c.p[r]); out.normals.push_back(c.n[r]); }
					if (anyUV) { out.texCoords.push_back(c.uv[0]); out.texCoords.push_back(c.uv[1]); }
					out.vertexMaterial.push_back(entry.first);
				}
				out.indices.push_back(it->second);
			}
		}
	}
}

void visitNode(const Gltf& g, int nodeIndex, const Mat4& parent, std::vector<Part>& out, int depth)
{
	if (depth > 64) throw std::runtime_error("glTF: node hierarchy too deep");
	const Json& node = g.doc["nodes"][static_cast<size_t>(nodeIndex)];
	if (node.type != Json::Object) return;
	const Mat4 world = mul(parent, nodeMatrix(node));
	if (node.has("mesh")) {
		const Json& mesh = g.doc["meshes"][static_cast<size_t>(node["mesh"].integer(0))];
		for (size_t p = 0; p < mesh["primitives"].size(); p++) addPrimitive(g, mesh["primitives"][p], world, out);
	}
	for (size_t c = 0; c < node["children"].size(); c++) visitNode(g, node["children"][c].integer(0), world, out, depth + 1);
}
}

MeshData MeshData::loadGltf(const std::string& path)
{
	Gltf g;
	std::string glbBin; bool haveGlbBin = false;
	const bool glb = path.size() > 4 && path.compare(path.size() - 4, 4, ".glb") == 0;
	if (glb) {
		// binary container (glTF 2.0 specification, "GLB file format"): 12-byte header, then chunks (length, type, data): JSON first, BIN optional
		const std::string file = readFile(path, true);
		auto u32 = [&](size_t at) -> uint32_t { if (at + 4 > file.size()) throw std::runtime_error("GLB: truncated"); uint32_t v; std::memcpy(&v, file.data() + at, 4); return v; };
		if (u32(0) != 0x46546C67u || u32(4) != 2u) throw std::runtime_error("GLB: bad magic or version");
		size_t pos = 12; bool haveJson = false;
		while (pos + 8 <= file.size()) {
			const uint32_t len = u32(pos), type = u32(pos + 4);
			if (pos + 8 + (size_t)len > file.size()) throw std::runtime_error("GLB: chunk exceeds the file");
			if (type == 0x4E4F534Au && !haveJson) { g.doc = Json::parse(file.substr(pos + 8, len)); haveJson = true; }   // "JSON"
			else if (type == 0x004E4942u && !haveGlbBin) { glbBin = file.substr(pos + 8, len); haveGlbBin = true; }      // "BIN\0"
			pos += 8 + (size_t)len;
		}
		if (!haveJson) throw std::runtime_error("GLB: no JSON chunk");
	}
	else g.doc = Json::parse(readFile(path, false));
	const std::string dir = path.find_last_of("/\\") == std::string::npos ? std::string() : path.substr(0, path.find_last_of("/\\") + 1);
	for (size_t i = 0; i < g.doc["buffers"].size(); i++) {
		const std::string& uri = g.doc["buffers"][i]["uri"].string();
		const std::string tag = "base64,";
		if (uri.empty() && i == 0 && haveGlbBin) g.buffers.push_back(glbBin);                     // the BIN chunk is buffer 0 without a uri
		else if (uri.compare(0, 5, "data:") == 0 && uri.find(tag) != std::string::npos) g.buffers.push_back(decodeBase64(uri.substr(uri.find(tag) + tag.size())));
		else g.buffers.push_back(readFile(dir + uri, true));
	}
	MeshData out;
	// materials (Source/Scene.cpp:130-146)
	for (size_t i = 0; i < g.doc["materials"].size(); i++) {
		const Json& m = g.doc["materials"][i];
		const Json& pbr = m["pbrMetallicRoughness"];
		gmupt_material mp{};
		for (int k = 0; k < 4; k++) mp.color[k] = pbr.has("baseColorFactor") ? static_cast<float>(pbr["baseColorFactor"][static_cast<size_t>(k)].number(1.0)) : 1.f;
		mp.metallic = static_cast<float>(pbr["metallicFactor"].number(1.0));
		mp.roughness = static_cast<float>(pbr["roughnessFactor"].number(1.0));
		mp.refractIndex = 1.458f; mp.transmittance = 0.f;
		mp.textureIndices[0] = mp.textureIndices[1] = mp.textureIndices[2] = -1;
		mp.materialType = (m["alphaMode"].string() == "BLEND") ? GMUPT_MATERIAL_GLASS : GMUPT_MATERIAL_UE4; // Scene.cpp:138-143
		out.materials.push_back(mp);
	}
	// images referenced by the materials: textures[t].source -> images[i] = { uri | bufferView }
	auto imageBytes = [&](const Json& textureInfo) -> std::vector<uint8_t> {
		if (textureInfo.type != Json::Object) return {};
		const Json& tex = g.doc["textures"][static_cast<size_t>(textureInfo["index"].integer(0) < 0 ? 0 : textureInfo["index"].integer(0))];
		if (tex.type != Json::Object || tex["source"].integer(-1) < 0) return {};
		const Json& image = g.doc["images"][static_cast<size_t>(tex["source"].integer(0))];
		if (image.type != Json::Object) return {};
		std::string bytes;
		if (image.has("uri")) {
			const std::string& uri = image["uri"].string();
			const std::string tag = "base64,";
			bytes = (uri.compare(0, 5, "data:") == 0 && uri.find(tag) != std::string::npos) ? decodeBase64(uri.substr(uri.find(tag) + tag.size())) : readFile(dir + uri, true);
		} else if (image["bufferView"].integer(-1) >= 0) {
			const Json& view = g.doc["bufferViews"][static_cast<size_t>(image["bufferView"].integer(0))];
			const size_t buffer = static_cast<size_t>(view["buffer"].integer(0) < 0 ? 0 : view["buffer"].integer(0));
			const size_t offset = static_cast<size_t>(view["byteOffset"].number(0.0)), length = static_cast<size_t>(view["byteLength"].number(0.0));
			if (buffer >= g.buffers.size() || offset + length > g.buffers[buffer].size()) throw std::runtime_error("glTF: image bufferView out of range");
			bytes = g.buffers[buffer].substr(offset, length);
		}
		return std::vector<uint8_t>(bytes.begin(), bytes.end());
	};
	for (int t = 0; t < 3; t++) out.textureFiles[t].resize(g.doc["materials"].size());
	for (size_t i = 0; i < g.doc["materials"].size(); i++) {
		const Json& m = g.doc["materials"][i];
		out.textureFiles[0][i] = imageBytes(m["pbrMetallicRoughness"]["baseColorTexture"]);
		out.textureFiles[1][i] = imageBytes(m["pbrMetallicRoughness"]["metallicRoughnessTexture"]);
		out.textureFiles[2][i] = imageBytes(m["normalTexture"]);
	}
	if (out.materials.empty()) { gmupt_material mp{}; mp.color[0] = mp.color[1] = mp.color[2] = 0.6f; mp.color[3] = 1.f; mp.metallic = 0.f; mp.roughness = 1.f; mp.textureIndices[0] = mp.textureIndices[1] = mp.textureIndices[2] = -1; out.materials.push_back(mp); }
	const Json& scenes = g.doc["scenes"];
	const Json& scene = scenes[static_cast<size_t>(g.doc["scene"].integer(0) < 0 ? 0 : g.doc["scene"].integer(0))];
	std::vector<Part> parts;
	if (scene.type == Json::Object) for (size_t i = 0; i < scene["nodes"].size(); i++) visitNode(g, scene["nodes"][i].integer(0), identity(), parts, 0);
	else for (size_t i = 0; i < g.doc["nodes"].size(); i++) visitNode(g, static_cast<int>(i), identity(), parts, 0);
	finishMeshes(parts, out);
	for (uint32_t& m : out.vertexMaterial) if (m >= out.materials.size()) m = 0;
	if (out.numTriangles() == 0) throw std::runtime_error("glTF: no triangles in " + path);
	return out;
}
