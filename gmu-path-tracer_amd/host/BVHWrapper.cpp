#include "BVHWrapper.hpp"
#include "sbvh_builder.hpp"
#include <cstring>

BVHWrapper::BVHWrapper(const MeshData& scene)
{
	buildSBVH(scene);
}

void BVHWrapper::buildSBVH(const MeshData& scene)
{
	mVertices = scene.vertices;

	// one TriangleProperties per vertex (Source/BVHWrapper.cpp:30-37)
	mTriangleProperties.resize(scene.numVertices());
	for (size_t n = 0; n < scene.numVertices(); n++)
	{
		auto& p = mTriangleProperties[n];
		std::memset(&p, 0, sizeof(p));
		for (int k = 0; k < 3; k++) p.normal[k] = scene.normals[3 * n + k];
		if (!scene.texCoords.empty()) { p.uv[0] = scene.texCoords[2 * n]; p.uv[1] = scene.texCoords[2 * n + 1]; }
		p.materialID = scene.vertexMaterial[n];
	}

	gmupt_sbvh_params params;
	gmupt_sbvh_default_params(&params); // default Platform / BuildParams of Source/BVHWrapper.cpp:52-54
	gmupt::SbvhBuilder builder(mVertices.data(), static_cast<uint32_t>(scene.numVertices()), scene.indices.data(),
	                           static_cast<uint32_t>(scene.numTriangles()), params);
	builder.build();
	mSAH = builder.sah();

	mGPUTree.resize(builder.numNodes());
	mIndices.resize(builder.numReferences());
	builder.flatten(scene.vertexMaterial.data(), mGPUTree.data(), mIndices.data(), nullptr);
}
