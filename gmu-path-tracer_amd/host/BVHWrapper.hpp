// Same role and nested PODs as the reference's BVHWrapper (Include/BVHWrapper.hpp:10-51): turns the imported meshes into the
// flattened SBVH buffers the ray-cast kernels consume.  The aiScene input becomes MeshData; the vendored Nvidia-SBVH classes
// are replaced by gmupt::SbvhBuilder.
#pragma once
#include <vector>
#include "MeshData.hpp"

class BVHWrapper
{
public:
	using BVHNode = gmupt_bvh_node;                 // 48 B, Include/BVHWrapper.hpp:13-21
	using Triangle = gmupt_triangle;                // 16 B, :23-27
	using TriangleProperties = gmupt_tri_props;     // 32 B per vertex, :29-34

	BVHWrapper() = default;
	explicit BVHWrapper(const MeshData& scene);

	const std::vector<BVHNode>& tree() const { return mGPUTree; }
	const std::vector<Triangle>& indices() const { return mIndices; }
	const std::vector<TriangleProperties>& triangleProperties() const { return mTriangleProperties; }
	const std::vector<float>& vertices() const { return mVertices; }
	float sah() const { return mSAH; }

private:
	void buildSBVH(const MeshData& scene);

	std::vector<BVHNode> mGPUTree;
	std::vector<Triangle> mIndices;
	std::vector<TriangleProperties> mTriangleProperties;
	std::vector<float> mVertices;
	float mSAH = 0.f;

	friend class Scene;
};
