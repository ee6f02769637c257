// Host-side split-BVH builder (Stich, Friedrich, Dietrich: "Spatial Splits in Bounding
// Volume Hierarchies", HPG 2009) that emits the reference's flattened buffers.
//
// Replaces the vendored builder of the reference (Source/Nvidia-SBVH/SplitBVHBuilder.cpp,
// driven by BVHWrapper::buildSBVH, Source/BVHWrapper.cpp:13-96).  It is a new
// implementation (index-based node pool, std::sort, no virtual node classes); the split
// decisions follow the same rules and the same reference-stack discipline, so that the
// resulting tree is the one the reference's default Platform / BuildParams would build:
//   SAH costs 1/1, min leaf 1, no max leaf, splitAlpha 1e-5, max depth 64, spatial
//   splits down to depth 48 with 32 bins (Include/Nvidia-SBVH/SplitBVHBuilder.h:36-41,
//   Include/Nvidia-SBVH/Util.h:73, Include/Nvidia-SBVH/BVH.h:72-78).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/gmupt.h"

namespace gmupt {

struct Aabb {
    float mn[3], mx[3];
    Aabb();
    void grow(const float* p);
    void grow(const Aabb& o);
    void clip(const Aabb& o);
    bool valid() const;
    float area() const;
};

struct SbvhNode {
    Aabb bounds;
    int32_t child[2]; // -1 for a leaf
    int32_t lo, hi;   // leaf: range in refTriangles
};

class SbvhBuilder {
public:
    SbvhBuilder(const float* vertices, uint32_t numVertices, const int32_t* indices, uint32_t numTriangles,
                const gmupt_sbvh_params& params);
    void build();

    const std::vector<SbvhNode>& nodes() const { return mNodes; }
    int32_t root() const { return mRoot; }
    const std::vector<int32_t>& refTriangles() const { return mRefTriangles; }
    uint32_t numDuplicates() const { return mNumDuplicates; }
    uint32_t depth() const { return mDepth; }
    float sah() const;

    // Reference layout (Source/BVHWrapper.cpp:56-95): root at 0, the two children of an inner node
    // in consecutive slots (right == left + 1), leaves index a triangle array filled in DFS order.
    void flatten(const uint32_t* vertexMaterial, gmupt_bvh_node* outNodes, gmupt_triangle* outTris, int32_t* outRefTri) const;

private:
    struct Ref { int32_t tri; Aabb b; };
    struct Spec { int32_t numRef; Aabb b; };
    struct ObjSplit { float sah; int dim; int numLeft; Aabb lb, rb; };
    struct SpaSplit { float sah; int dim; float pos; };

    // sub-builder of one subtree: takes the top `numRef` references of the parent's stack (the parent keeps building the sibling on its
    // own thread and absorbs the result afterwards).  The tree and every leaf's reference order are those of the sequential build.
    SbvhBuilder(SbvhBuilder& parent, int numRef);
    int32_t absorb(const SbvhBuilder& sub, int32_t subRoot);
    int32_t buildNode(const Spec& spec, int level);
    int32_t makeLeaf(const Spec& spec);
    ObjSplit findObjectSplit(const Spec& spec, float nodeSAH);
    void doObjectSplit(Spec& l, Spec& r, const Spec& spec, const ObjSplit& s);
    SpaSplit findSpatialSplit(const Spec& spec, float nodeSAH);
    void doSpatialSplit(Spec& l, Spec& r, const Spec& spec, const SpaSplit& s);
    void splitRef(Ref& l, Ref& r, const Ref& ref, int dim, float pos) const;
    void sortTail(int numRef, int dim);
    float triCost(int n) const { return (float)n * mP.tri_cost; }

    const float* mVerts;
    const int32_t* mIdx;
    uint32_t mNumTris;
    gmupt_sbvh_params mP;

    std::vector<Ref> mStack;
    std::vector<Aabb> mRight;
    std::vector<SbvhNode> mNodes;
    std::vector<int32_t> mRefTriangles;
    float mMinOverlap = 0.f;
    uint32_t mNumDuplicates = 0;
    uint32_t mDepth = 0;
    int32_t mRoot = -1;
};

} // namespace gmupt
