// Host-side split-BVH builder (Stich, Friedrich, Dietrich: "Spatial Splits in Bounding Volume Hierarchies", HPG 2009) that emits the
// reference's flattened buffers.  It takes the place of the vendored builder the reference drives from BVHWrapper::buildSBVH
// (Source/BVHWrapper.cpp:13-96) and must produce THE SAME TREE for the reference's default Platform / BuildParams: SAH costs 1 / 1,
// min leaf 1, no max leaf, splitAlpha 1e-5, max depth 64, spatial splits down to depth 48 with 32 bins
// (Include/Nvidia-SBVH/SplitBVHBuilder.h:36-41, Include/Nvidia-SBVH/Util.h:73, Include/Nvidia-SBVH/BVH.h:72-78) -- node boxes, leaf
// contents and the order of the references inside every leaf decide which hit wins an exact tie on the GPU.
//
// Design (host-first, nothing of the reference's program structure):
//   * references are 32-byte records in a chunked, append-only pool; nodes never move them, they hold index lists;
//   * every node task carries its references in THREE lists, pre-sorted along x, y and z by the reference's total order (centroid
//     along the axis, then triangle id).  They are sorted ONCE for the root (parallel merge sort on 64-bit keys) and handed down the
//     tree by stable partition; references created by a spatial split are sorted among themselves and merged in.  The SAH sweep of a
//     node is three linear passes over those lists, not three sorts;
//   * the tree is built by a pool of workers from a shared LIFO of node tasks (big nodes near the root also fan their sweeps,
//     binning and partitions out over helper threads); subtrees below a size threshold are finished by the worker that owns them on a
//     private stack.  Nodes live in per-worker arenas and are linked by pointers, so no numbering depends on the schedule; flatten()
//     numbers them afterwards in the reference's order.
// Which rule of the reference every step reproduces is cited in sbvh_builder.cpp.  The algorithm is the published one (HPG 2009); the
// tie-breaking and arithmetic conventions that make the tree identical are those of the BSD-licensed implementation the reference
// vendors (Copyright (c) 2009-2011, NVIDIA Corporation); none of its code is used here.
#pragma once
#include <cstdint>
#include <memory>
#include "../../include/gmupt.h"

namespace gmupt {

class SbvhBuilder {
public:
    SbvhBuilder(const float* vertices, uint32_t numVertices, const int32_t* indices, uint32_t numTriangles, const gmupt_sbvh_params& params);
    ~SbvhBuilder();
    SbvhBuilder(const SbvhBuilder&) = delete;
    SbvhBuilder& operator=(const SbvhBuilder&) = delete;

    void build();

    uint32_t numNodes() const;
    uint32_t numReferences() const;   // triangle references in all leaves (triangles + duplicates made by spatial splits)
    uint32_t numDuplicates() const;
    uint32_t depth() const;           // level of the deepest node, root = 0
    float sah() const;                // expected traversal cost, summed as BVHNode::computeSubtreeProbabilities does (Source/Nvidia-SBVH/BVHNode.cpp:65-79)

    // Reference layout (Source/BVHWrapper.cpp:56-95): root at 0, the two children of an inner node in consecutive slots
    // (right == left + 1), leaves index a triangle array filled in depth-first order, left subtree first.
    // outNodes: numNodes() records; outTris / outRefTri (either may be null): numReferences() entries.
    void flatten(const uint32_t* vertexMaterial, gmupt_bvh_node* outNodes, gmupt_triangle* outTris, int32_t* outRefTri) const;

private:
    struct Impl;
    std::unique_ptr<Impl> m;
};

} // namespace gmupt
