// Sanitizer driver of the host SBVH builder (CPU only): random triangle soup with large overlapping triangles (many spatial splits and
// duplicated references), built with the task pool AND the helper-thread slices forced on small nodes.  tools/sanitize/run.sh compiles this
// with -fsanitize=thread and with -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../gmu-path-tracer_amd/host/sbvh_builder.hpp"

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 20000;
    std::vector<float> v; std::vector<int32_t> idx;
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (int t = 0; t < n; t++) {
        const float c[3] = { rnd() * 8 - 4, rnd() * 8 - 4, rnd() * 8 - 4 }, size = (t % 7 == 0) ? 3.0f : 0.3f;
        for (int k = 0; k < 3; k++) { for (int a = 0; a < 3; a++) v.push_back(c[a] + (rnd() - 0.5f) * size); idx.push_back(3 * t + k); }
    }
    gmupt_sbvh_params p; p.split_alpha = 1.0e-5f; p.max_depth = 64; p.max_spatial_depth = 48; p.min_leaf_size = 1; p.max_leaf_size = 0x7FFFFFF; p.node_cost = 1.0f; p.tri_cost = 1.0f;
    gmupt::SbvhBuilder b(v.data(), (uint32_t)(v.size() / 3), idx.data(), (uint32_t)n, p);
    b.build();
    std::vector<gmupt_bvh_node> nodes(b.numNodes()); std::vector<gmupt_triangle> tris(b.numReferences()); std::vector<int32_t> ref(b.numReferences());
    b.flatten(nullptr, nodes.data(), tris.data(), ref.data());
    std::printf("nodes %u references %u duplicates %u depth %u sah %.3f\n", b.numNodes(), b.numReferences(), b.numDuplicates(), b.depth(), b.sah());
    return 0;
}
