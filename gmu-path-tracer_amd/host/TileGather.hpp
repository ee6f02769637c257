// Multi-GPU for the C++ host: the frame is split into horizontal row bands, one per rank (one process per GPU, each with its own
// Renderer on its own device; the ranks share nothing while they render), and at accumulate time the bands travel to rank 0 -- the only
// exchange of the path (the Python host does the same with torch.distributed, tiles.py).
// xGMI is point to point, so the gather is what it looks like: every rank sends its band straight to rank 0 (ncclSend / ncclRecv inside
// one group: RCCL routes each pair over its own xGMI link), no ring, no reduction.  The reference is single-adapter (Renderer.cpp:243).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace gmupt {

// rows [first, first + count) of rank `rank`: contiguous bands whose sizes differ by at most one row (== tiles.py: row_bands)
std::pair<uint32_t, uint32_t> rowBand(uint32_t height, uint32_t ranks, uint32_t rank);

class TileGather
{
public:
	// Every rank constructs one with the same `ranks` and `rendezvousFile` (a path all ranks can reach: rank 0 writes the RCCL unique id
	// there -- under a temporary name, then renames it --, the others wait for it) after selecting its device (the Renderer's).
	TileGather(uint32_t rank, uint32_t ranks, int hipDevice, const std::string& rendezvousFile, double timeoutSeconds = 120.0);
	~TileGather();
	TileGather(const TileGather&) = delete;
	TileGather& operator=(const TileGather&) = delete;

	uint32_t rank() const;
	uint32_t ranks() const;
	// deviceBand: this rank's band, rowBand(height, ranks, rank).second rows of `width` RGBA32F texels in device memory.
	// Returns the assembled frame (height x width x 4 floats, host memory) on rank 0, an empty vector elsewhere.  Collective: all ranks call it.
	std::vector<float> gatherToRoot(const void* deviceBand, uint32_t width, uint32_t height);

private:
	struct Impl;
	std::unique_ptr<Impl> m;
};

} // namespace gmupt
