#include "TileGather.hpp"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace gmupt {

std::pair<uint32_t, uint32_t> rowBand(uint32_t height, uint32_t ranks, uint32_t rank)
{
	if (ranks == 0 || rank >= ranks) throw std::invalid_argument("rowBand: rank out of range");
	const uint32_t base = height / ranks, extra = height % ranks;
	const uint32_t first = rank * base + (rank < extra ? rank : extra);
	return { first, base + (rank < extra ? 1u : 0u) };
}

namespace {
void hipCheck(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
void ncclCheck(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
}

struct TileGather::Impl
{
	uint32_t rank = 0, ranks = 1;
	int device = 0;
	ncclComm_t comm = nullptr;
	hipStream_t stream = nullptr;
	void* frame = nullptr; size_t frameBytes = 0;      // rank 0: the assembled frame in device memory (the bands arrive in place)
};

TileGather::TileGather(uint32_t rank, uint32_t ranks, int hipDevice, const std::string& rendezvousFile, double timeoutSeconds)
	: m(new Impl())
{
	if (ranks == 0 || rank >= ranks) throw std::invalid_argument("TileGather: rank out of range");
	m->rank = rank; m->ranks = ranks; m->device = hipDevice;
	hipCheck(hipSetDevice(hipDevice), "hipSetDevice");
	ncclUniqueId id;
	if (rank == 0)
	{
		ncclCheck(ncclGetUniqueId(&id), "ncclGetUniqueId");
		const std::string tmp = rendezvousFile + ".tmp";
		std::FILE* f = std::fopen(tmp.c_str(), "wb");
		if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) { if (f) std::fclose(f); throw std::runtime_error("TileGather: cannot write " + tmp); }
		std::fclose(f);
		if (std::rename(tmp.c_str(), rendezvousFile.c_str()) != 0) throw std::runtime_error("TileGather: cannot publish " + rendezvousFile);
	}
	else
	{
		const auto start = std::chrono::steady_clock::now();
		for (;;)
		{
			std::FILE* f = std::fopen(rendezvousFile.c_str(), "rb");
			if (f)
			{
				const size_t n = std::fread(&id, sizeof(id), 1, f);
				std::fclose(f);
				if (n == 1) break;
			}
			if (std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() > timeoutSeconds)
				throw std::runtime_error("TileGather: rank 0 did not publish " + rendezvousFile);
			std::this_thread::sleep_for(std::chrono::milliseconds(20));
		}
	}
	ncclCheck(ncclCommInitRank(&m->comm, static_cast<int>(ranks), id, static_cast<int>(rank)), "ncclCommInitRank");
	hipCheck(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking), "hipStreamCreate");
}

TileGather::~TileGather()
{
	if (m->frame) (void)hipFree(m->frame);
	if (m->stream) (void)hipStreamDestroy(m->stream);
	if (m->comm) (void)ncclCommDestroy(m->comm);
}

uint32_t TileGather::rank() const { return m->rank; }
uint32_t TileGather::ranks() const { return m->ranks; }

std::vector<float> TileGather::gatherToRoot(const void* deviceBand, uint32_t width, uint32_t height)
{
	hipCheck(hipSetDevice(m->device), "hipSetDevice");
	const size_t rowFloats = static_cast<size_t>(width) * 4;
	const auto mine = rowBand(height, m->ranks, m->rank);
	if (m->rank != 0)
	{
		ncclCheck(ncclGroupStart(), "ncclGroupStart");
		if (mine.second) ncclCheck(ncclSend(deviceBand, rowFloats * mine.second, ncclFloat, 0, m->comm, m->stream), "ncclSend");
		ncclCheck(ncclGroupEnd(), "ncclGroupEnd");
		hipCheck(hipStreamSynchronize(m->stream), "hipStreamSynchronize");
		return {};
	}
	const size_t bytes = rowFloats * height * sizeof(float);
	if (bytes > m->frameBytes)
	{
		if (m->frame) hipCheck(hipFree(m->frame), "hipFree");
		m->frame = nullptr; m->frameBytes = 0;
		hipCheck(hipMalloc(&m->frame, bytes), "hipMalloc (assembled frame)");
		m->frameBytes = bytes;
	}
	float* frame = static_cast<float*>(m->frame);
	// the bands land where they belong: no staging copy, no padding (send / receive sizes are per pair)
	if (mine.second) hipCheck(hipMemcpyAsync(frame + rowFloats * mine.first, deviceBand, rowFloats * mine.second * sizeof(float), hipMemcpyDeviceToDevice, m->stream), "hipMemcpyAsync (own band)");
	ncclCheck(ncclGroupStart(), "ncclGroupStart");
	for (uint32_t r = 1; r < m->ranks; r++)
	{
		const auto b = rowBand(height, m->ranks, r);
		if (b.second) ncclCheck(ncclRecv(frame + rowFloats * b.first, rowFloats * b.second, ncclFloat, static_cast<int>(r), m->comm, m->stream), "ncclRecv");
	}
	ncclCheck(ncclGroupEnd(), "ncclGroupEnd");
	std::vector<float> host(rowFloats * height);
	hipCheck(hipMemcpyAsync(host.data(), frame, bytes, hipMemcpyDeviceToHost, m->stream), "hipMemcpyAsync (read back)");
	hipCheck(hipStreamSynchronize(m->stream), "hipStreamSynchronize");
	return host;
}

} // namespace gmupt
