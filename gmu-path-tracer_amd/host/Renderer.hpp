// Same public surface as the reference's Renderer (Include/Renderer.hpp:9-68): Renderer(HWND, Resolution), update(dt),
// draw().  The D3D11 device / swap chain / UAVs are replaced by C-ABI handles (include/gmupt.h); HWND becomes an opaque,
// nullable window handle (this build is headless).  render() = update(dt); draw(); is the alias BASELINE.json names.
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "Scene.hpp"

class Renderer
{
	using Resolution = std::pair<unsigned, unsigned>;
public:
	// multi-GPU extension (one process per GPU, TileGather.hpp): this renderer generates paths for the rows [y0, y0 + rows) of the frame only
	// and accumulates into a target of that size; `resolution` and the camera stay those of the whole frame.  rows == 0: the whole frame.
	struct RowBand { unsigned y0, rows; };

	Renderer(void* hwnd, Resolution resolution, const std::string& scene = "cornell", int hipDevice = 0,
	         unsigned poolPaths = PATHCOUNT, unsigned livePaths = REFERENCE_LIVE_PATHS, RowBand band = RowBand{ 0, 0 });
	~Renderer();

	void update(float dt);
	void draw();
	void render(float dt = 0.f) { update(dt); draw(); }

	// requests the reference takes from keyboard / window events (Source/Renderer.cpp:146-156)
	void requestResize(const Resolution& resolution) { mPendingResize = resolution; mHasResize = true; }
	void requestCapture() { mCaptureRequested = true; }
	void initScene(const std::string& name);

	// headless extras
	std::vector<float> readFramebuffer();            // RGBA32F, a = sample count bits (the row band when one was given)
	void copyFramebufferToDevice(void* deviceDst);   // the same texels into caller-owned device memory (the tile gather's source), synchronised
	Resolution targetSize() const { return { mResolution.first, mBand.rows ? mBand.rows : mResolution.second }; }
	void writePfm(const std::string& path);          // raw RGB float export ("PF", little-endian, bottom row first) for image comparisons
	std::string lastCapturePath() const { return mLastCapture; }
	Scene& scene() { return mScene; }
	unsigned long long iterations() const { return mIterations; }

private:
	void createDevice(int hipDevice);
	void createBuffers(Resolution res);
	void captureScreen();
	void resize(const Resolution& resolution);

	struct DeviceDeleter { void operator()(gmupt_device* d) const { gmupt_device_destroy(d); } };
	struct RendererDeleter { void operator()(gmupt_renderer* r) const { gmupt_renderer_destroy(r); } };

	void* mHwnd;
	std::unique_ptr<gmupt_device, DeviceDeleter> mDevice;
	std::unique_ptr<gmupt_renderer, RendererDeleter> mRenderer; // path state, queues, counters, accumulation target
	Scene mScene;
	Resolution mResolution;
	RowBand mBand;
	unsigned mPoolPaths, mLivePaths;
	bool mSceneBound = false;
	bool mHasResize = false, mCaptureRequested = false;
	Resolution mPendingResize{};
	std::string mLastCapture;
	unsigned long long mIterations = 0;
};
