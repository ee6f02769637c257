#include <cstdio>
#include "Renderer.hpp"
#include "png_writer.hpp"
#include <cstdlib>
#include <filesystem>
#include <stdexcept>

namespace fs = std::filesystem;

namespace {
void check(int rc) { if (rc != GMUPT_OK) throw std::runtime_error(gmupt_last_error()); }
}

Renderer::Renderer(void* hwnd, Resolution resolution, const std::string& scene, int hipDevice, unsigned poolPaths, unsigned livePaths, RowBand band)
	: mHwnd(hwnd)
	, mResolution(resolution)
	, mBand(band)
	, mPoolPaths(poolPaths)
	, mLivePaths(livePaths)
{
	createDevice(hipDevice);
	createBuffers(resolution);
	initScene(scene);
	mScene.mCamera.updateResolution(resolution.first, resolution.second);
}

Renderer::~Renderer() = default;

void Renderer::createDevice(int hipDevice)
{
	gmupt_device* d = nullptr;
	check(gmupt_device_create(hipDevice, &d));
	mDevice.reset(d);
}

void Renderer::createBuffers(Resolution res)
{
	gmupt_renderer_desc desc{};
	desc.width = res.first; desc.height = res.second;
	if (mBand.rows)
	{
		if (mBand.y0 + mBand.rows > res.second) throw std::invalid_argument("Renderer: row band outside the frame");
		desc.height = mBand.rows; desc.tile_enabled = 1; desc.tile_x0 = 0; desc.tile_y0 = mBand.y0; // the camera stays the whole frame's
	}
	desc.pool_paths = mPoolPaths; desc.live_paths = mLivePaths;
	gmupt_renderer* r = nullptr;
	check(gmupt_renderer_create(mDevice.get(), &desc, &r));
	mRenderer.reset(r);
}

void Renderer::initScene(const std::string& name)
{
	mScene = Scene(mDevice.get(), name); // old resources die with the temporary (Source/Renderer.cpp:55)
	mScene.mCamera.getBuffer()->lightCount = static_cast<uint32_t>(mScene.lightCount() < 2 ? 2 : mScene.lightCount()); // Camera.hpp:20: never below the default 2
	mSceneBound = false;
}

void Renderer::update(float dt)
{
	if (mHasResize) { resize(mPendingResize); mHasResize = false; }
	if (mCaptureRequested) { captureScreen(); mCaptureRequested = false; }

	mScene.update(dt);
	check(gmupt_set_camera(mRenderer.get(), mScene.mCamera.getBuffer())); // UpdateSubresource(mCameraBuffer), Renderer.cpp:161
}

void Renderer::draw()
{
	if (!mSceneBound)
	{
		check(gmupt_renderer_bind_scene(mRenderer.get(), mScene.mBVHBuffer.get(), mScene.mIndexBuffer.get(), mScene.mVertexBuffer.get(),
		                                mScene.mLightBuffer.get(), mScene.mTriangleProperties.get(), mScene.mMaterialPropertyBuffer.get()));
		// CSSetShaderResources t5..t7 + sampler s0 (Renderer.cpp:173-175,192)
		check(gmupt_renderer_bind_textures(mRenderer.get(), mScene.mDiffuse.get(), mScene.mMetallicRoughness.get(), mScene.mNormal.get()));
		mSceneBound = true;
	}
	check(gmupt_iterate(mRenderer.get())); // logic, newPath, materialUE4, materialGlass, extensionRay, shadowRay (Renderer.cpp:195-211)
	mIterations++;
}

std::vector<float> Renderer::readFramebuffer()
{
	const Resolution t = targetSize();
	std::vector<float> rgba(static_cast<size_t>(t.first) * t.second * 4);
	check(gmupt_read_framebuffer(mRenderer.get(), rgba.data(), rgba.size() * sizeof(float)));
	return rgba;
}

void Renderer::copyFramebufferToDevice(void* deviceDst)
{
	const Resolution t = targetSize();
	check(gmupt_copy_framebuffer_to_device(mRenderer.get(), deviceDst, static_cast<size_t>(t.first) * t.second * 4 * sizeof(float)));
}

void Renderer::writePfm(const std::string& path)
{
	if (mBand.rows) throw std::runtime_error("writePfm: a row band is written by the rank that gathers the frame");
	const auto rgba = readFramebuffer();
	std::FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) throw std::runtime_error("Failed to write " + path);
	std::fprintf(f, "PF\n%u %u\n-1.0\n", mResolution.first, mResolution.second);
	std::vector<float> row(static_cast<size_t>(mResolution.first) * 3);
	for (unsigned y = mResolution.second; y-- > 0;)
	{
		for (unsigned x = 0; x < mResolution.first; x++)
			for (int c = 0; c < 3; c++) row[static_cast<size_t>(x) * 3 + static_cast<size_t>(c)] = rgba[(static_cast<size_t>(y) * mResolution.first + x) * 4 + static_cast<size_t>(c)];
		std::fwrite(row.data(), sizeof(float), row.size(), f);
	}
	std::fclose(f);
}

void Renderer::captureScreen()
{
	if (mBand.rows) throw std::runtime_error("captureScreen: a row band is captured by the rank that gathers the frame");
	// Source/Renderer.cpp:355-406: uint8 = float * 255 (truncation), alpha 255, next free Captures/potatoN.png
	const auto rgba = readFramebuffer();
	std::vector<unsigned char> image(rgba.size());
	for (size_t i = 0; i < rgba.size(); i += 4)
	{
		image[i] = static_cast<unsigned char>(rgba[i] * 255);
		image[i + 1] = static_cast<unsigned char>(rgba[i + 1] * 255);
		image[i + 2] = static_cast<unsigned char>(rgba[i + 2] * 255);
		image[i + 3] = 255;
	}
	auto counter = -1;
	if (fs::exists(CAPTURE_DIR_NAME))
	{
		for (const auto& entry : fs::directory_iterator(CAPTURE_DIR_NAME))
		{
			const auto stem = entry.path().filename().string();
			if (stem.rfind(CAPTURE_NAME, 0) == 0) counter = std::max(counter, std::atoi(stem.substr(std::string(CAPTURE_NAME).size()).c_str()));
		}
	}
	else
		fs::create_directory(CAPTURE_DIR_NAME);
	mLastCapture = std::string(CAPTURE_DIR_NAME) + "/" + CAPTURE_NAME + std::to_string(counter + 1) + ".png";
	if (!gmupt::writePngRGBA8(mLastCapture, image.data(), mResolution.first, mResolution.second))
		throw std::runtime_error("Failed to write " + mLastCapture);
}

void Renderer::resize(const Resolution& resolution)
{
	if (mBand.rows) throw std::runtime_error("resize: the ranks of a tiled render are recreated with their new bands");
	mScene.mCamera.updateResolution(resolution.first, resolution.second); // Renderer.cpp:410
	check(gmupt_resize(mRenderer.get(), resolution.first, resolution.second)); // createRenderTexture, :412
	mResolution = resolution;
}
