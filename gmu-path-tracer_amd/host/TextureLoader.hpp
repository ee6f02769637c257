// Texture ingestion of the host side: what Scene::loadSpecificTexture + Scene::createTextures do before the D3D11 upload
// (reference Source/Scene.cpp:209-244 and :246-290): decode every material's PNG of one texture type to RGBA8, give it the next layer
// index, pick the common square size by the reference's "median of the distinct byte sizes" rule and resize the other layers to it.
// The reference resizes with the avir library; resizeSquare (AvirResize.cpp) restates avir's pipeline for that call and returns the same
// bytes (tests/golden/texture_ref.npz, tests/test_textures_cpu.py::test_resize_against_avir_fixture).  Layers that already have the common
// size are not resized, as in the reference (Scene.cpp:268).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/gmupt.h"

namespace gmupt {

struct TextureSet                                  // Scene::LoadedTextures (Include/Scene.hpp:98-99)
{
	std::vector<std::vector<uint8_t>> layers;      // RGBA8, dimension x dimension each after finalize()
	unsigned dimension = 0;
};

// Scene.cpp:232-241: the distinct layer byte sizes in ascending order, the element at position count/2, width = sqrt(bytes / 4)
unsigned commonDimension(const std::vector<size_t>& layerBytes);

// square RGBA8 resize with the bytes of avir::CImageResizer<fpclass_float8_dil>(8).resizeImage(src, n, n, 0, dst, m, m, 4, 0)
// (Scene.cpp:276-279); defined in AvirResize.cpp
std::vector<uint8_t> resizeSquare(const uint8_t* rgba, unsigned oldDimension, unsigned newDimension);

// encoded[i] = the PNG file of material i for this texture type (empty: none).  Sets materials[i].textureIndices[index] to the layer
// number (order of appearance, Scene.cpp:221) and returns the decoded layers, all resized to the common dimension (Scene.cpp:268-285).
// Throws std::runtime_error for undecodable or non-square images (the reference asserts width == height, Scene.cpp:230).
TextureSet loadSpecificTexture(const std::vector<std::vector<uint8_t>>& encoded, std::vector<gmupt_material>& materials, int index);

} // namespace gmupt
