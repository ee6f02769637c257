#include "TextureLoader.hpp"
#include "png_reader.hpp"
#include <cmath>
#include <set>
#include <stdexcept>

namespace gmupt {

unsigned commonDimension(const std::vector<size_t>& layerBytes)
{
	std::set<size_t> distinct(layerBytes.begin(), layerBytes.end());
	if (distinct.empty()) return 0;
	auto it = distinct.begin();
	std::advance(it, static_cast<long>(distinct.size() / 2));
	return static_cast<unsigned>(std::sqrt(static_cast<double>(*it / 4)));
}

TextureSet loadSpecificTexture(const std::vector<std::vector<uint8_t>>& encoded, std::vector<gmupt_material>& materials, int index)
{
	TextureSet set;
	std::vector<size_t> sizes;
	for (size_t i = 0; i < encoded.size() && i < materials.size(); i++)
	{
		if (encoded[i].empty()) continue;
		png::Image img = png::decode(encoded[i].data(), encoded[i].size());
		if (img.width != img.height) throw std::runtime_error("texture of material " + std::to_string(i) + " is not square (the reference asserts width == height)");
		materials[i].textureIndices[index] = static_cast<int32_t>(set.layers.size());
		sizes.push_back(img.rgba.size());
		set.layers.emplace_back(std::move(img.rgba));
	}
	set.dimension = commonDimension(sizes);
	const size_t want = static_cast<size_t>(set.dimension) * set.dimension * 4;
	for (auto& layer : set.layers)
		if (layer.size() != want)
		{
			const unsigned old = static_cast<unsigned>(std::sqrt(static_cast<double>(layer.size() >> 2)));
			layer = resizeSquare(layer.data(), old, set.dimension);
		}
	return set;
}

} // namespace gmupt
