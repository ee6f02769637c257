#include "TextureLoader.hpp"
#include "png_reader.hpp"
#include <algorithm>
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

namespace {
double lanczos3(double x)
{
	if (x == 0.0) return 1.0;
	if (x <= -3.0 || x >= 3.0) return 0.0;
	const double px = 3.14159265358979323846 * x;
	return 3.0 * std::sin(px) * std::sin(px / 3.0) / (px * px);
}

struct Taps { int first; std::vector<float> weight; };

std::vector<Taps> makeTaps(unsigned from, unsigned to)
{
	const double scale = static_cast<double>(from) / static_cast<double>(to);
	const double stretch = scale > 1.0 ? scale : 1.0;
	std::vector<Taps> taps(to);
	for (unsigned i = 0; i < to; i++)
	{
		// geometry of avir's default resizing step (Include/avir/avir.h:4301-4324, k == 0): when ENLARGING, the centres of the first and the
		// last pixel of source and destination coincide (step (from - 1) / (to - 1), no offset); when reducing, the images are aligned by
		// their outer edges (step from / to, offset (step - 1) / 2)
		const double center = to > from ? (to > 1 ? static_cast<double>(i) * (static_cast<double>(from) - 1.0) / (static_cast<double>(to) - 1.0) : 0.0)
		                                : (static_cast<double>(i) + 0.5) * scale - 0.5;
		const int lo = static_cast<int>(std::floor(center - 3.0 * stretch)) + 1, hi = static_cast<int>(std::floor(center + 3.0 * stretch));
		Taps& t = taps[i]; t.first = lo;
		double sum = 0.0;
		std::vector<double> w;
		for (int k = lo; k <= hi; k++) { const double v = lanczos3((static_cast<double>(k) - center) / stretch); w.push_back(v); sum += v; }
		for (double v : w) t.weight.push_back(static_cast<float>(v / sum));
	}
	return taps;
}
}

std::vector<uint8_t> resizeSquare(const uint8_t* rgba, unsigned from, unsigned to)
{
	if (from == 0 || to == 0) throw std::runtime_error("resizeSquare: empty image");
	if (from == to) return std::vector<uint8_t>(rgba, rgba + static_cast<size_t>(from) * from * 4);
	const std::vector<Taps> taps = makeTaps(from, to);
	const int last = static_cast<int>(from) - 1;
	// horizontal pass: from x from -> to x from (float)
	std::vector<float> mid(static_cast<size_t>(to) * from * 4);
	for (unsigned y = 0; y < from; y++)
		for (unsigned x = 0; x < to; x++)
		{
			float acc[4] = { 0.f, 0.f, 0.f, 0.f };
			const Taps& t = taps[x];
			for (size_t k = 0; k < t.weight.size(); k++)
			{
				const int sx = std::min(std::max(t.first + static_cast<int>(k), 0), last);
				const uint8_t* s = rgba + (static_cast<size_t>(y) * from + static_cast<size_t>(sx)) * 4;
				for (int c = 0; c < 4; c++) acc[c] += t.weight[k] * static_cast<float>(s[c]);
			}
			for (int c = 0; c < 4; c++) mid[(static_cast<size_t>(y) * to + x) * 4 + static_cast<size_t>(c)] = acc[c];
		}
	// vertical pass: to x from -> to x to
	std::vector<uint8_t> out(static_cast<size_t>(to) * to * 4);
	for (unsigned y = 0; y < to; y++)
	{
		const Taps& t = taps[y];
		for (unsigned x = 0; x < to; x++)
		{
			float acc[4] = { 0.f, 0.f, 0.f, 0.f };
			for (size_t k = 0; k < t.weight.size(); k++)
			{
				const int sy = std::min(std::max(t.first + static_cast<int>(k), 0), last);
				const float* s = &mid[(static_cast<size_t>(sy) * to + x) * 4];
				for (int c = 0; c < 4; c++) acc[c] += t.weight[k] * s[c];
			}
			for (int c = 0; c < 4; c++)
			{
				const float v = std::floor(acc[c] + 0.5f);
				out[(static_cast<size_t>(y) * to + x) * 4 + static_cast<size_t>(c)] = static_cast<uint8_t>(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
			}
		}
	}
	return out;
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
