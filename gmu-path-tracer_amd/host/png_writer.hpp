// Tiny dependency-free PNG writer (RGBA8, zlib "stored" blocks) standing in for lodepng::encode in captureScreen.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace gmupt {

inline uint32_t crc32Update(uint32_t crc, const unsigned char* d, size_t n)
{
	static uint32_t table[256]; static bool init = false;
	if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
	for (size_t i = 0; i < n; i++) crc = table[(crc ^ d[i]) & 0xFF] ^ (crc >> 8);
	return crc;
}

inline bool writePngRGBA8(const std::string& path, const unsigned char* rgba, unsigned w, unsigned h)
{
	std::vector<unsigned char> raw; raw.reserve((size_t)h * (w * 4 + 1));
	for (unsigned y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4); }
	std::vector<unsigned char> z = { 0x78, 0x01 };
	uint32_t a = 1, b = 0;
	for (unsigned char c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
	for (size_t pos = 0; pos < raw.size() || pos == 0;) {
		const size_t n = std::min<size_t>(65535, raw.size() - pos);
		const bool last = pos + n >= raw.size();
		z.push_back(last ? 1 : 0);
		z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
		z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
		pos += n;
		if (last) break;
	}
	const uint32_t adler = (b << 16) | a;
	for (int s = 24; s >= 0; s -= 8) z.push_back((adler >> s) & 0xFF);

	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) return false;
	auto be32 = [](unsigned char* p, uint32_t v) { p[0] = v >> 24; p[1] = (v >> 16) & 0xFF; p[2] = (v >> 8) & 0xFF; p[3] = v & 0xFF; };
	auto chunk = [&](const char* type, const unsigned char* data, size_t n) {
		unsigned char len[4]; be32(len, (uint32_t)n); std::fwrite(len, 1, 4, f);
		std::fwrite(type, 1, 4, f); if (n) std::fwrite(data, 1, n, f);
		uint32_t crc = crc32Update(0xFFFFFFFFu, reinterpret_cast<const unsigned char*>(type), 4);
		crc = crc32Update(crc, data, n) ^ 0xFFFFFFFFu;
		unsigned char c[4]; be32(c, crc); std::fwrite(c, 1, 4, f);
	};
	const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
	std::fwrite(sig, 1, 8, f);
	unsigned char ihdr[13]; be32(ihdr, w); be32(ihdr + 4, h); ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
	chunk("IHDR", ihdr, 13);
	chunk("IDAT", z.data(), z.size());
	chunk("IEND", nullptr, 0);
	return std::fclose(f) == 0;
}

} // namespace gmupt
