// PNG decoder to 8-bit RGBA: what the reference gets from lodepng::decode(out, w, h, file) with its default settings
// (Source/Scene.cpp:226: LCT_RGBA, bit depth 8).  Self-contained: inflate (RFC 1951), zlib framing (RFC 1950), chunk CRCs,
// the five scan-line filters, Adam7, all colour types / bit depths of the PNG specification, tRNS.  16-bit samples keep their
// high byte, samples below 8 bits are scaled by 255 / (2^depth - 1) -- the conversion rules lodepng documents for LCT_RGBA/8.
// Written from the specifications; no code of lodepng is used.  Errors throw std::runtime_error.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace gmupt {
namespace png {

inline void fail(const char* what) { throw std::runtime_error(std::string("PNG: ") + what); }

// ------------------------------------------------------------------------------------------------ inflate
class Inflater
{
public:
	Inflater(const uint8_t* data, size_t size) : mData(data), mSize(size) {}

	std::vector<uint8_t> run()
	{
		std::vector<uint8_t> out;
		bool last = false;
		while (!last)
		{
			last = bits(1) != 0;
			const uint32_t type = bits(2);
			if (type == 0) stored(out);
			else if (type == 1) { fixedTables(); block(out); }
			else if (type == 2) { dynamicTables(); block(out); }
			else fail("reserved deflate block type");
		}
		return out;
	}
	size_t bytesConsumed() const { return mPos - static_cast<size_t>(mBitCount / 8); }

private:
	struct Table { uint16_t count[16]; uint16_t symbol[288]; };

	uint32_t bits(int need)
	{
		while (mBitCount < need)
		{
			if (mPos >= mSize) fail("deflate stream ends early");
			mBitBuf |= static_cast<uint32_t>(mData[mPos++]) << mBitCount;
			mBitCount += 8;
		}
		const uint32_t v = mBitBuf & ((need == 32) ? 0xFFFFFFFFu : ((1u << need) - 1u));
		mBitBuf >>= need; mBitCount -= need;
		return v;
	}

	static void build(Table& t, const uint8_t* lengths, int n)
	{
		std::memset(t.count, 0, sizeof(t.count));
		for (int i = 0; i < n; i++) t.count[lengths[i]]++;
		t.count[0] = 0;
		int left = 1;
		for (int len = 1; len < 16; len++) { left = (left << 1) - t.count[len]; if (left < 0) fail("over-subscribed Huffman code"); }
		uint16_t offs[16]; offs[1] = 0;
		for (int len = 1; len < 15; len++) offs[len + 1] = static_cast<uint16_t>(offs[len] + t.count[len]);
		for (int i = 0; i < n; i++) if (lengths[i]) t.symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
	}

	int decode(const Table& t)
	{
		int code = 0, first = 0, index = 0;
		for (int len = 1; len < 16; len++)
		{
			code |= static_cast<int>(bits(1));
			const int count = t.count[len];
			if (code - count < first) return t.symbol[index + (code - first)];
			index += count; first += count; first <<= 1; code <<= 1;
		}
		fail("invalid Huffman code");
		return -1;
	}

	void stored(std::vector<uint8_t>& out)
	{
		mBitBuf = 0; mBitCount = 0; // skip to the byte boundary
		if (mPos + 4 > mSize) fail("stored block header truncated");
		const uint32_t len = mData[mPos] | (mData[mPos + 1] << 8), nlen = mData[mPos + 2] | (mData[mPos + 3] << 8);
		mPos += 4;
		if ((len ^ 0xFFFFu) != nlen) fail("stored block length check");
		if (mPos + len > mSize) fail("stored block truncated");
		out.insert(out.end(), mData + mPos, mData + mPos + len);
		mPos += len;
	}

	void fixedTables()
	{
		uint8_t lengths[288];
		for (int i = 0; i < 144; i++) lengths[i] = 8;
		for (int i = 144; i < 256; i++) lengths[i] = 9;
		for (int i = 256; i < 280; i++) lengths[i] = 7;
		for (int i = 280; i < 288; i++) lengths[i] = 8;
		build(mLit, lengths, 288);
		for (int i = 0; i < 30; i++) lengths[i] = 5;
		build(mDist, lengths, 30);
	}

	void dynamicTables()
	{
		static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		const int nlen = static_cast<int>(bits(5)) + 257, ndist = static_cast<int>(bits(5)) + 1, ncode = static_cast<int>(bits(4)) + 4;
		if (nlen > 286 || ndist > 30) fail("bad dynamic table sizes");
		uint8_t lengths[320] = {};
		for (int i = 0; i < ncode; i++) lengths[order[i]] = static_cast<uint8_t>(bits(3));
		Table codeLengths;
		build(codeLengths, lengths, 19);
		std::memset(lengths, 0, sizeof(lengths));
		int index = 0;
		while (index < nlen + ndist)
		{
			const int sym = decode(codeLengths);
			if (sym < 16) lengths[index++] = static_cast<uint8_t>(sym);
			else
			{
				uint8_t value = 0; int repeat;
				if (sym == 16) { if (index == 0) fail("repeat without a previous length"); value = lengths[index - 1]; repeat = 3 + static_cast<int>(bits(2)); }
				else if (sym == 17) repeat = 3 + static_cast<int>(bits(3));
				else repeat = 11 + static_cast<int>(bits(7));
				if (index + repeat > nlen + ndist) fail("code length repeat overruns the table");
				while (repeat--) lengths[index++] = value;
			}
		}
		if (lengths[256] == 0) fail("no end-of-block code");
		build(mLit, lengths, nlen);
		build(mDist, lengths + nlen, ndist);
	}

	void block(std::vector<uint8_t>& out)
	{
		static const uint16_t lenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
		static const uint8_t lenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
		static const uint16_t distBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
		static const uint8_t distExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
		for (;;)
		{
			int sym = decode(mLit);
			if (sym < 256) { out.push_back(static_cast<uint8_t>(sym)); continue; }
			if (sym == 256) return;
			sym -= 257;
			if (sym >= 29) fail("invalid length symbol");
			const size_t len = lenBase[sym] + bits(lenExtra[sym]);
			const int dsym = decode(mDist);
			if (dsym >= 30) fail("invalid distance symbol");
			const size_t dist = distBase[dsym] + bits(distExtra[dsym]);
			if (dist > out.size()) fail("distance reaches before the start of the output");
			const size_t from = out.size() - dist;
			for (size_t i = 0; i < len; i++) out.push_back(out[from + i]); // overlapping copies replicate, as deflate requires
		}
	}

	const uint8_t* mData; size_t mSize; size_t mPos = 0;
	uint32_t mBitBuf = 0; int mBitCount = 0;
	Table mLit, mDist;
};

inline std::vector<uint8_t> zlibDecompress(const uint8_t* data, size_t size)
{
	if (size < 6) fail("zlib stream too short");
	const uint32_t cmf = data[0], flg = data[1];
	if ((cmf & 15u) != 8u || (cmf >> 4) > 7u) fail("zlib: not deflate with a window of at most 32 KiB");
	if (((cmf << 8) | flg) % 31u != 0u) fail("zlib header check");
	if (flg & 32u) fail("zlib preset dictionary");
	Inflater inf(data + 2, size - 2);
	std::vector<uint8_t> out = inf.run();
	const size_t tail = 2 + inf.bytesConsumed();
	if (tail + 4 > size) fail("zlib checksum missing");
	uint32_t a = 1, b = 0;
	for (uint8_t v : out) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
	const uint32_t stored = (static_cast<uint32_t>(data[tail]) << 24) | (data[tail + 1] << 16) | (data[tail + 2] << 8) | data[tail + 3];
	if (stored != ((b << 16) | a)) fail("zlib Adler-32 mismatch");
	return out;
}

// ------------------------------------------------------------------------------------------------ PNG container
inline uint32_t crc32(const uint8_t* data, size_t size)
{
	struct Table { uint32_t v[256]; Table() { for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1); v[n] = c; } } };
	static const Table t; // initialised once, thread-safe (the three texture workers decode concurrently)
	const uint32_t* table = t.v;
	uint32_t c = 0xFFFFFFFFu;
	for (size_t i = 0; i < size; i++) c = table[(c ^ data[i]) & 255u] ^ (c >> 8);
	return c ^ 0xFFFFFFFFu;
}

inline uint32_t be32(const uint8_t* p) { return (static_cast<uint32_t>(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

struct Header { uint32_t width = 0, height = 0; int depth = 0, colorType = 0, interlace = 0; };

inline int channelsOf(int colorType)
{
	switch (colorType) { case 0: return 1; case 2: return 3; case 3: return 1; case 4: return 2; case 6: return 4; }
	fail("invalid colour type");
	return 0;
}

// reverses the scan-line filters of one (sub)image in place; `raw` holds height * (1 + rowBytes) bytes; returns packed rows
inline void unfilter(const uint8_t* raw, uint8_t* rows, uint32_t height, size_t rowBytes, size_t bpp)
{
	for (uint32_t y = 0; y < height; y++)
	{
		const uint8_t type = raw[y * (rowBytes + 1)];
		const uint8_t* in = raw + y * (rowBytes + 1) + 1;
		uint8_t* cur = rows + y * rowBytes;
		const uint8_t* up = y ? rows + (y - 1) * rowBytes : nullptr;
		for (size_t i = 0; i < rowBytes; i++)
		{
			const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
			int pred = 0;
			switch (type)
			{
			case 0: pred = 0; break;
			case 1: pred = a; break;
			case 2: pred = b; break;
			case 3: pred = (a + b) >> 1; break;
			case 4: { const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p; pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
			default: fail("invalid filter type");
			}
			cur[i] = static_cast<uint8_t>(in[i] + pred);
		}
	}
}

struct Image { uint32_t width = 0, height = 0; std::vector<uint8_t> rgba; };

inline Image decode(const uint8_t* data, size_t size)
{
	static const uint8_t signature[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
	if (size < 8 || std::memcmp(data, signature, 8) != 0) fail("bad signature");
	Header h; bool haveHeader = false, ended = false;
	std::vector<uint8_t> idat, palette, trns;
	size_t pos = 8;
	while (!ended)
	{
		if (pos + 12 > size) fail("chunk truncated");
		const uint32_t len = be32(data + pos);
		if (len > 0x7FFFFFFFu || pos + 12 + static_cast<size_t>(len) > size) fail("chunk length exceeds the file");
		const uint8_t* type = data + pos + 4; const uint8_t* body = data + pos + 8;
		if (crc32(type, 4 + static_cast<size_t>(len)) != be32(body + len)) fail("chunk CRC mismatch");
		if (!haveHeader && std::memcmp(type, "IHDR", 4) != 0) fail("first chunk is not IHDR");
		if (std::memcmp(type, "IHDR", 4) == 0)
		{
			if (len != 13) fail("IHDR size");
			h.width = be32(body); h.height = be32(body + 4); h.depth = body[8]; h.colorType = body[9]; h.interlace = body[12];
			if (h.width == 0 || h.height == 0 || h.width > 32768 || h.height > 32768) fail("unsupported image size");
			if (body[10] != 0 || body[11] != 0 || h.interlace > 1) fail("unknown compression / filter / interlace method");
			const int ch = channelsOf(h.colorType); (void)ch;
			const bool depthOk = (h.colorType == 0) ? (h.depth == 1 || h.depth == 2 || h.depth == 4 || h.depth == 8 || h.depth == 16)
				: (h.colorType == 3) ? (h.depth == 1 || h.depth == 2 || h.depth == 4 || h.depth == 8) : (h.depth == 8 || h.depth == 16);
			if (!depthOk) fail("bit depth not allowed for the colour type");
			haveHeader = true;
		}
		else if (std::memcmp(type, "PLTE", 4) == 0) { if (len % 3 != 0 || len > 768) fail("PLTE size"); palette.assign(body, body + len); }
		else if (std::memcmp(type, "tRNS", 4) == 0) trns.assign(body, body + len);
		else if (std::memcmp(type, "IDAT", 4) == 0) idat.insert(idat.end(), body, body + len);
		else if (std::memcmp(type, "IEND", 4) == 0) ended = true;
		else if ((type[0] & 32u) == 0) fail("unknown critical chunk");
		pos += 12 + static_cast<size_t>(len);
	}
	if (idat.empty()) fail("no image data");
	if (h.colorType == 3 && palette.empty()) fail("palette image without PLTE");

	const int channels = channelsOf(h.colorType);
	const size_t bitsPerPixel = static_cast<size_t>(channels) * static_cast<size_t>(h.depth);
	const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
	const std::vector<uint8_t> raw = zlibDecompress(idat.data(), idat.size());

	// sample (x, y, channel) of a packed row set
	auto sampleAt = [&](const uint8_t* row, uint32_t x, int c) -> uint32_t {
		if (h.depth == 8) return row[static_cast<size_t>(x) * channels + c];
		if (h.depth == 16) { const uint8_t* p = row + (static_cast<size_t>(x) * channels + c) * 2; return (static_cast<uint32_t>(p[0]) << 8) | p[1]; }
		const size_t bit = static_cast<size_t>(x) * h.depth; // one channel below 8 bits
		return (row[bit >> 3] >> (8 - h.depth - static_cast<int>(bit & 7))) & ((1u << h.depth) - 1u);
	};
	auto to8 = [&](uint32_t v) -> uint8_t {
		if (h.depth == 8) return static_cast<uint8_t>(v);
		if (h.depth == 16) return static_cast<uint8_t>(v >> 8);
		return static_cast<uint8_t>((v * 255u) / ((1u << h.depth) - 1u));
	};
	auto trnsKey = [&](int c) -> uint32_t { return (static_cast<uint32_t>(trns[static_cast<size_t>(c) * 2]) << 8) | trns[static_cast<size_t>(c) * 2 + 1]; };

	Image img; img.width = h.width; img.height = h.height;
	img.rgba.assign(static_cast<size_t>(h.width) * h.height * 4, 0);
	auto emit = [&](const uint8_t* row, uint32_t sx, uint32_t dx, uint32_t dy) {
		uint8_t* o = &img.rgba[(static_cast<size_t>(dy) * h.width + dx) * 4];
		switch (h.colorType)
		{
		case 0: { const uint32_t g = sampleAt(row, sx, 0); o[0] = o[1] = o[2] = to8(g); o[3] = (trns.size() >= 2 && g == trnsKey(0)) ? 0 : 255; break; }
		case 2: { const uint32_t r = sampleAt(row, sx, 0), g = sampleAt(row, sx, 1), b = sampleAt(row, sx, 2); o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
		          o[3] = (trns.size() >= 6 && r == trnsKey(0) && g == trnsKey(1) && b == trnsKey(2)) ? 0 : 255; break; }
		case 3: { const uint32_t i = sampleAt(row, sx, 0); if (static_cast<size_t>(i) * 3 + 2 >= palette.size()) fail("palette index out of range");
		          o[0] = palette[i * 3]; o[1] = palette[i * 3 + 1]; o[2] = palette[i * 3 + 2]; o[3] = i < trns.size() ? trns[i] : 255; break; }
		case 4: { o[0] = o[1] = o[2] = to8(sampleAt(row, sx, 0)); o[3] = to8(sampleAt(row, sx, 1)); break; }
		default: { for (int c = 0; c < 4; c++) o[c] = to8(sampleAt(row, sx, c)); break; }
		}
	};

	if (h.interlace == 0)
	{
		const size_t rowBytes = (static_cast<size_t>(h.width) * bitsPerPixel + 7) / 8;
		if (raw.size() < (rowBytes + 1) * h.height) fail("image data too short");
		std::vector<uint8_t> rows(rowBytes * h.height);
		unfilter(raw.data(), rows.data(), h.height, rowBytes, bpp);
		for (uint32_t y = 0; y < h.height; y++) for (uint32_t x = 0; x < h.width; x++) emit(rows.data() + y * rowBytes, x, x, y);
	}
	else
	{
		static const uint32_t x0[7] = { 0, 4, 0, 2, 0, 1, 0 }, y0[7] = { 0, 0, 4, 0, 2, 0, 1 }, dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
		size_t offset = 0;
		for (int pass = 0; pass < 7; pass++)
		{
			const uint32_t pw = (h.width + dx[pass] - 1 - x0[pass]) / dx[pass], ph = (h.height + dy[pass] - 1 - y0[pass]) / dy[pass];
			if (h.width <= x0[pass] || h.height <= y0[pass] || pw == 0 || ph == 0) continue;
			const size_t rowBytes = (static_cast<size_t>(pw) * bitsPerPixel + 7) / 8;
			if (raw.size() < offset + (rowBytes + 1) * ph) fail("interlaced image data too short");
			std::vector<uint8_t> rows(rowBytes * ph);
			unfilter(raw.data() + offset, rows.data(), ph, rowBytes, bpp);
			for (uint32_t y = 0; y < ph; y++) for (uint32_t x = 0; x < pw; x++) emit(rows.data() + y * rowBytes, x, x0[pass] + x * dx[pass], y0[pass] + y * dy[pass]);
			offset += (rowBytes + 1) * ph;
		}
	}
	return img;
}

} // namespace png
} // namespace gmupt
