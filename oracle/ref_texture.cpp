// oracle/ref_texture.cpp -- TEST INFRASTRUCTURE ONLY.  Thin C entry points around the two portable libraries the reference vendors IN SOURCE
// for its texture path and calls at Source/Scene.cpp:226 (lodepng::decode) and :269-279 (avir::CImageResizer<fpclass_float8_dil>(8)):
//   /root/reference/Include/lodepng/lodepng.{h,cpp}   (compiled where it lies, see oracle/Makefile target _ref/libreftex.so)
//   /root/reference/Include/avir/avir.h, avir_float8_avx.h (header-only)
// Nothing of them is copied into this repository; the library is built in the build container only (the reference tree does not travel) and is
// used by tools/make_texture_golden.py to produce tests/golden/texture_ref.npz and by tests/test_textures_cpu.py to re-check that fixture.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lodepng/lodepng.h"
#include "avir/avir.h"
#include "avir/avir_float8_avx.h"

extern "C" {

// lodepng::decode(out, w, h, file) with the defaults the reference uses (RGBA, 8 bit).  Returns lodepng's error code (0 = ok);
// *rgba is malloc'ed, release it with ref_free.
unsigned ref_lodepng_decode(const unsigned char* png, size_t bytes, unsigned* width, unsigned* height, unsigned char** rgba)
{
    std::vector<unsigned char> out;
    const unsigned err = lodepng::decode(out, *width, *height, png, bytes);
    *rgba = nullptr;
    if (err) return err;
    *rgba = static_cast<unsigned char*>(std::malloc(out.size() ? out.size() : 1));
    std::memcpy(*rgba, out.data(), out.size());
    return 0;
}

void ref_free(unsigned char* p) { std::free(p); }

// exactly the call of Scene::createTextures: resizeImage(src, old, old, 0, dst, dim, dim, 4, 0) on a resizer constructed with (8)
void ref_avir_resize_square(const unsigned char* rgba, unsigned old_dim, unsigned char* dst, unsigned new_dim)
{
    avir::CImageResizer<avir::fpclass_float8_dil> resizer(8);
    resizer.resizeImage(rgba, (int)old_dim, (int)old_dim, 0, dst, (int)new_dim, (int)new_dim, 4, 0);
}

}
