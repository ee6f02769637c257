// Sanitizer driver of the host-side scene ingestion (CPU only): glTF / GLB reader, PNG decoder, texture set assembly, .params parser.
// Every argument is a file; a malformed file must end in a std::exception, never in a crash or a sanitizer report.
#include <cstdio>
#include <exception>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
#include "../../gmu-path-tracer_amd/host/MeshData.hpp"
#include "../../gmu-path-tracer_amd/host/TextureLoader.hpp"
#include "../../gmu-path-tracer_amd/host/png_reader.hpp"
#include "../../gmu-path-tracer_amd/host/Scene.hpp"

static bool endsWith(const std::string& s, const char* e) { const std::string t(e); return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }

int main(int argc, char** argv)
{
    int ok = 0, rejected = 0;
    for (int i = 1; i < argc; i++) {
        const std::string path = argv[i];
        try {
            if (endsWith(path, ".png")) {
                std::ifstream f(path, std::ios::binary);
                std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
                const gmupt::png::Image img = gmupt::png::decode(bytes.data(), bytes.size());
                if (img.width == img.height && img.width && img.width <= 512) for (unsigned to : { 24u, 1u, 7u, img.width, img.width + 1, img.width * 2 + 3 }) (void)gmupt::resizeSquare(img.rgba.data(), img.width, to);   // every filter structure of AvirResize.cpp under the sanitizers
            } else if (endsWith(path, ".params")) {
                (void)SceneParams::load(path);
            } else {
                MeshData m = endsWith(path, ".gmesh") ? MeshData::load(path) : MeshData::loadGltf(path);
                for (int t = 0; t < 3; t++) if (!m.textureFiles[t].empty()) (void)gmupt::loadSpecificTexture(m.textureFiles[t], m.materials, t);
                for (int32_t v : m.indices) if (v < 0 || (size_t)v >= m.numVertices()) { std::printf("BAD INDEX in %s\n", path.c_str()); return 3; }
            }
            ok++;
        } catch (const std::exception&) { rejected++; }
    }
    std::printf("loader fuzz: %d accepted, %d rejected\n", ok, rejected);
    return 0;
}
