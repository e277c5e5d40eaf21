// PNG -> 8-bit RGBA for preset LUT textures.
// Contract: reference ShaderEngine::loadTextureReference (ShaderEngine.cpp:2535-2706), which
// drives libpng with: strip 16-bit to 8, palette -> RGB, gray 1/2/4 -> 8, tRNS -> alpha,
// gray -> RGB, opaque alpha 0xFF added to RGB / gray images; row 0 of the file is t = 0.
// Implemented on zlib's inflate (libpng headers are not part of the build image).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rc {
bool loadPngRgba8(const std::string& path, std::vector<uint8_t>* rgba, int* width, int* height, std::string* error);
}
