#include "png_lut.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <filesystem>

namespace rc {
namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

}  // namespace

bool loadPngRgba8(const std::string& path, std::vector<uint8_t>* rgba, int* width, int* height, std::string* error) {
  auto fail = [&](const std::string& m) {
    if (error) *error = m + ": " + path;
    return false;
  };
  std::error_code ec;
  if (!std::filesystem::exists(path, ec)) return fail("Texture file not found");
  FILE* fp = std::fopen(path.c_str(), "rb");
  if (!fp) return fail("Failed to open texture file");
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), fp)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(fp);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return fail("File is not a valid PNG");

  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  size_t pos = 8;
  bool gotHeader = false;
  while (pos + 12 <= file.size()) {
    uint32_t len = be32(&file[pos]);
    const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
    if (pos + 12 + (size_t)len > file.size()) return fail("Truncated PNG chunk");
    const uint8_t* data = &file[pos + 8];
    if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
      W = be32(data);
      H = be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
      gotHeader = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "tRNS", 4)) {
      trns.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!gotHeader || W == 0 || H == 0) return fail("PNG has no header");
  if (interlace != 0) return fail("Interlaced PNG is not supported");
  int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!channels || (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)) return fail("Unsupported PNG format");

  const size_t bpp_bits = (size_t)channels * depth;
  const size_t stride = (W * bpp_bits + 7) / 8;
  const size_t filt_bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
  std::vector<uint8_t> raw((stride + 1) * (size_t)H);
  uLongf rawLen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
    return fail("Error processing PNG");

  // undo the per-row filters in place
  std::vector<uint8_t> img(stride * (size_t)H);
  for (uint32_t y = 0; y < H; ++y) {
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t* src = &raw[(stride + 1) * y + 1];
    uint8_t* dst = &img[stride * y];
    const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
    for (size_t i = 0; i < stride; ++i) {
      int a = i >= filt_bpp ? dst[i - filt_bpp] : 0;
      int b = up ? up[i] : 0;
      int c = (up && i >= filt_bpp) ? up[i - filt_bpp] : 0;
      int v = src[i];
      switch (ft) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: break;
      }
      dst[i] = (uint8_t)v;
    }
  }

  rgba->assign((size_t)W * H * 4, 255);
  auto sample = [&](const uint8_t* row, size_t index) -> unsigned {  // index-th sample of the row
    if (depth == 8) return row[index];
    if (depth == 16) return row[index * 2];  // strip_16 keeps the high byte
    const size_t bit = index * depth;
    return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
  };
  auto sample16 = [&](const uint8_t* row, size_t index) -> unsigned { return ((unsigned)row[index * 2] << 8) | row[index * 2 + 1]; };
  for (uint32_t y = 0; y < H; ++y) {
    const uint8_t* row = &img[stride * y];
    uint8_t* out = &(*rgba)[(size_t)y * W * 4];
    for (uint32_t x = 0; x < W; ++x, out += 4) {
      if (ctype == 3) {
        unsigned idx = sample(row, x);
        for (int c = 0; c < 3; ++c) out[c] = idx * 3 + c < plte.size() ? plte[idx * 3 + c] : 0;
        out[3] = idx < trns.size() ? trns[idx] : 255;
      } else if (ctype == 0 || ctype == 4) {
        unsigned g = sample(row, (size_t)x * channels);
        bool transparent = false;
        if (ctype == 0 && trns.size() >= 2) {
          unsigned key = ((unsigned)trns[0] << 8) | trns[1];
          unsigned full = depth == 16 ? sample16(row, x) : g;
          transparent = full == key;
        }
        if (depth < 8) g = g * 255u / ((1u << depth) - 1u);  // expand_gray_1_2_4_to_8
        out[0] = out[1] = out[2] = (uint8_t)g;
        out[3] = ctype == 4 ? (uint8_t)sample(row, (size_t)x * 2 + 1) : (transparent ? 0 : 255);
      } else {  // RGB / RGBA
        for (int c = 0; c < 3; ++c) out[c] = (uint8_t)sample(row, (size_t)x * channels + c);
        if (ctype == 6) {
          out[3] = (uint8_t)sample(row, (size_t)x * 4 + 3);
        } else if (trns.size() >= 6) {
          bool match = true;
          for (int c = 0; c < 3; ++c) {
            unsigned key = ((unsigned)trns[c * 2] << 8) | trns[c * 2 + 1];
            unsigned full = depth == 16 ? sample16(row, (size_t)x * 3 + c) : sample(row, (size_t)x * 3 + c);
            match &= full == key;
          }
          out[3] = match ? 0 : 255;
        }
      }
    }
  }
  *width = (int)W;
  *height = (int)H;
  return true;
}

}  // namespace rc
