// OpenGLRenderer::renderTexture as the reference uses it OFF-screen (reference
// src/renderer/OpenGLRenderer.cpp:378-470, fragment program :141-158, quad :292-307):
//   pre-pass   NEAREST downscale + overscan crop of the captured frame into a GL_RGB target through
//              an enlarged, offset viewport            (src/core/FrameCapturePipeline.cpp:160-250)
//   resize     shader output -> the configured output resolution, GL_RGBA target        (:413-505)
//   bake       brightness / contrast baked into the frame handed to stream / recording  (:739-804)
// One textured quad over the viewport: t = texture(coord); rgb = ((t.rgb * brightness) - 0.5) *
// contrast + 0.5 (every operation rounded, as llvmpipe executes it); alpha = t.a; UNORM8 store.
// The reference runs resize and bake as two draws with an RGBA8 texture in between and then strips
// alpha on the host; here they are one kernel: the first draw's bytes stay in registers (the second
// draw samples texel centres, which returns the texel itself under either filter), and the store
// can go straight to tightly packed RGB24 with the rows reversed (the readback's alpha strip and
// flip, FrameCapturePipeline.cpp:1060-1080).
// HBM streaming with a gather on the read side: four target pixels per thread where the row length
// allows it (16-byte RGBA8 / 12-byte RGB24 stores), one otherwise.
#include "present.h"

using namespace rcd;

namespace {

__device__ __forceinline__ float adjust(float t, float b, float c) { return ((t * b) - 0.5f) * c + 0.5f; }

template <int FMT, int LINEAR>
__device__ __forceinline__ uint32_t present_px(const rck::PresentLaunch& L, const uint8_t* img, int x, int y) {
  if (x < L.cov_x0 || x >= L.cov_x1 || y < L.cov_y0 || y >= L.cov_y1) return L.clear;
  const float u = fma_(L.u_dx, (float)x, L.u_a0);
  float v = fma_(L.v_dy, (float)y, L.v_a0);
  if (L.flip_y) v = 1.0f - v;
  const float4 t = sample<FMT, LINEAR, WRAP_EDGE>(L.src, img, u, v, nullptr);
  uint32_t r = unorm8(adjust(t.x, L.brightness, L.contrast)), g = unorm8(adjust(t.y, L.brightness, L.contrast));
  uint32_t b = unorm8(adjust(t.z, L.brightness, L.contrast)), a = unorm8(t.w);
  if (L.bake) {
    const float k = 1.0f / 255.0f;
    r = unorm8(adjust((float)r * k, L.bake_brightness, L.bake_contrast));
    g = unorm8(adjust((float)g * k, L.bake_brightness, L.bake_contrast));
    b = unorm8(adjust((float)b * k, L.bake_brightness, L.bake_contrast));
  }
  return r | (g << 8) | (b << 16) | (a << 24);
}

// QUAD: items are runs of four pixels of one row (dst_w % 4 == 0), else single pixels
template <int FMT, int LINEAR, int KIND, bool QUAD>
__global__ void __launch_bounds__(256) k_present(const rck::PresentLaunch L) {
  const uint32_t per_row = QUAD ? (uint32_t)L.dst_w / 4u : (uint32_t)L.dst_w;
  const uint32_t per_frame = per_row * (uint32_t)L.dst_h;
  const uint64_t total = (uint64_t)per_frame * (uint32_t)L.n_frames;
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
    const uint32_t z = (uint32_t)(i / per_frame), r = (uint32_t)(i - (uint64_t)z * per_frame);
    const int y = (int)(r / per_row), xi = (int)(r - (uint32_t)y * per_row);
    const uint8_t* img = frame_ptr(L.src, (int)z);
    const int oy = L.out_flip_rows ? L.dst_h - 1 - y : y;
    uint8_t* row = static_cast<uint8_t*>(L.dst) + L.dst_frame_stride * (uint64_t)z;
    if (QUAD) {
      uint32_t p[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] = present_px<FMT, LINEAR>(L, img, xi * 4 + k, y);
      if (KIND == rck::PRESENT_RGB24) {
        uint32_t* d = reinterpret_cast<uint32_t*>(row + ((size_t)oy * L.dst_w + (size_t)xi * 4) * 3);
        d[0] = (p[0] & 0xffffffu) | (p[1] << 24);
        d[1] = ((p[1] >> 8) & 0xffffu) | (p[2] << 16);
        d[2] = ((p[2] >> 16) & 0xffu) | (p[3] << 8);
      } else {
        if (KIND == rck::PRESENT_RGBX8) {
#pragma unroll
          for (int k = 0; k < 4; ++k) p[k] |= 0xff000000u;
        }
        *reinterpret_cast<uint4*>(row + ((size_t)oy * L.dst_w + (size_t)xi * 4) * 4) = make_uint4(p[0], p[1], p[2], p[3]);
      }
    } else {
      const uint32_t p = present_px<FMT, LINEAR>(L, img, xi, y);
      if (KIND == rck::PRESENT_RGB24) {
        uint8_t* d = row + ((size_t)oy * L.dst_w + xi) * 3;
        d[0] = (uint8_t)p;
        d[1] = (uint8_t)(p >> 8);
        d[2] = (uint8_t)(p >> 16);
      } else {
        *reinterpret_cast<uint32_t*>(row + ((size_t)oy * L.dst_w + xi) * 4) = KIND == rck::PRESENT_RGBX8 ? (p | 0xff000000u) : p;
      }
    }
  }
}

template <int FMT, int LINEAR, int KIND>
hipError_t go(const rck::PresentLaunch& L, hipStream_t s) {
  const bool quad = (L.dst_w & 3) == 0;
  const uint64_t items = (uint64_t)(quad ? L.dst_w / 4 : L.dst_w) * L.dst_h * L.n_frames;
  const uint64_t blocks = (items + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks));
  if (quad) hipLaunchKernelGGL((k_present<FMT, LINEAR, KIND, true>), dim3(grid), dim3(256), 0, s, L);
  else hipLaunchKernelGGL((k_present<FMT, LINEAR, KIND, false>), dim3(grid), dim3(256), 0, s, L);
  return hipGetLastError();
}

template <int FMT, int LINEAR>
hipError_t go_kind(const rck::PresentLaunch& L, hipStream_t s) {
  switch (L.dst_kind) {
    case rck::PRESENT_RGBA8: return go<FMT, LINEAR, rck::PRESENT_RGBA8>(L, s);
    case rck::PRESENT_RGBX8: return go<FMT, LINEAR, rck::PRESENT_RGBX8>(L, s);
    case rck::PRESENT_RGB24: return go<FMT, LINEAR, rck::PRESENT_RGB24>(L, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

namespace rck {

hipError_t launch_present(const PresentLaunch& L, hipStream_t s) {
  if (L.n_frames <= 0 || L.dst_w <= 0 || L.dst_h <= 0) return hipSuccess;
  if (L.src.fmt == FMT_RGBX8) return L.src.linear ? go_kind<FMT_RGBX8, 1>(L, s) : go_kind<FMT_RGBX8, 0>(L, s);
  if (L.src.fmt == FMT_RGBA8) return L.src.linear ? go_kind<FMT_RGBA8, 1>(L, s) : go_kind<FMT_RGBA8, 0>(L, s);
  return hipErrorInvalidValue;
}

}  // namespace rck
