// Frame ingest / egress: the byte shuffles either side of the shader chain.
//   ingest  = what FrameProcessor::processFrame makes of a captured buffer before the chain sees it
//             (reference src/processing/FrameProcessor.cpp:43-222): RGB24 / BGRA / RGBA / YUYV422 ->
//             the chain's RGBA8 source frame (alpha is irrelevant downstream - the reference's source
//             texture is GL_RGB - and is written as 255), row 0 first, no flip.
//   egress  = the readback's alpha strip (reference src/core/FrameCapturePipeline.cpp:1060-1080):
//             RGBA8 -> tightly packed RGB24, optionally with the rows reversed.
// Pure HBM streaming: every thread moves 4 pixels with dword-wide accesses (RGB24: three dwords
// <-> four dwords), frames are addressed as flat pixel arrays (tight packing at both ends), the
// grid is sized to the data with a grid-stride tail.
#include "pass_launch.h"

namespace {

__device__ __forceinline__ uint32_t clip8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// BT.601 limited-range YCbCr -> RGB in 16.16 fixed point with FFmpeg's ITU-601 coefficients
// (libswscale yuv2rgb.c ff_yuv2rgb_coeffs[SWS_CS_ITU601] = {104597, 132201, 25675, 53279},
// luma gain 65536*255/219 = 76309).  libswscale itself is not in the build image, so this is the
// published algorithm, not a pinned copy of the library's (SIMD, build-specific) output.
__device__ __forceinline__ uint32_t ycc_to_rgba(int y, int u, int v) {
  const int c = 76309 * (y - 16) + 32768, d = u - 128, e = v - 128;
  uint32_t r = clip8((c + 104597 * e) >> 16);
  uint32_t g = clip8((c - 25675 * d - 53279 * e) >> 16);
  const uint32_t b = clip8((c + 132201 * d) >> 16);
  // Keep r and g opaque to the instruction selector: ROCm 7.2's gfx950 backend otherwise fuses
  // "shift right, clamp to a byte, pack two" into v_ashr_pk_u8_i32 and then treats the upper 16
  // bits of that result as zero, which they are not on the device - the OR below picked up
  // stray bits in the blue byte (caught by tests/test_frame_io.py).
  asm volatile("" : "+v"(r), "+v"(g));
  return r | (g << 8) | (b << 16) | 0xff000000u;
}

// 4 pixels per thread: 12 bytes in, 16 bytes out
__global__ void __launch_bounds__(256) k_rgb24_to_rgba8(const uint32_t* __restrict__ src, uint4* __restrict__ dst,
                                                        size_t n_quads, const uint8_t* src_b, uint32_t* dst_px, size_t n_px) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * 256) {
    const uint32_t a = src[3 * q], b = src[3 * q + 1], c = src[3 * q + 2];
    uint4 o;
    o.x = (a & 0x00ffffffu) | 0xff000000u;
    o.y = (a >> 24) | ((b & 0x0000ffffu) << 8) | 0xff000000u;
    o.z = (b >> 16) | ((c & 0x000000ffu) << 16) | 0xff000000u;
    o.w = (c >> 8) | 0xff000000u;
    dst[q] = o;
  }
  // tail (n_px % 4 pixels), one thread
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t p = n_quads * 4; p < n_px; ++p)
      dst_px[p] = src_b[3 * p] | (src_b[3 * p + 1] << 8) | (src_b[3 * p + 2] << 16) | 0xff000000u;
}

template <bool SWAP_RB>
__global__ void __launch_bounds__(256) k_x8_to_rgba8(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n_quads,
                                                     const uint32_t* src_px, uint32_t* dst_px, size_t n_px) {
  auto cvt = [](uint32_t p) -> uint32_t {
    if (SWAP_RB) p = (p & 0x0000ff00u) | ((p >> 16) & 0xffu) | ((p & 0xffu) << 16);
    return p | 0xff000000u;
  };
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * 256) {
    const uint4 i = src[q];
    dst[q] = make_uint4(cvt(i.x), cvt(i.y), cvt(i.z), cvt(i.w));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t p = n_quads * 4; p < n_px; ++p) dst_px[p] = cvt(src_px[p]);
}

// 4 pixels = 2 YUYV macropixels (8 bytes) in, 16 bytes out; width is even by format definition
__global__ void __launch_bounds__(256) k_yuyv_to_rgba8(const uint2* __restrict__ src, uint4* __restrict__ dst, size_t n_quads,
                                                       const uint32_t* src_mp, uint2* dst_mp, size_t n_macro) {
  auto cvt = [](uint32_t m) -> uint2 {
    const int y0 = m & 255, u = (m >> 8) & 255, y1 = (m >> 16) & 255, v = m >> 24;
    return make_uint2(ycc_to_rgba(y0, u, v), ycc_to_rgba(y1, u, v));
  };
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * 256) {
    const uint2 i = src[q];
    const uint2 a = cvt(i.x), b = cvt(i.y);
    dst[q] = make_uint4(a.x, a.y, b.x, b.y);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t p = n_quads * 2; p < n_macro; ++p) dst_mp[p] = cvt(src_mp[p]);
}

// RGBA8 -> RGB24, 4 pixels per thread; with flip the source row is mirrored per frame, which
// needs row-aware addressing: handled per pixel quad inside a row when w % 4 == 0, else per pixel.
__global__ void __launch_bounds__(256) k_rgba8_to_rgb24(const uint4* __restrict__ src, uint32_t* __restrict__ dst, size_t n_quads,
                                                        const uint32_t* src_px, uint8_t* dst_b, size_t n_px) {
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * 256) {
    const uint4 i = src[q];
    dst[3 * q] = (i.x & 0x00ffffffu) | (i.y << 24);
    dst[3 * q + 1] = ((i.y >> 8) & 0x0000ffffu) | (i.z << 16);
    dst[3 * q + 2] = ((i.z >> 16) & 0x000000ffu) | (i.w << 8);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t p = n_quads * 4; p < n_px; ++p) {
      const uint32_t v = src_px[p];
      dst_b[3 * p] = (uint8_t)v;
      dst_b[3 * p + 1] = (uint8_t)(v >> 8);
      dst_b[3 * p + 2] = (uint8_t)(v >> 16);
    }
}

// flipped variant: one thread per 4-pixel group of a row (w % 4 == 0) - rows reversed within each frame
__global__ void __launch_bounds__(256) k_rgba8_to_rgb24_flip(const uint4* __restrict__ src, uint32_t* __restrict__ dst,
                                                             uint32_t quads_per_row, uint32_t h, uint32_t n_frames) {
  const size_t total = (size_t)quads_per_row * h * n_frames;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
    const size_t row_all = q / quads_per_row;
    const uint32_t col = (uint32_t)(q - row_all * quads_per_row);
    const size_t frame = row_all / h;
    const uint32_t row = (uint32_t)(row_all - frame * h);
    const size_t sq = (frame * h + (h - 1 - row)) * quads_per_row + col;
    const uint4 i = src[sq];
    dst[3 * q] = (i.x & 0x00ffffffu) | (i.y << 24);
    dst[3 * q + 1] = ((i.y >> 8) & 0x0000ffffu) | (i.z << 16);
    dst[3 * q + 2] = ((i.z >> 16) & 0x000000ffu) | (i.w << 8);
  }
}
// generic flipped variant, one pixel per thread (any width)
__global__ void __launch_bounds__(256) k_rgba8_to_rgb24_flip_px(const uint32_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                uint32_t w, uint32_t h, uint32_t n_frames) {
  const size_t total = (size_t)w * h * n_frames;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
    const size_t row_all = p / w;
    const uint32_t col = (uint32_t)(p - row_all * w);
    const size_t frame = row_all / h;
    const uint32_t row = (uint32_t)(row_all - frame * h);
    const uint32_t v = src[(frame * h + (h - 1 - row)) * w + col];
    dst[3 * p] = (uint8_t)v;
    dst[3 * p + 1] = (uint8_t)(v >> 8);
    dst[3 * p + 2] = (uint8_t)(v >> 16);
  }
}

inline unsigned io_grid(size_t items) {
  const size_t b = (items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));  // >= 64 workgroups per CU at full size, grid-stride beyond
}

}  // namespace

namespace rck {

hipError_t launch_ingest(const void* src, int fmt, uint32_t w, uint32_t h, uint32_t n, void* dst, hipStream_t s) {
  const size_t n_px = (size_t)w * h * n;
  if (n_px == 0) return hipSuccess;
  switch (fmt) {
    case 0: {  // RGB24
      const size_t nq = n_px / 4;
      hipLaunchKernelGGL(k_rgb24_to_rgba8, dim3(io_grid(nq)), dim3(256), 0, s, (const uint32_t*)src, (uint4*)dst, nq,
                         (const uint8_t*)src, (uint32_t*)dst, n_px);
      break;
    }
    case 1:  // BGRA
    case 2: {  // RGBA
      const size_t nq = n_px / 4;
      if (fmt == 1)
        hipLaunchKernelGGL(k_x8_to_rgba8<true>, dim3(io_grid(nq)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, nq,
                           (const uint32_t*)src, (uint32_t*)dst, n_px);
      else
        hipLaunchKernelGGL(k_x8_to_rgba8<false>, dim3(io_grid(nq)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, nq,
                           (const uint32_t*)src, (uint32_t*)dst, n_px);
      break;
    }
    case 3: {  // YUYV422
      if (w & 1u) return hipErrorInvalidValue;
      const size_t n_macro = n_px / 2, nq = n_macro / 2;
      hipLaunchKernelGGL(k_yuyv_to_rgba8, dim3(io_grid(nq)), dim3(256), 0, s, (const uint2*)src, (uint4*)dst, nq,
                         (const uint32_t*)src, (uint2*)dst, n_macro);
      break;
    }
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_egress_rgb24(const void* src, uint32_t w, uint32_t h, uint32_t n, int flip_y, void* dst, hipStream_t s) {
  const size_t n_px = (size_t)w * h * n;
  if (n_px == 0) return hipSuccess;
  if (!flip_y) {
    const size_t nq = n_px / 4;
    hipLaunchKernelGGL(k_rgba8_to_rgb24, dim3(io_grid(nq)), dim3(256), 0, s, (const uint4*)src, (uint32_t*)dst, nq,
                       (const uint32_t*)src, (uint8_t*)dst, n_px);
  } else if ((w & 3u) == 0) {
    hipLaunchKernelGGL(k_rgba8_to_rgb24_flip, dim3(io_grid(n_px / 4)), dim3(256), 0, s, (const uint4*)src, (uint32_t*)dst, w / 4, h, n);
  } else {
    hipLaunchKernelGGL(k_rgba8_to_rgb24_flip_px, dim3(io_grid(n_px)), dim3(256), 0, s, (const uint32_t*)src, (uint8_t*)dst, w, h, n);
  }
  return hipGetLastError();
}

}  // namespace rck
