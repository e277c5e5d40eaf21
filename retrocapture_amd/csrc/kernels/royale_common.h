// Shared by the crt-royale kernel files (pass_royale.hip, pass_royale_scan.hip): float helpers with the GL's
// NaN behaviour, and the compile-time / run-time sampler and store policies the kernels are instantiated on.
#pragma once
#include "pass_launch.h"
#include "rc_vecmath.h"
#include "royale_params.h"

namespace rcroyale {
using namespace rcd;

__device__ __forceinline__ float minps(float a, float b) { return a < b ? a : b; }  // NaN -> b
__device__ __forceinline__ float maxps(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return minps(maxps(x, lo), hi); }
__device__ __forceinline__ float fractf(float x) { return x - __builtin_floorf(x); }
__device__ __forceinline__ float mod_glsl(float x, float y) { return x - y * __builtin_floorf(x / y); }
__device__ __forceinline__ float mix_rt(float a, float b, float t) { return a + t * (b - a); }

// Sampler / store policies: the shipped preset's texture formats and sampler states get
// compile-time specialised kernels; any other configuration runs the run-time selected ones.
template <int FMT, int LIN, int WRAP>
struct S {
  static __device__ __forceinline__ float4 get(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* l) {
    return sample<FMT, LIN, WRAP>(t, img, s, v, l);
  }
  static bool matches(const Tex& t) { return t.fmt == FMT && (t.linear != 0) == (LIN != 0) && t.wrap == WRAP; }
  // 8-bit texels: every sampled value is 0 or in [2^-40, 1], which is what div_safe_ needs
  static constexpr bool kUnitRange = FMT != FMT_F32 && FMT != FMT_F16;
};
struct SRT {
  static __device__ __forceinline__ float4 get(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* l) {
    return sample_rt(t, img, s, v, l);
  }
  static bool matches(const Tex&) { return true; }
  static constexpr bool kUnitRange = false;  // may be an RGBA32F texture with arbitrary values
};
template <int OUT>
struct St {
  static __device__ __forceinline__ void put(const PassLaunch& L, int z, int x, int y, float4 c, const SrgbLds* l) { store<OUT>(L, z, x, y, c, l); }
  static bool matches(const PassLaunch& L) { return L.out_fmt == OUT; }
};
struct StRT {
  static __device__ __forceinline__ void put(const PassLaunch& L, int z, int x, int y, float4 c, const SrgbLds* l) { store_rt(L, z, x, y, c, l); }
  static bool matches(const PassLaunch&) { return true; }
};
using SrgbLinEdge = S<FMT_SRGB8, 1, WRAP_EDGE>;
using SrgbNearEdge = S<FMT_SRGB8, 0, WRAP_EDGE>;

constexpr float kUnderHalf = 0.4995f;

// ---- the last pass's output gamma from a table with a measured bound (built and described in pass_royale.hip; used by the flat
// strip form there and by the general form in pass_royale_last_general.hip).  641 log-spaced nodes from 2^-20 to 1, a quadratic in
// the colour itself per node and the largest error over every float of the node in .w.
#ifndef RC_AS3
#define RC_AS3 __attribute__((address_space(3)))
#endif
constexpr uint32_t kLastTabBits0 = 0x35800000u;   // 2^-20
constexpr int kLastTabShift = 18;                 // 2^18 floats per node: 32 nodes per octave
constexpr int kLastTabNodes = (int)((0x3f800000u - kLastTabBits0) >> kLastTabShift) + 1;   // 641: the last one is the colour 1.0 alone
constexpr uint32_t kLastLdsTab = 1024u;   // LDS byte offset of the table (behind the decode table; the kernels have no static LDS)
// one channel of one pixel from the gamma table: the byte, *fail set when it is not certain.  0 <= c <= 1 (callers send anything
// else to the exact code).  The record's polynomial gives y ~ 255 G(c) and its .w the threshold 0.5 - E, E the measured bound on the
// distance of y from the value the exact code rounds (pass_royale.hip k_last_gamma_err / _finish): with r = rint(y) the byte is r
// whenever |y - r| < .w.  r needs no clamp: y lies within E of [0, 255], and the integer conversion takes a negative zero to 0.
__device__ __forceinline__ uint32_t last_gamma_byte(float c, bool* fail) {
  typedef float last_v4f __attribute__((ext_vector_type(4)));
  // colours below the table's first node store 0 like that node's first colour does (G is monotone, G(2^-20) * 255 < 0.47)
  const uint32_t cb = max(f2bits(c), kLastTabBits0);
  const uint32_t off = ((cb - kLastTabBits0) >> (kLastTabShift - 4)) & ~15u;
  const last_v4f e = *reinterpret_cast<const RC_AS3 last_v4f*>((uintptr_t)(kLastLdsTab + off));
  const float cc = bits2f(cb);
  const float y = fma_(cc, fma_(cc, e.z, e.y), e.x);
  const float r = __builtin_rintf(y);
  *fail = *fail || !(__builtin_fabsf(y - r) < e.w);
  return (uint32_t)r;
}

}  // namespace rcroyale
