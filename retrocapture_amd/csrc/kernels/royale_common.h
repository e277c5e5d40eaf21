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

}  // namespace rcroyale
