// The float primitives of rc_device.h once more, generic over one float or a pair of floats held in
// a 2-wide vector.  CDNA's VALU issues a packed v_pk_{fma,mul,add}_f32 at (nearly) the cost of the
// scalar instruction - measured on MI355X: 35 T scalar FMA/s vs 62 T packed FMA/s per device - so a
// kernel whose time is polynomial arithmetic runs two independent evaluations per lane for the
// price of one.  Every operation is the same IEEE operation per component, so results are bit-identical
// to the scalar functions (the GPU parity tests compare the packed kernels against the oracle).
// Device only; compiled with -ffp-contract=off like everything else.
#pragma once
#include "rc_device.h"

namespace rcd {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

template <class F> struct Lanes;
template <> struct Lanes<float> { using I = int32_t; using U = uint32_t; };
template <> struct Lanes<v2f> { using I = v2i; using U = v2u; };

#define RC_D __device__ __forceinline__
RC_D float fma_v(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RC_D v2f fma_v(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
RC_D float floor_v(float x) { return __builtin_floorf(x); }
RC_D v2f floor_v(v2f x) { return v2f{__builtin_floorf(x.x), __builtin_floorf(x.y)}; }
RC_D float abs_v(float x) { return __builtin_fabsf(x); }
RC_D v2f abs_v(v2f x) { return v2f{__builtin_fabsf(x.x), __builtin_fabsf(x.y)}; }
RC_D float rcp_v(float x) { return __builtin_amdgcn_rcpf(x); }
RC_D v2f rcp_v(v2f x) { return v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
RC_D uint32_t bits_v(float x) { return f2bits(x); }
RC_D v2u bits_v(v2f x) { return __builtin_bit_cast(v2u, x); }
RC_D float float_of_bits(uint32_t u) { return bits2f(u); }
RC_D v2f float_of_bits(v2u u) { return __builtin_bit_cast(v2f, u); }
RC_D int32_t int_of(float x) { return (int32_t)x; }
RC_D v2i int_of(v2f x) { return __builtin_convertvector(x, v2i); }
RC_D float float_of(int32_t i) { return (float)i; }
RC_D v2f float_of(v2i i) { return __builtin_convertvector(i, v2f); }
RC_D uint32_t as_u(int32_t i) { return (uint32_t)i; }
RC_D v2u as_u(v2i i) { return __builtin_bit_cast(v2u, i); }
RC_D int32_t as_i(uint32_t u) { return (int32_t)u; }
RC_D v2i as_i(v2u u) { return __builtin_bit_cast(v2i, u); }
// x > c ? c : x and x < c ? c : x per component (a NaN passes through, as in the scalar code)
RC_D float clamp_hi(float x, float c) { return x > c ? c : x; }
RC_D v2f clamp_hi(v2f x, float c) { return v2f{x.x > c ? c : x.x, x.y > c ? c : x.y}; }
RC_D float clamp_lo(float x, float c) { return x < c ? c : x; }
RC_D v2f clamp_lo(v2f x, float c) { return v2f{x.x < c ? c : x.x, x.y < c ? c : x.y}; }

// clamp to [lo, hi] in one v_med3_f32 per component; differs from the two selects only for a NaN
RC_D float clamp_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
RC_D v2f clamp_med3(v2f x, float lo, float hi) { return v2f{__builtin_amdgcn_fmed3f(x.x, lo, hi), __builtin_amdgcn_fmed3f(x.y, lo, hi)}; }

// NONAN: the caller guarantees the argument is not a NaN (it may be +-inf)
template <class F, bool NONAN = false>
RC_D F exp2_v(F x) {
  if (NONAN) {
    x = clamp_med3(x, -126.99999f, 128.0f);
  } else {
    x = clamp_hi(x, 128.0f);
    x = clamp_lo(x, -126.99999f);
  }
  const F ip = floor_v(x);
  const F fp = x - ip;
  const F e = float_of_bits(as_u((int_of(ip) + 127) << 23));
  const F x2 = fp * fp;
  F even = fma_v(x2, F(0.00898934009049466391101f), F(0.240153617044375388211f));
  even = fma_v(x2, even, F(1.0f));
  F odd = fma_v(x2, F(0.00187757667519147912699f), F(0.0558263180532956664775f));
  odd = fma_v(x2, odd, F(0.693153073200168932794f));
  return e * fma_v(odd, fp, even);
}
template <class F, bool NONAN = false>
RC_D F exp_v(F x) { return exp2_v<F, NONAN>(x * 1.4426950408889634f); }

// div_log2_ / div_safe_ / div_const_ of rc_device.h
template <class F>
RC_D F div_log2_v(F n, F d) {
  F r = rcp_v(d);
  r = fma_v(fma_v(-d, r, F(1.0f)), r, r);
  const F q = n * r;
  return fma_v(fma_v(-d, q, n), r, q);
}
template <class F>
RC_D F div_safe_v(F n, F d) {
  F r = rcp_v(d);
  r = fma_v(fma_v(-d, r, F(1.0f)), r, r);
  F q = n * r;
  q = fma_v(fma_v(-d, q, n), r, q);
  return fma_v(fma_v(-d, q, n), r, q);
}
template <class F>
RC_D F div_const_v(F x, float c, float rc) {
  const F q = x * rc;
  const F r = fma_v(F(-c), q, x);
  return fma_v(r, F(rc), q);
}

template <class F>
RC_D F log2_core_v(F x) {
  const auto i = bits_v(x);
  const F logexp = float_of(as_i((i & 0x7f800000u) >> 23) - 127);
  const F mant = float_of_bits((i & 0x007fffffu) | 0x3f800000u);
  const F y = div_log2_v<F>(mant - 1.0f, mant + 1.0f);
  const F z = y * y;
  const F z2 = z * z;
  F even = fma_v(z2, F(0.406718052498846252698f), F(0.577440339438736392009f));
  even = fma_v(z2, even, F(2.88539009343309178325f));
  const F odd = fma_v(z2, F(0.403343858251329912514f), F(0.961791550404184197881f));
  const F p = fma_v(odd, z, even);
  return fma_v(y, p, logexp);
}
// the edge cases of log2_: zero / denormal -> -inf, negative -> NaN, +inf -> +inf
RC_D float log2_edge(float core, float x) {
  const uint32_t i = f2bits(x);
  if ((i & 0x7f800000u) == 0u) return -__builtin_inff();
  if (i & 0x80000000u) return __builtin_nanf("");
  if (i == 0x7f800000u) return __builtin_inff();
  return core;
}
RC_D float log2_v(float x) { return log2_edge(log2_core_v<float>(x), x); }
RC_D v2f log2_v(v2f x) {
  const v2f c = log2_core_v<v2f>(x);
  return v2f{log2_edge(c.x, x.x), log2_edge(c.y, x.y)};
}
#undef RC_D

}  // namespace rcd
