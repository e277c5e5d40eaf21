/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * Float primitives as Mesa 23.2.1 llvmpipe evaluates GLSL built-ins (gallivm
 * lp_bld_arit.c: lp_build_exp2 / lp_build_log2_approx / lp_build_pow /
 * lp_build_sin_or_cos).  Mesa is a third-party dependency that is not vendored in the
 * reference (system package libgl1-mesa-dri 23.2.1-1ubuntu3.1~22.04.3); its published
 * algorithm is restated here and was checked bit-for-bit against the real driver over
 * 2.6e5 random arguments per function (oracle/probes/llvmpipe_math_model.py).
 * All multiply-adds that llvmpipe fuses are explicit fmaf(); the file is compiled with
 * -ffp-contract=off so nothing else fuses.
 */
#include <math.h>
#include <string.h>
#include <xmmintrin.h>

#include "rc_oracle.h"

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* minimax 2^x on [0,1), degree 5, evaluated as even/odd halves in x^2 */
float o_exp2(float x) {
  if (x > 128.0f) x = 128.0f;        /* NaN passes through (min/max keep the NaN) */
  if (x < -126.99999f) x = -126.99999f;
  float ip = floorf(x);
  float fp = x - ip;
  float e = bits2f((uint32_t)(((int32_t)ip + 127) << 23));
  float x2 = fp * fp;
  float even = fmaf(x2, 0.00898934009049466391101f, 0.240153617044375388211f);
  even = fmaf(x2, even, 1.0f);
  float odd = fmaf(x2, 0.00187757667519147912699f, 0.0558263180532956664775f);
  odd = fmaf(x2, odd, 0.693153073200168932794f);
  return e * fmaf(odd, fp, even);
}

/* log2(m) for m in [1,2) as y*P(y^2), y = (m-1)/(m+1), degree-4 P; + exponent */
float o_log2(float x) {
  /* edge cases as llvmpipe's lp_build_log2 "safe" form returns them (measured): zero and
   * denormals (the GL runs with denormals-are-zero) -> -inf, negative -> NaN, +inf -> +inf */
  uint32_t i = f2bits(x);
  if ((i & 0x7f800000u) == 0) return -INFINITY;
  if (i & 0x80000000u) return NAN;
  if (i == 0x7f800000u) return INFINITY;
  float logexp = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
  float mant = bits2f((i & 0x007fffffu) | 0x3f800000u);
  float y = (mant - 1.0f) / (mant + 1.0f);
  float z = y * y;
  float z2 = z * z;
  float even = fmaf(z2, 0.406718052498846252698f, 0.577440339438736392009f);
  even = fmaf(z2, even, 2.88539009343309178325f);
  float odd = fmaf(z2, 0.403343858251329912514f, 0.961791550404184197881f);
  float p = fmaf(odd, z, even);
  return fmaf(y, p, logexp);
}

/* lp_build_pow selects 0 where "x == 0" under an unordered compare, so a NaN base gives 0 too (measured:
 * pow(NaN, y) = 0 for quiet / signalling / negative NaNs, while log2(NaN) and exp2(NaN) are NaN) */
float o_pow(float x, float y) {
  if (x != x) return 0.0f;
  return o_exp2(o_log2(x) * y);
}
float o_exp(float x) { return o_exp2(x * 1.4426950408889634f); }
float o_log(float x) { return o_log2(x) * 0.69314718055994529f; }

/* cephes-style sin/cos (4/pi range reduction, two polynomials) */
static float sincos_impl(float x, int want_cos) {
  uint32_t xi = f2bits(x);
  float xa = bits2f(xi & 0x7fffffffu);
  uint32_t sign = xi & 0x80000000u;
  float y = xa * 1.27323954473516f;
  int32_t j = (int32_t)y;
  j = (j + 1) & ~1;
  float y2 = (float)j;
  int32_t j2 = want_cos ? j - 2 : j;
  uint32_t swap = want_cos ? (((uint32_t)~j2 & 4u) << 29) : (((uint32_t)j & 4u) << 29);
  int poly_sin = (j2 & 2) == 0;
  float x3 = fmaf(y2, -0.78515625f, xa);
  x3 = fmaf(y2, -2.4187564849853515625e-4f, x3);
  x3 = fmaf(y2, -3.77489497744594108e-8f, x3);
  float z = x3 * x3;
  float r;
  if (poly_sin) {
    float ys = fmaf(-1.9515295891E-4f, z, 8.3321608736E-3f);
    ys = fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    r = fmaf(ys, x3, x3);
  } else {
    float yc = fmaf(2.443315711809948E-005f, z, -1.388731625493765E-003f);
    yc = fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    yc = yc * z;
    yc = yc - z * 0.5f;
    r = yc + 1.0f;
  }
  uint32_t sb = want_cos ? swap : (sign ^ swap);
  return bits2f(f2bits(r) ^ sb);
}
float o_sin(float x) { return sincos_impl(x, 0); }
float o_cos(float x) { return sincos_impl(x, 1); }

/* llvmpipe executes shaders with flush-to-zero and denormals-are-zero set in MXCSR; every
 * oracle pass runs between these two calls so that tiny intermediates behave the same. */
unsigned o_fp_enter(void) {
  unsigned old = _mm_getcsr();
  _mm_setcsr(old | 0x8040u); /* FTZ (bit 15) | DAZ (bit 6) */
  return old;
}
void o_fp_leave(unsigned old) { _mm_setcsr(old); }
