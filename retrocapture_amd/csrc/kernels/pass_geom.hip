// crt/crt-geom.glslp: one pass, reference shaders/shaders_glsl/crt/shaders/crt-geom.glsl (FS main 377-504, corner 327-337,
// scanlineWeights 345-367, saturation 369-382).  Restated in oracle/rc_passes_geom.c, bit-identical to llvmpipe on the
// float goldens (tests/golden/f32_crt_geom_*); this kernel follows the oracle operation by operation.
// plane[0], plane[1]: TEX0 = TexCoord * 1.0001; plane[2]: mod_factor.  params: geom_math.h GP_*.
#include "geom_math.h"
#include "pass_launch.h"

namespace rck {
using namespace rcd;
using namespace rcgeom;

namespace {

// llvmpipe's pow selects 0 where "x == 0" under an unordered compare: a NaN base gives 0 (oracle/rc_math.c o_pow).
// NaNs arise here where the viewing ray misses the tube (strong tilt, screen corners).
__device__ __forceinline__ float pow_gl(float x, float y) { return x != x ? 0.0f : pow_(x, y); }

__device__ __forceinline__ void scanline_weights(float distance, const float* col, float sw, float lum, float* out) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float c2 = col[c] * col[c];
    const float wid = 2.0f + 2.0f * (c2 * c2);   // pow(color, 4.0) is lowered to two squarings
    const float w = distance / sw;
    const float p = pow_gl(w * (1.0f / __builtin_sqrtf(0.5f * wid)), wid);
    out[c] = ((lum + 1.4f) * exp_(-p)) / (0.6f + 0.2f * wid);
  }
}

// GENERIC false: GL_RGB source (RGBX8) NEAREST clamp-to-edge into a plain RGBA8 target - the shipped preset
template <bool GENERIC>
__global__ void __launch_bounds__(256) k_crt_geom(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float* P = L.params;
  const float crt_gamma = P[GP_CRTGAMMA], inv_mon = 1.0f / P[GP_MONGAMMA], cornersize = P[GP_CORNERSIZE], cornersmooth = P[GP_CORNERSMOOTH];
  const float ovx = P[GP_OVERSCAN_X] / 100.0f, ovy = P[GP_OVERSCAN_Y] / 100.0f, dotmask = P[GP_DOTMASK], sw = P[GP_SCANLINE_WEIGHT];
  const float lum = P[GP_LUM], satur = P[GP_SATURATION];
  const bool curved = P[GP_CURVATURE] > 0.5f;
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;   // TextureSize == InputSize
  const float aspx = 1.0f, aspy = 0.75f;
  const Tube g = {P[GP_R], P[GP_D], P[GP_SIN_X], P[GP_SIN_Y], P[GP_COS_X], P[GP_COS_Y]};
  const float stx = P[GP_STRETCH_X], sty = P[GP_STRETCH_Y], stz = P[GP_STRETCH_Z];
  const float ilfac_y = minps(maxps(__builtin_floorf(tsy / 200.0f), 1.0f), 2.0f);
  const float one_x = 1.0f / (P[GP_SHARPER] * tsx), one_y = ilfac_y / tsy;
  const float filter_ = tsy / (float)L.out_h;
  const float pi = 3.141592653589f;
  RC_TILE_LOOP_BEGIN
  // ilvec.y: the interlacing simulation alternates fields with FrameCount when the source has >= 400 lines
  const float fc = (float)(L.frame_count0 + z);
  const float ilvec_y = ilfac_y * P[GP_INTERLACE] > 1.5f ? fc - 2.0f * __builtin_floorf(fc / 2.0f) : 0.0f;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  float xx = u, xy = v;
  if (curved) {   // transform()
    V2 c = {u * (tsx / tsx), v * (tsy / tsy)};
    c.x = ((c.x - 0.5f) * aspx) * stz + stx;
    c.y = ((c.y - 0.5f) * aspy) * stz + sty;
    const V2 b = bkwtrans<true>(g, c);
    xx = ((((b.x / ovx) / aspx) + 0.5f) * tsx) / tsx;
    xy = ((((b.y / ovy) / aspy) + 0.5f) * tsy) / tsy;
  }
  float cval;
  {   // corner()
    const float tx = (xx * (tsx / tsx) - 0.5f) * ovx, ty = (xy * (tsy / tsy) - 0.5f) * ovy;
    float cx = minps(tx + 0.5f, 0.5f - tx) * aspx, cy = minps(ty + 0.5f, 0.5f - ty) * aspy;
    cx = cornersize - minps(cx, cornersize);
    cy = cornersize - minps(cy, cornersize);
    const float dist = __builtin_sqrtf(cx * cx + cy * cy);
    cval = minps(maxps((cornersize - dist) * cornersmooth, 0.0f), 1.0f) * 1.0001f;
  }
  const float rsx = xx * tsx - 0.5f, rsy = (xy * tsy + (ilvec_y - 0.5f)) / ilfac_y;
  const float flx = __builtin_floorf(rsx), fly = __builtin_floorf(rsy);
  const float uvx = rsx - flx;
  float uvy = rsy - fly;
  const float px = (flx + 0.5f) / tsx, py = (fly * ilfac_y + (0.5f - ilvec_y)) / tsy;
  float k[4] = {pi * (1.0f + uvx), pi * uvx, pi * (1.0f - uvx), pi * (2.0f - uvx)};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float c = fix(k[q]);
    k[q] = ((2.0f * sin_(c)) * sin_(c * 0.5f)) / (c * c);
  }
  const float ksum = k[0] + (k[1] + (k[2] + k[3]));
#pragma unroll
  for (int q = 0; q < 4; ++q) k[q] = k[q] / ksum;
  const uint8_t* img = frame_ptr(L.in, z);
  float col[3], col2[3];
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const float ty = row ? py + one_y : py;
    const float tx[4] = {px + -one_x, px, px + one_x, px + 2.0f * one_x};
    float m[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = GENERIC ? sample_rt(L.in, img, tx[q], ty, &lds) : sample<FMT_RGBX8, 0, WRAP_EDGE>(L.in, img, tx[q], ty, &lds);
      m[q][0] = pow_gl(t.x, crt_gamma);
      m[q][1] = pow_gl(t.y, crt_gamma);
      m[q][2] = pow_gl(t.z, crt_gamma);
    }
    float* dst = row ? col2 : col;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dst[c] = minps(maxps(((m[0][c] * k[0] + m[1][c] * k[1]) + m[2][c] * k[2]) + m[3][c] * k[3], 0.0f), 1.0f);
  }
  float w1[3], w2[3], t1[3], t2[3];
  scanline_weights(uvy, col, sw, lum, w1);
  scanline_weights(1.0f - uvy, col2, sw, lum, w2);
  uvy = uvy + 0.333333343f * filter_;
  scanline_weights(uvy, col, sw, lum, t1);
  scanline_weights(__builtin_fabsf(1.0f - uvy), col2, sw, lum, t2);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    w1[c] = (w1[c] + t1[c]) / 3.0f;
    w2[c] = (w2[c] + t2[c]) / 3.0f;
  }
  uvy = uvy - 0.666666687f * filter_;
  scanline_weights(__builtin_fabsf(uvy), col, sw, lum, t1);
  scanline_weights(__builtin_fabsf(1.0f - uvy), col2, sw, lum, t2);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    w1[c] = w1[c] + t1[c] / 3.0f;
    w2[c] = w2[c] + t2[c] / 3.0f;
  }
  const float mf = vary(L.plane[2], x, y, lo);
  const float t = __builtin_floorf(mf - 2.0f * __builtin_floorf(mf / 2.0f));
  const float ma[3] = {1.0f, 1.0f - dotmask, 1.0f}, mb[3] = {1.0f - dotmask, 1.0f, 1.0f - dotmask};
  float res[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float r = (col[c] * w1[c] + col2[c] * w2[c]) * cval;
    r = r * (ma[c] + t * (mb[c] - ma[c]));
    res[c] = pow_gl(r, inv_mon);
  }
  // saturation(): mix(grey, colour, SATURATION) with a uniform weight is grey*(1-t) + colour*t
  const float len = __builtin_sqrtf(res[0] * res[0] + (res[1] * res[1] + res[2] * res[2])) * 0.5775f;
  const bool dark = len < 0.5f;
  const float l0 = dark ? 0.3f * 0.3f + 0.3f * 0.3f : 0.3f, l1 = dark ? 0.6f * 0.6f + 0.6f * 0.6f : 0.6f, l2 = dark ? 0.1f * 0.1f + 0.1f * 0.1f : 0.1f;
  const float grey = res[0] * l0 + (res[1] * l1 + res[2] * l2);
  const float gs = grey * (1.0f - satur);
  const float4 o = make_float4(gs + res[0] * satur, gs + res[1] * satur, gs + res[2] * satur, 1.0f);
  if (GENERIC) store_rt(L, z, x, y, o, &lds);
  else store<FMT_RGBA8>(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

}  // namespace

hipError_t launch_crt_geom(const PassLaunch& L, hipStream_t s) {
  const bool fast = !(L.flags & RC_FLAG_GENERAL_ONLY) && L.in.fmt == FMT_RGBX8 && !L.in.linear && L.in.wrap == WRAP_EDGE &&
                    L.in.n_levels <= 1 && L.out_fmt == FMT_RGBA8;
  if (fast) hipLaunchKernelGGL(k_crt_geom<false>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else hipLaunchKernelGGL(k_crt_geom<true>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}

}  // namespace rck
