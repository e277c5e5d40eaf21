/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * Texture unit and render-target store, as the GL that runs the reference's shaders
 * (Mesa 23.2.1 llvmpipe) behaves for the state the reference sets
 * (ShaderEngine.cpp:1008-1036 filter/wrap on the input texture, :2872-2923 target formats,
 * :944-952 GL_FRAMEBUFFER_SRGB).  Measured facts restated here (oracle/probes/):
 *   - UNORM8 texel -> float: k * (1/255f);  sRGB8 texel: table, decoded before filtering;
 *     GL_RGB source: alpha 1.
 *   - NEAREST: texel floor(s*W) (wrap applied to the integer index).
 *   - LINEAR, float path (sRGB8 / F32 textures, or any texture with clamp_to_border):
 *     u = s*W - 0.5, weights frac(u), lerp = fma(w, b - a, a), x first then y.
 *   - LINEAR, 8-bit fixed-point path (RGBA8 / RGBX8 with edge / repeat / mirror wrap):
 *     weights rint(frac(u)*256), lerp = a + ((w*(b-a) + 128) >> 8) on bytes, x then y,
 *     result k * (1/255f).
 *   - store: UNORM8 = rint(clamp(x,0,1) * 255) (ties to even); sRGB8: monotone table.
 *   - mip-mapped (GL_LINEAR_MIPMAP_LINEAR, measured with textureQueryLOD and level dumps):
 *     rho^2 = max((dsdx*W)^2 + (dtdx*H)^2, (dsdy*W)^2 + (dtdy*H)^2), ONE set of differences per 2x2 quad taken at its
 *     top-left pixel (right - left on the top row, bottom - top in the left column; lp_bld_sample.c
 *     lp_build_packed_ddx_ddy_twocoord).  Where s depends on x alone and t on y alone - every pass but crt-royale's curved
 *     last pass - the differences along the pixel's own row / column are the same floats, and the callers pass those
 *     (pinned either way by tests/golden f32_crt_royale_fake_bloom_geom_*); lod = max(0, 0.5 * (exponent(rho^2) + mantissa(rho^2) - 1)) (a linear "fast
 *     log2"), clamped to the last level; result = fma(frac(lod), S(l+1) - S(l), S(l)) with S = the LINEAR
 *     sample of a level.  glGenerateMipmap = one LINEAR blit per level (sRGB8 decoded / re-encoded).
 *   - GL_NEAREST_MIPMAP_NEAREST (mipmap_input without filter_linear; measured through texture() results against dumped
 *     levels, 64 geometries with razor-edge rho^2, 0 mismatches): the same rho^2 per quad, level =
 *     clamp((exponent(rho^2) + 1) >> 1, 0, last) - the exponent alone, NOT round(lod) of the float above (they differ when
 *     rho^2 is one ulp below an odd power of two) - and the NEAREST texel of that level.
 */
#include <math.h>

#include "rc_oracle.h"
#include <string.h>
#include "rc_tables.inc"

static inline o_vec4 v4(float x, float y, float z, float w) { o_vec4 r = {x, y, z, w}; return r; }

o_vec4 o_texel(const o_tex* t, int x, int y) {
  if (t->fmt == O_FMT_F32) {
    const float* p = (const float*)t->data + ((size_t)y * t->w + x) * 4;
    return v4(p[0], p[1], p[2], p[3]);
  }
  const uint8_t* p = (const uint8_t*)t->data + ((size_t)y * t->w + x) * 4;
  const float k = 1.0f / 255.0f;
  if (t->fmt == O_FMT_SRGB8)
    return v4(o_srgb_decode_table[p[0]], o_srgb_decode_table[p[1]], o_srgb_decode_table[p[2]],
              (float)p[3] * k);
  if (t->fmt == O_FMT_RGBX8) return v4((float)p[0] * k, (float)p[1] * k, (float)p[2] * k, 1.0f);
  return v4((float)p[0] * k, (float)p[1] * k, (float)p[2] * k, (float)p[3] * k);
}

static inline int nan_to_min(float f) { return f != f ? (-2147483647 - 1) : (int)f; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int modi(int v, int n) { int r = v % n; return r < 0 ? r + n : r; }
static inline int mirrori(int v, int n) { int p = modi(v, 2 * n); return p < n ? p : 2 * n - 1 - p; }

/* integer texel index after wrap; returns -1 for "border" */
static inline int wrap_index(int i, int n, int wrap) {
  switch (wrap) {
    case O_WRAP_REPEAT: return modi(i, n);
    case O_WRAP_MIRROR: return mirrori(i, n);
    case O_WRAP_BORDER: return (i < 0 || i >= n) ? -1 : i;
    default: return clampi(i, 0, n - 1);
  }
}

static inline o_vec4 fetch_wrapped(const o_tex* t, int x, int y) {
  int xi = wrap_index(x, t->w, t->wrap), yi = wrap_index(y, t->h, t->wrap);
  if (xi < 0 || yi < 0) return v4(0.f, 0.f, 0.f, 0.f); /* border colour (GL default) */
  return o_texel(t, xi, yi);
}

static inline float lerpf(float w, float a, float b) { return fmaf(w, b - a, a); }

static inline float linear_coord(float s, int n, int wrap) {
  if (wrap == O_WRAP_REPEAT) s = s - floorf(s);
  float u = s * (float)n;
  if (wrap == O_WRAP_EDGE) {
    /* lp_bld_sample_soa.c, linear + CLAMP_TO_EDGE: min(u, n) first (MINPS: n when u is NaN - a NaN coordinate filters at
     * the LAST texel), then - 0.5, then max(.., 0) */
    u = u < (float)n ? u : (float)n;
    u = u - 0.5f;
    return u > 0.0f ? u : 0.0f;
  }
  return u - 0.5f;
}

o_vec4 o_sample(const o_tex* t, float s, float v) {
  if (!t->linear) {
    float fs = s, fv = v;
    if (t->wrap == O_WRAP_REPEAT) { fs = s - floorf(s); fv = v - floorf(v); }
    /* a NaN coordinate converts to INT_MIN (cvttps2dq): texel 0 after clamp-to-edge, the border otherwise */
    int x = nan_to_min(floorf(fs * (float)t->w)), y = nan_to_min(floorf(fv * (float)t->h));
    if (t->wrap == O_WRAP_REPEAT) { x = clampi(x, 0, t->w - 1); y = clampi(y, 0, t->h - 1); }
    return fetch_wrapped(t, x, y);
  }
  /* 8-bit fixed-point filter for RGBA8 / RGBX8 with clamp-to-edge or repeat; clamp-to-border and
   * mirrored-repeat filter in float (measured: tests/golden/wrap_*, probes with random coordinates) */
  int fixed = (t->fmt == O_FMT_RGBA8 || t->fmt == O_FMT_RGBX8) && t->wrap != O_WRAP_BORDER && t->wrap != O_WRAP_MIRROR;
  if (!fixed) {
    float u = linear_coord(s, t->w, t->wrap), w = linear_coord(v, t->h, t->wrap);
    float x0f = floorf(u), y0f = floorf(w);
    float wx = u - x0f, wy = w - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    o_vec4 a = fetch_wrapped(t, x0, y0), b = fetch_wrapped(t, x0 + 1, y0);
    o_vec4 c = fetch_wrapped(t, x0, y0 + 1), d = fetch_wrapped(t, x0 + 1, y0 + 1);
    o_vec4 r;
    r.x = lerpf(wy, lerpf(wx, a.x, b.x), lerpf(wx, c.x, d.x));
    r.y = lerpf(wy, lerpf(wx, a.y, b.y), lerpf(wx, c.y, d.y));
    r.z = lerpf(wy, lerpf(wx, a.z, b.z), lerpf(wx, c.z, d.z));
    r.w = lerpf(wy, lerpf(wx, a.w, b.w), lerpf(wx, c.w, d.w));
    return r;
  }
  /* 8-bit fixed-point path */
  float fs = s, fv = v;
  if (t->wrap == O_WRAP_REPEAT) { fs = s - floorf(s); fv = v - floorf(v); }
  float u = fs * (float)t->w - 0.5f, w = fv * (float)t->h - 0.5f;
  if (t->wrap == O_WRAP_EDGE) {
    u = fminf(fmaxf(u, 0.0f), (float)(t->w - 1));
    w = fminf(fmaxf(w, 0.0f), (float)(t->h - 1));
  }
  float x0f = floorf(u), y0f = floorf(w);
  int wx = (int)rintf((u - x0f) * 256.0f), wy = (int)rintf((w - y0f) * 256.0f);
  int x0 = wrap_index((int)x0f, t->w, t->wrap), x1 = wrap_index((int)x0f + 1, t->w, t->wrap);
  int y0 = wrap_index((int)y0f, t->h, t->wrap), y1 = wrap_index((int)y0f + 1, t->h, t->wrap);
  const uint8_t* base = (const uint8_t*)t->data;
  const uint8_t* p00 = base + ((size_t)y0 * t->w + x0) * 4;
  const uint8_t* p10 = base + ((size_t)y0 * t->w + x1) * 4;
  const uint8_t* p01 = base + ((size_t)y1 * t->w + x0) * 4;
  const uint8_t* p11 = base + ((size_t)y1 * t->w + x1) * 4;
  float out[4];
  for (int c = 0; c < 4; ++c) {
    int a = p00[c], b = p10[c], cc = p01[c], d = p11[c];
    if (c == 3 && t->fmt == O_FMT_RGBX8) a = b = cc = d = 255;
    int top = (a + ((wx * (b - a) + 128) >> 8)) & 255;
    int bot = (cc + ((wx * (d - cc) + 128) >> 8)) & 255;
    int r = (top + ((wy * (bot - top) + 128) >> 8)) & 255;
    out[c] = (float)r * (1.0f / 255.0f);
  }
  return v4(out[0], out[1], out[2], out[3]);
}

/* ---- mip-mapped sampling -------------------------------------------------------------------- */
int o_mip_levels(int w, int h) {
  int n = 1;
  while (w > 1 || h > 1) { w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; ++n; }
  return n < O_MAX_LEVELS ? n : O_MAX_LEVELS;
}

static float fast_log2(float x) {
  union { float f; uint32_t u; } b = {x};
  const int e = (int)((b.u >> 23) & 255u) - 127;
  b.u = (b.u & 0x7fffffu) | 0x3f800000u;
  return (float)e + (b.f - 1.0f);
}

float o_lod_from_quad(const o_tex* t, float s_dx0, float s_dx1, float v_dx0, float v_dx1,
                      float s_dy0, float s_dy1, float v_dy0, float v_dy1) {
  const float fw = (float)t->w, fh = (float)t->h;
  const float ax = (s_dx1 - s_dx0) * fw, bx = (v_dx1 - v_dx0) * fh;
  const float ay = (s_dy1 - s_dy0) * fw, by = (v_dy1 - v_dy0) * fh;
  const float rx = ax * ax + bx * bx, ry = ay * ay + by * by;
  const float rho2 = rx > ry ? rx : ry;
  if (!t->linear) {   /* NEAREST_MIPMAP_NEAREST: the level itself, from the exponent */
    union { float f; uint32_t u; } b = {rho2};
    int lv = ((int)((b.u >> 23) & 255u) - 127 + 1) >> 1;
    if (!(rho2 > 0.0f)) lv = 0;
    if (lv < 0) lv = 0;
    return (float)(lv > t->n_levels - 1 ? t->n_levels - 1 : lv);
  }
  float lod = 0.5f * fast_log2(rho2);
  if (!(lod > 0.0f)) lod = 0.0f;
  const float last = (float)(t->n_levels - 1);
  return lod > last ? last : lod;
}

o_vec4 o_sample_quad(const o_tex* t, float s, float v, float s_dx0, float s_dx1, float v_dx0, float v_dx1,
                     float s_dy0, float s_dy1, float v_dy0, float v_dy1) {
  if (t->n_levels <= 1) return o_sample(t, s, v);
  const float lod = o_lod_from_quad(t, s_dx0, s_dx1, v_dx0, v_dx1, s_dy0, s_dy1, v_dy0, v_dy1);
  const float fl = floorf(lod), w = lod - fl;
  int l0 = (int)fl, l1 = l0 + 1;
  if (l1 > t->n_levels - 1) l1 = t->n_levels - 1;
  o_tex a = *t, b = *t;
  if (!t->linear) {   /* one level, NEAREST */
    a.data = t->mip[l0]; a.w = t->w >> l0 ? t->w >> l0 : 1; a.h = t->h >> l0 ? t->h >> l0 : 1; a.n_levels = 0;
    return o_sample(&a, s, v);
  }
  a.data = t->mip[l0]; a.w = t->w >> l0 ? t->w >> l0 : 1; a.h = t->h >> l0 ? t->h >> l0 : 1; a.n_levels = 0;
  b.data = t->mip[l1]; b.w = t->w >> l1 ? t->w >> l1 : 1; b.h = t->h >> l1 ? t->h >> l1 : 1; b.n_levels = 0;
  const o_vec4 c0 = o_sample(&a, s, v), c1 = o_sample(&b, s, v);
  /* RGBA8 / GL_RGB textures on the 8-bit filter path (edge / repeat wrap): the two level samples are bytes and so is
   * the blend between them - weight floor(frac(lod) * 256), a + ((w (b - a) + 128) >> 8) (measured with a float
   * target at four LODs: every value k/255, 0 mismatches) */
  if (t->linear && (t->fmt == O_FMT_RGBA8 || t->fmt == O_FMT_RGBX8) && t->wrap != O_WRAP_BORDER && t->wrap != O_WRAP_MIRROR) {
    const int w8 = (int)floorf(w * 256.0f);
    const float* p0 = &c0.x; const float* p1 = &c1.x;
    float out[4];
    for (int c = 0; c < 4; ++c) {
      const int a8 = (int)rintf(p0[c] * 255.0f), b8 = (int)rintf(p1[c] * 255.0f);
      out[c] = (float)((a8 + ((w8 * (b8 - a8) + 128) >> 8)) & 255) * (1.0f / 255.0f);
    }
    return v4(out[0], out[1], out[2], out[3]);
  }
  return v4(fmaf(w, c1.x - c0.x, c0.x), fmaf(w, c1.y - c0.y, c0.y), fmaf(w, c1.z - c0.z, c0.z), fmaf(w, c1.w - c0.w, c0.w));
}

void o_gen_mipmaps(const void* level0, int w, int h, int fmt, void* const* dst, int n_levels) {
  const void* src = level0;
  for (int k = 1; k < n_levels; ++k) {
    const int sw = (w >> (k - 1)) ? (w >> (k - 1)) : 1, sh = (h >> (k - 1)) ? (h >> (k - 1)) : 1;
    const int dw = (w >> k) ? (w >> k) : 1, dh = (h >> k) ? (h >> k) : 1;
    o_tex t = {0};
    t.data = src; t.w = sw; t.h = sh; t.fmt = fmt; t.linear = 1; t.wrap = O_WRAP_EDGE;
    o_pass_args a = {0};
    a.in = &t; a.src_w = sw; a.src_h = sh; a.out_w = dw; a.out_h = dh; a.dst = dst[k];
    a.out_fmt = fmt == O_FMT_RGBX8 ? O_FMT_RGBA8 : fmt;   /* GL_RGB: alpha reads 1 at every level */
    a.y0 = 0; a.y1 = dh; a.n_passes = 1; a.vp_w = dw; a.vp_h = dh;
    a.flags = O_FLAG_STOCK_NO_BLIT;
    o_pass_stock(&a);
    src = dst[k];
  }
}

uint8_t o_store_unorm8(float x) {
  if (!(x > 0.0f)) return 0; /* also NaN */
  if (x > 1.0f) x = 1.0f;
  return (uint8_t)rintf(x * 255.0f);
}

/* RSQRTPS of a positive normal float (oracle/rc_rsqrt_table.inc) */
#include "rc_rsqrt_table.inc"
static float rsqrtps_(float x) {
  uint32_t b;
  memcpy(&b, &x, 4);
  const int E = (int)(b >> 23);
  const int k = (E & 1) ? (E - 127) / 2 : (E - 128) / 2;
  const uint32_t rb = (((uint32_t)o_rsqrtps_table[(b >> 13) & 0x7ffu] + 0x7e000u) << 11) - ((uint32_t)k << 23);
  float r;
  memcpy(&r, &rb, 4);
  return r;
}

/* llvmpipe's linear -> sRGB8 conversion (Mesa 23.2.1 gallivm lp_bld_format_srgb.c, lp_build_linear_to_srgb with
 * the fast rsqrt available): clamp to [0,1] (NaN -> 0); x <= 0.0031308: x * (12.92 * 255); else
 * a * x^0.375 + (b * x^0.5 + c) with x^0.5 = x * rsqrt(x), x^0.375 = rsqrt(rsqrt(x * x^0.5)), the b/c term fused;
 * round to nearest even.  Verified against the GL for EVERY float in [0,1] and samples outside it
 * (oracle/probes/srgb_encode_sweep.py): the stored byte is this function of the float, bit for bit. */
uint8_t o_store_srgb8(float x) {
  if (!(x > 0.0f)) return 0; /* also NaN */
  if (x > 1.0f) x = 1.0f;
  if (x <= 0.0031308f) return (uint8_t)rintf(x * (12.92f * 255.0f));
  const float x05 = x * rsqrtps_(x);
  const float t = x05 * x;
  const float x0375 = rsqrtps_(rsqrtps_(t));
  const float a = (float)(0.675f * 1.0622 * 255.0f), b = (float)(0.325f * 1.0622 * 255.0f), c = -0.0620f * 255.0f;
  const float y = a * x0375 + fmaf(b, x05, c);
  return (uint8_t)rintf(y);
}

void o_store_srgb8_array(const float* src, uint8_t* dst, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = o_store_srgb8(src[i]);
}
