// Pass kernel for xbr/shaders/xbr-lv3.glsl (arithmetic spec = the GLSL text: VS lines 77-100,
// FS lines 171-352).  Operation order is the one Mesa's GLSL compiler produces (measured, see
// oracle/rc_passes_ntsc_xbr.c): mat*vec accumulates columns left to right; the five-term
// weighted_distance sum is rebalanced to ((ab+ac)+(de+df))+4gh; the smoothstep numerators are
// (A*fp.y - e0) + B*fp.x; a mix whose weight is step() is a select; with no rule firing the
// undefined pix/blend resolve to the last arm of the if-chain (blend 0).
//
// plane[0..4]: TEX0.x + {-2dx,-dx,0,dx,2dx};  plane[5..9]: TEX0.y + {-2dy,-dy,0,dy,2dy}
// params: XBR_Y_WEIGHT, XBR_EQ_THRESHOLD, XBR_EQ_THRESHOLD2, XBR_LV2_COEFFICIENT, corner_type
#include <algorithm>
#include <cstring>
#include <vector>

#include "royale_strip.h"

using namespace rcd;

namespace {

// L.params layout beyond the five shader parameters (ints stored as bit patterns): the target rows
// and columns whose five sampled source rows / columns are not centre-2..centre+2
enum { XBR_P_NROWS = 6, XBR_P_NCOLS = 7, XBR_P_ROWS = 8, XBR_P_COLS = 28, XBR_MAX_IRREGULAR = 20 };

struct F4 { float v[4]; };
struct B4 { bool v[4]; };

__device__ __forceinline__ float lum(const float4 p, const float* w) { return (p.x * w[0] + p.y * w[1]) + p.z * w[2]; }
__device__ __forceinline__ F4 lum4(const float4 a, const float4 b, const float4 c, const float4 d, const float* w) {
  return F4{{lum(a, w), lum(b, w), lum(c, w), lum(d, w)}};
}
__device__ __forceinline__ F4 yzwx(const F4& a) { return F4{{a.v[1], a.v[2], a.v[3], a.v[0]}}; }
__device__ __forceinline__ F4 wxyz(const F4& a) { return F4{{a.v[3], a.v[0], a.v[1], a.v[2]}}; }
__device__ __forceinline__ F4 zwxy(const F4& a) { return F4{{a.v[2], a.v[3], a.v[0], a.v[1]}}; }
__device__ __forceinline__ float df1(float a, float b) { return __builtin_fabsf(a - b); }
// xbr-lv2's seven-term weighted_distance: (((ab + ac) + (de + df)) + (ij + kl)) + 2*gh (oracle/rc_passes_ntsc_xbr.c wd7)
__device__ __forceinline__ F4 wd7(const F4& a, const F4& b, const F4& c, const F4& d, const F4& e, const F4& f, const F4& g,
                                  const F4& h, const F4& i, const F4& j, const F4& k, const F4& l) {
  F4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    r.v[q] = (((df1(a.v[q], b.v[q]) + df1(a.v[q], c.v[q])) + (df1(d.v[q], e.v[q]) + df1(d.v[q], f.v[q]))) +
              (df1(i.v[q], j.v[q]) + df1(k.v[q], l.v[q]))) + 2.0f * df1(g.v[q], h.v[q]);
  return r;
}
__device__ __forceinline__ F4 wd(const F4& a, const F4& b, const F4& c, const F4& d, const F4& e, const F4& f,
                                 const F4& g, const F4& h) {
  F4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    r.v[k] = ((df1(a.v[k], b.v[k]) + df1(a.v[k], c.v[k])) + (df1(d.v[k], e.v[k]) + df1(d.v[k], f.v[k]))) +
             4.0f * df1(g.v[k], h.v[k]);
  return r;
}
// smoothstep(C - 0.4, C + 0.4, A*fy + B*fx) for one component
__device__ __forceinline__ float line_sstep(float A, float B, float C, float fy, float fx) {
  const float e0 = C - 0.4f, e1 = C + 0.4f;
  const float num = ((A == -1.0f && B == -1.0f) || (A == 1.0f && B == 1.0f)) ? (A * fy + B * fx) - e0 : (A * fy - e0) + B * fx;
  float t = num / (e1 - e0);
  t = t > 0.0f ? t : 0.0f;
  t = t < 1.0f ? t : 1.0f;
  return t * (t * (3.0f - 2.0f * t));
}
__device__ __forceinline__ float4 mix3(const float4 a, const float4 b, float t) {
  return make_float4(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), 1.0f);
}
__device__ __forceinline__ float c_df(const float4 a, const float4 b) {
  return (__builtin_fabsf(a.x - b.x) + __builtin_fabsf(a.y - b.y)) + __builtin_fabsf(a.z - b.z);
}

// One target pixel, general form: 21 samples through the five interpolated column / row coordinates.
template <int IN_FMT, int IN_WRAP, bool GENERIC>
__device__ __forceinline__ float4 xbr_pixel(const PassLaunch& L, const SrgbLds* lds_p, int x, int y, int z, bool lo) {
  const float yw = L.params[0], thr = L.params[1], thr2 = L.params[2], lv2 = L.params[3], corner = L.params[4];
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  const float w[3] = {yw * 0.299f, yw * 0.587f, yw * 0.114f};
  float cx[5], cy[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    cx[k] = vary(L.plane[k], x, y, lo);
    cy[k] = vary(L.plane[5 + k], x, y, lo);
  }
  float fpx = cx[2] * tsx, fpy = cy[2] * tsy;
  fpx = fpx - __builtin_floorf(fpx);
  fpy = fpy - __builtin_floorf(fpy);
  const uint8_t* img = frame_ptr(L.in, z);
#define T(i, j) (GENERIC ? sample_rt(L.in, img, cx[i], cy[j], lds_p) : sample<IN_FMT, 0, IN_WRAP>(L.in, img, cx[i], cy[j], lds_p))
  const float4 A1 = T(1, 0), B1 = T(2, 0), C1 = T(3, 0);
  const float4 A = T(1, 1), B = T(2, 1), C = T(3, 1);
  const float4 D = T(1, 2), E = T(2, 2), F = T(3, 2);
  const float4 G = T(1, 3), H = T(2, 3), I = T(3, 3);
  const float4 G5 = T(1, 4), H5 = T(2, 4), I5 = T(3, 4);
  const float4 A0 = T(0, 1), D0 = T(0, 2), G0 = T(0, 3);
  const float4 C4 = T(4, 1), F4_ = T(4, 2), I4 = T(4, 3);
#undef T
  const F4 b = lum4(B, D, H, F, w), c = lum4(C, A, G, I, w);
  const float le = lum(E, w);
  const F4 e = F4{{le, le, le, le}};
  const F4 d = yzwx(b), f = wxyz(b), g = zwxy(c), h = zwxy(b), i = wxyz(c);
  const F4 i4 = lum4(I4, C1, A0, G5, w), i5 = lum4(I5, C4, A1, G0, w), h5 = lum4(H5, F4_, B1, D0, w);
  const F4 f4 = yzwx(h5), c1 = yzwx(i4), g0 = wxyz(i5), b1 = zwxy(h5), d0 = wxyz(h5);
  const F4 wd1 = wd(e, c, g, i, h5, f4, h, f), wd2 = wd(h, d, i5, f, i4, b, e, i);

  const float Ao[4] = {1.0f, -1.0f, -1.0f, 1.0f}, Bo[4] = {1.0f, 1.0f, -1.0f, -1.0f}, Co[4] = {1.5f, 0.5f, -0.5f, 0.5f};
  const float Bx[4] = {0.5f, 2.0f, -0.5f, -2.0f}, Cx[4] = {1.0f, 1.0f, -0.5f, 0.0f};
  const float By[4] = {2.0f, 0.5f, -2.0f, -0.5f}, Cy[4] = {2.0f, 0.0f, -1.0f, 0.5f};
  const float Az[4] = {6.0f, -2.0f, -6.0f, 2.0f}, Bz[4] = {2.0f, 6.0f, -2.0f, -6.0f}, Cz[4] = {5.0f, 3.0f, -3.0f, -1.0f};
  const float Aw[4] = {2.0f, -6.0f, -2.0f, 6.0f}, Bw[4] = {6.0f, 2.0f, -6.0f, -2.0f}, Cw[4] = {5.0f, -1.0f, -3.0f, 3.0f};

  bool nc[4], px[4];
  float maximo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#define EQ(P, Q) (df1(P.v[k], Q.v[k]) < thr)
#define EQ2(P, Q) (df1(P.v[k], Q.v[k]) < thr2)
    const bool ne = (e.v[k] != f.v[k]) && (e.v[k] != h.v[k]);
    bool r1;
    if (corner == 1.0f) {
      r1 = ne;
    } else if (corner == 2.0f) {
      bool t = !EQ(f, b) && !EQ(h, d);
      t = t || EQ(e, i);
      t = t && !EQ(f, i4);
      t = t && !EQ(h, i5);
      t = t || EQ(e, g);
      t = t || EQ(e, c);
      r1 = ne && t;
    } else {
      const bool t1 = (!EQ(f, b) && !EQ(f, c)) || (!EQ(h, d) && !EQ(h, g));
      const bool t2 = EQ(e, i) && ((!EQ(f, f4) && !EQ(f, i4)) || (!EQ(h, h5) && !EQ(h, i5)));
      const bool t3 = EQ(e, g) || EQ(e, c);
      r1 = ne && (t1 || (t2 || t3));
    }
    const bool r2_left = (e.v[k] != g.v[k]) && (d.v[k] != g.v[k]);
    const bool r2_up = (e.v[k] != c.v[k]) && (b.v[k] != c.v[k]);
    const bool r3_left = EQ2(g, g0) && !EQ2(d0, g0);
    const bool r3_up = EQ2(c, c1) && !EQ2(b1, c1);
#undef EQ
#undef EQ2
    const float dfg = df1(f.v[k], g.v[k]), dhc = df1(h.v[k], c.v[k]);
    const bool edr = (wd1.v[k] < wd2.v[k]) && r1;
    const bool edr_left = (lv2 * dfg <= dhc) && r2_left;
    const bool edr_up = (dfg >= lv2 * dhc) && r2_up;
    float m = 0.0f;
    bool any = false;
    if (edr) {  // every new-colour rule needs edr; without it all five finals are 0
      const float fx45 = line_sstep(Ao[k], Bo[k], Co[k], fpy, fpx);
      const bool nc45 = fx45 != 0.0f;
      float f30 = 0.0f, f60 = 0.0f, f15 = 0.0f, f75 = 0.0f;
      bool nc30 = false, nc60 = false, nc15 = false, nc75 = false;
      if (edr_left) {
        const float fx30 = line_sstep(Ao[k], Bx[k], Cx[k], fpy, fpx);
        nc30 = fx30 != 0.0f;
        f30 = nc30 ? fx30 : 0.0f;
        if (r3_left) {
          const float fx15 = line_sstep(Az[k], Bz[k], Cz[k], fpy, fpx);
          nc15 = fx15 != 0.0f;
          f15 = nc15 ? fx15 : 0.0f;
        }
      }
      if (edr_up) {
        const float fx60 = line_sstep(Ao[k], By[k], Cy[k], fpy, fpx);
        nc60 = fx60 != 0.0f;
        f60 = nc60 ? fx60 : 0.0f;
        if (r3_up) {
          const float fx75 = line_sstep(Aw[k], Bw[k], Cw[k], fpy, fpx);
          nc75 = fx75 != 0.0f;
          f75 = nc75 ? fx75 : 0.0f;
        }
      }
      const float f45 = nc45 ? fx45 : 0.0f;
      const float m1 = f15 > f75 ? f15 : f75, m2 = f30 > f60 ? f30 : f60;
      const float m3 = m1 > m2 ? m1 : m2;
      m = m3 > f45 ? m3 : f45;
      any = nc75 || nc15 || nc30 || nc60 || nc45;
    }
    maximo[k] = m;
    nc[k] = any;
    px[k] = df1(e.v[k], f.v[k]) <= df1(e.v[k], h.v[k]);
  }
  const float4 pk0 = px[0] ? F : H, pk1 = px[1] ? B : F, pk2 = px[2] ? D : B, pk3 = px[3] ? H : D;
  const float4 pix1 = nc[0] ? pk0 : nc[1] ? pk1 : nc[2] ? pk2 : pk3;
  const float bl1 = nc[0] ? maximo[0] : nc[1] ? maximo[1] : nc[2] ? maximo[2] : maximo[3];
  const float4 pix2 = nc[3] ? pk3 : nc[2] ? pk2 : nc[1] ? pk1 : pk0;
  const float bl2 = nc[3] ? maximo[3] : nc[2] ? maximo[2] : nc[1] ? maximo[1] : maximo[0];
  const float4 res1 = mix3(E, pix1, bl1), res2 = mix3(E, pix2, bl2);
  float4 res = c_df(E, res2) < c_df(E, res1) ? res1 : res2;
  res.w = 1.0f;
  return res;
}

template <int IN_FMT, int IN_WRAP, int OUT_FMT, bool GENERIC>
__global__ void __launch_bounds__(256) k_xbr_lv3(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float4 res = xbr_pixel<IN_FMT, IN_WRAP, GENERIC>(L, &lds, x, y, z, lo);
  if (GENERIC) store_rt(L, z, x, y, res, &lds);
  else store<OUT_FMT>(L, z, x, y, res, &lds);
  RC_TILE_LOOP_END
}

// ----------------------------------------------------------------------------- xbr-lv2 ------
// xbr/shaders/xbr-lv2.glsl FS 260-361 (CORNER_C, SMOOTH_TIPS, both branches of small_details), same 5x5 coordinate set.
// Restated as in oracle/rc_passes_ntsc_xbr.c, including what llvmpipe makes of the shader's unassigned `f4`
// (reads as `i` in wd1; eq(f, f4) true; |x - f4| = 0 in the small_details distances); byte-exact against llvmpipe.
// params: XBR_SCALE (unused: a commented-out pragma the reference's scan still lists), XBR_Y_WEIGHT, XBR_EQ_THRESHOLD,
// XBR_LV1_COEFFICIENT, XBR_LV2_COEFFICIENT, small_details
// SHARED: the two 45-degree lines share s = B*fx + (A*fy + delta); the others fold (delta - C) first (oracle line_clamp)
template <bool SHARED>
__device__ __forceinline__ float lv2_line(float A, float B, float dl, float C, float ci, float fy, float fx) {
  const float cc = ci != 0.0f ? C + ci : C;
  const float num = SHARED ? (B * fx + (A * fy + dl)) - cc : B * fx + (A * fy + (dl - cc));
  float t = num / (2.0f * dl);
  t = t > 0.0f ? t : 0.0f;
  return t < 1.0f ? t : 1.0f;
}
__device__ __forceinline__ float dot_rgbw(const float4 p) { return p.x * 14.352f + (p.y * 28.176f + p.z * 5.472f); }

// GENERIC false: RGBX8 / RGBA8 NEAREST clamp-to-edge source and a plain RGBA8 target (the shipped preset)
template <int IN_FMT, bool GENERIC, bool DETAILS>
__global__ void __launch_bounds__(256) k_xbr_lv2(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float thr = L.params[2], lv2 = L.params[4];
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  float cx[5], cy[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    cx[k] = vary(L.plane[k], x, y, lo);
    cy[k] = vary(L.plane[5 + k], x, y, lo);
  }
  float fpx = cx[2] * tsx, fpy = cy[2] * tsy;
  fpx = fpx - __builtin_floorf(fpx);
  fpy = fpy - __builtin_floorf(fpy);
  const uint8_t* img = frame_ptr(L.in, z);
#define T(i, j) (GENERIC ? sample_rt(L.in, img, cx[i], cy[j], &lds) : sample<IN_FMT, 0, WRAP_EDGE>(L.in, img, cx[i], cy[j], &lds))
  const float4 A1 = T(1, 0), C1 = T(3, 0);
  const float4 A = T(1, 1), B = T(2, 1), C = T(3, 1);
  const float4 D = T(1, 2), E = T(2, 2), F = T(3, 2);
  const float4 G = T(1, 3), H = T(2, 3), I = T(3, 3);
  const float4 G5 = T(1, 4), H5 = T(2, 4), I5 = T(3, 4);
  const float4 A0 = T(0, 1), D0 = T(0, 2), G0 = T(0, 3);
  const float4 C4 = T(4, 1), F4_ = T(4, 2), I4 = T(4, 3), B1 = T(2, 0);
#undef T
  const F4 b = F4{{dot_rgbw(B), dot_rgbw(D), dot_rgbw(H), dot_rgbw(F)}}, c = F4{{dot_rgbw(C), dot_rgbw(A), dot_rgbw(G), dot_rgbw(I)}};
  const float le = dot_rgbw(E);
  const F4 e = F4{{le, le, le, le}};
  const F4 d = yzwx(b), f = wxyz(b), g = zwxy(c), h = zwxy(b), i = wxyz(c);
  F4 i4 = F4{{dot_rgbw(I4), dot_rgbw(C1), dot_rgbw(A0), dot_rgbw(G5)}}, i5 = F4{{dot_rgbw(I5), dot_rgbw(C4), dot_rgbw(A1), dot_rgbw(G0)}};
  F4 h5 = F4{{dot_rgbw(H5), dot_rgbw(F4_), dot_rgbw(B1), dot_rgbw(D0)}};
  F4 wd1 = wd(e, c, g, i, h5, i, h, f), wd2 = wd(h, d, i5, f, i4, b, e, i);
  if (DETAILS) {   // small_details >= 0.5 (FS 285-290, 320-323): outer lumas with XBR_Y_WEIGHT * Y, seven-term distance
    const float y0 = L.params[1] * 0.2126f, y1 = L.params[1] * 0.7152f, y2 = L.params[1] * 0.0722f;
    auto ly = [&](const float4 p) { return y0 * p.x + (y1 * p.y + y2 * p.z); };
    i4 = F4{{ly(I4), ly(C1), ly(A0), ly(G5)}};
    i5 = F4{{ly(I5), ly(C4), ly(A1), ly(G0)}};
    h5 = F4{{ly(H5), ly(F4_), ly(B1), ly(D0)}};
    wd1 = wd7(e, c, g, i, i, h5, h, f, b, d, i4, i5);   // the unassigned f4: |x - f4| = 0 in both calls (oracle)
    wd2 = wd7(h, d, i5, f, b, i4, e, i, g, h5, c, c);
  }
  const float Ao[4] = {1.0f, -1.0f, -1.0f, 1.0f}, Bo[4] = {1.0f, 1.0f, -1.0f, -1.0f}, Co[4] = {1.5f, 0.5f, -0.5f, 0.5f};
  const float Bx[4] = {0.5f, 2.0f, -0.5f, -2.0f}, Cx[4] = {1.0f, 1.0f, -0.5f, 0.0f};
  const float By[4] = {2.0f, 0.5f, -2.0f, -0.5f}, Cy[4] = {2.0f, 0.0f, -1.0f, 0.5f};
  const float third = 1.0f / 3.0f, sixth = 0.5f / 3.0f;
  const float dl[4] = {sixth, third, sixth, third}, du[4] = {third, sixth, third, sixth};
  float maximos[4];
  bool px[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#define EQ(p, q) (df1(p.v[k], q.v[k]) <= thr)
    const bool ne = e.v[k] != f.v[k] && e.v[k] != h.v[k];
    const bool t1 = (!EQ(f, b) && !EQ(f, c)) || (!EQ(h, d) && !EQ(h, g));
    const bool t2 = EQ(e, i) && (!EQ(h, h5) && !EQ(h, i5));
    const bool t3 = EQ(e, g) || EQ(e, c);
#undef EQ
    const bool r1 = ne && (t1 || t2 || t3);
    const bool r2l = e.v[k] != g.v[k] && d.v[k] != g.v[k], r2u = e.v[k] != c.v[k] && b.v[k] != c.v[k];
    const float dfg = df1(f.v[k], g.v[k]), dhc = df1(h.v[k], c.v[k]);
    const bool edri = wd1.v[k] <= wd2.v[k] && ne;
    const bool edr = wd1.v[k] + 0.1f <= wd2.v[k] && r1;
    const bool edr_l = lv2 * dfg <= dhc && r2l && edr, edr_u = lv2 * dhc <= dfg && r2u && edr;
    const float f45 = edr ? lv2_line<true>(Ao[k], Bo[k], third, Co[k], 0.0f, fpy, fpx) : 0.0f;
    const float f45i = edri ? lv2_line<true>(Ao[k], Bo[k], third, Co[k], 0.25f, fpy, fpx) : 0.0f;
    const float f30 = edr_l ? lv2_line<false>(Ao[k], Bx[k], dl[k], Cx[k], 0.0f, fpy, fpx) : 0.0f;
    const float f60 = edr_u ? lv2_line<false>(Ao[k], By[k], du[k], Cy[k], 0.0f, fpy, fpx) : 0.0f;
    px[k] = df1(e.v[k], f.v[k]) <= df1(e.v[k], h.v[k]);
    const float m1 = f30 > f60 ? f30 : f60, m2 = f45 > f45i ? f45 : f45i;
    maximos[k] = m1 > m2 ? m1 : m2;
  }
  float4 res1 = mix3(E, px[0] ? F : H, maximos[0]);
  res1 = mix3(res1, px[2] ? D : B, maximos[2]);
  float4 res2 = mix3(E, px[1] ? B : F, maximos[1]);
  res2 = mix3(res2, px[3] ? H : D, maximos[3]);
  float4 res = c_df(E, res2) < c_df(E, res1) ? res1 : res2;
  res.w = 0.0f;   // FragColor.xyz only: alpha is never written and comes out 0 on llvmpipe
  if (GENERIC) store_rt(L, z, x, y, res, &lds);
  else store<FMT_RGBA8>(L, z, x, y, res, &lds);
  RC_TILE_LOOP_END
}

// The general form on the few target rows / columns whose sampling pattern is irregular (listed in
// L.params by the host, see below); launched after k_xbr_blend on the same stream.
// grid.x = (rows * ceil(out_w/256) + cols * ceil(out_h/256)), grid.y = frames
template <int IN_WRAP>
__global__ void __launch_bounds__(256) k_xbr_fix(const PassLaunch L) {
  const SrgbLds* none = nullptr;  // RGBX8 in, RGBA8 out: no sRGB tables involved
  const int n_rows = __float_as_int(L.params[XBR_P_NROWS]), n_cols = __float_as_int(L.params[XBR_P_NCOLS]);
  const int bx = (L.out_w + 255) >> 8, by = (L.out_h + 255) >> 8;
  const int z = blockIdx.y;
  int b = blockIdx.x, x, y;
  if (b < n_rows * bx) {
    y = __float_as_int(L.params[XBR_P_ROWS + b / bx]);
    x = (b % bx) * 256 + threadIdx.x;
  } else {
    b -= n_rows * bx;
    if (b >= n_cols * by) return;
    x = __float_as_int(L.params[XBR_P_COLS + b / by]);
    y = (b % by) * 256 + threadIdx.x;
  }
  if (x >= L.out_w || y >= L.out_h) return;
  const float4 res = xbr_pixel<FMT_RGBX8, IN_WRAP, false>(L, none, x, y, z, lower_tri(x, y, L.out_w, L.out_h));
  store<FMT_RGBA8>(L, z, x, y, res, none);
}


// ---- two-launch form --------------------------------------------------------------------------
// Everything except the five smoothstep line tests depends only on the 21-texel neighbourhood of
// the source pixel an output pixel falls in, not on where inside it the output pixel lies.  When
// the host has checked (kernel_registry.cpp, xbrNeighbourhoodIsRegular) that for every target
// column / row the five sampled columns / rows are exactly centre-2 .. centre+2, the rules are
// evaluated once per SOURCE pixel into a 24-bit record (k_xbr_rules) and the per-target-pixel
// kernel (k_xbr_blend) only evaluates the line tests of the rules that fired.  Same arithmetic,
// same results; ~225x fewer rule evaluations at 15x magnification.
//
// record bits, k = 0..3: [k] edr, [4+k] edr && edr_left, [8+k] edr && edr_up,
//   [12+k] edr && edr_left && lv3_left, [16+k] edr && edr_up && lv3_up, [20+k] px
template <int IN_WRAP>
__global__ void __launch_bounds__(256) k_xbr_rules(const PassLaunch L) {
  const float yw = L.params[0], thr = L.params[1], thr2 = L.params[2], lv2 = L.params[3], corner = L.params[4];
  const float w[3] = {yw * 0.299f, yw * 0.587f, yw * 0.114f};
  const int W = L.in.w, n = L.in.w * L.in.h, total = n * L.n_frames;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int z = idx / n, r = idx - z * n, sy = r / W, sx = r - sy * W;
    const uint8_t* img = frame_ptr(L.in, z);
#define T(i, j) fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx + (i) - 2, sy + (j) - 2, nullptr)
    const float4 A1 = T(1, 0), B1 = T(2, 0), C1 = T(3, 0);
    const float4 A = T(1, 1), B = T(2, 1), C = T(3, 1);
    const float4 D = T(1, 2), E = T(2, 2), F = T(3, 2);
    const float4 G = T(1, 3), H = T(2, 3), I = T(3, 3);
    const float4 G5 = T(1, 4), H5 = T(2, 4), I5 = T(3, 4);
    const float4 A0 = T(0, 1), D0 = T(0, 2), G0 = T(0, 3);
    const float4 C4 = T(4, 1), F4_ = T(4, 2), I4 = T(4, 3);
#undef T
    const F4 b = lum4(B, D, H, F, w), c = lum4(C, A, G, I, w);
    const float le = lum(E, w);
    const F4 e = F4{{le, le, le, le}};
    const F4 d = yzwx(b), f = wxyz(b), g = zwxy(c), h = zwxy(b), i = wxyz(c);
    const F4 i4 = lum4(I4, C1, A0, G5, w), i5 = lum4(I5, C4, A1, G0, w), h5 = lum4(H5, F4_, B1, D0, w);
    const F4 f4 = yzwx(h5), c1 = yzwx(i4), g0 = wxyz(i5), b1 = zwxy(h5), d0 = wxyz(h5);
    const F4 wd1 = wd(e, c, g, i, h5, f4, h, f), wd2 = wd(h, d, i5, f, i4, b, e, i);
    uint32_t rec = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#define EQ(P, Q) (df1(P.v[k], Q.v[k]) < thr)
#define EQ2(P, Q) (df1(P.v[k], Q.v[k]) < thr2)
      const bool ne = (e.v[k] != f.v[k]) && (e.v[k] != h.v[k]);
      bool r1;
      if (corner == 1.0f) {
        r1 = ne;
      } else if (corner == 2.0f) {
        bool t = !EQ(f, b) && !EQ(h, d);
        t = t || EQ(e, i);
        t = t && !EQ(f, i4);
        t = t && !EQ(h, i5);
        t = t || EQ(e, g);
        t = t || EQ(e, c);
        r1 = ne && t;
      } else {
        const bool t1 = (!EQ(f, b) && !EQ(f, c)) || (!EQ(h, d) && !EQ(h, g));
        const bool t2 = EQ(e, i) && ((!EQ(f, f4) && !EQ(f, i4)) || (!EQ(h, h5) && !EQ(h, i5)));
        const bool t3 = EQ(e, g) || EQ(e, c);
        r1 = ne && (t1 || (t2 || t3));
      }
      const bool r2_left = (e.v[k] != g.v[k]) && (d.v[k] != g.v[k]);
      const bool r2_up = (e.v[k] != c.v[k]) && (b.v[k] != c.v[k]);
      const bool r3_left = EQ2(g, g0) && !EQ2(d0, g0);
      const bool r3_up = EQ2(c, c1) && !EQ2(b1, c1);
#undef EQ
#undef EQ2
      const float dfg = df1(f.v[k], g.v[k]), dhc = df1(h.v[k], c.v[k]);
      const bool edr = (wd1.v[k] < wd2.v[k]) && r1;
      const bool el = edr && (lv2 * dfg <= dhc) && r2_left;
      const bool eu = edr && (dfg >= lv2 * dhc) && r2_up;
      const bool px = df1(e.v[k], f.v[k]) <= df1(e.v[k], h.v[k]);
      rec |= (edr ? 1u : 0u) << k | (el ? 1u : 0u) << (4 + k) | (eu ? 1u : 0u) << (8 + k) |
             ((el && r3_left) ? 1u : 0u) << (12 + k) | ((eu && r3_up) ? 1u : 0u) << (16 + k) | (px ? 1u : 0u) << (20 + k);
    }
    reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.scratch) + L.scratch_frame_stride * (uint64_t)z)[r] = rec;
  }
}

// smoothstep(C - 0.4, C + 0.4, A*fy + B*fx) with compile-time A, B, C: the division by the folded
// (e1 - e0) uses div_const_ (exhaustively checked exact for the four widths that occur)
template <int K, int LINE>
__device__ __forceinline__ float line_sstep_c(float fy, float fx) {
  constexpr float A_[5][4] = {{1.0f, -1.0f, -1.0f, 1.0f}, {1.0f, -1.0f, -1.0f, 1.0f}, {1.0f, -1.0f, -1.0f, 1.0f},
                              {6.0f, -2.0f, -6.0f, 2.0f}, {2.0f, -6.0f, -2.0f, 6.0f}};
  constexpr float B_[5][4] = {{1.0f, 1.0f, -1.0f, -1.0f}, {0.5f, 2.0f, -0.5f, -2.0f}, {2.0f, 0.5f, -2.0f, -0.5f},
                              {2.0f, 6.0f, -2.0f, -6.0f}, {6.0f, 2.0f, -6.0f, -2.0f}};
  constexpr float C_[5][4] = {{1.5f, 0.5f, -0.5f, 0.5f}, {1.0f, 1.0f, -0.5f, 0.0f}, {2.0f, 0.0f, -1.0f, 0.5f},
                              {5.0f, 3.0f, -3.0f, -1.0f}, {5.0f, -1.0f, -3.0f, 3.0f}};
  constexpr float A = A_[LINE][K], B = B_[LINE][K], C = C_[LINE][K];
  constexpr float e0 = C - 0.4f, e1 = C + 0.4f, d = e1 - e0, rd = 1.0f / d;
  const float num = ((A == -1.0f && B == -1.0f) || (A == 1.0f && B == 1.0f)) ? (A * fy + B * fx) - e0 : (A * fy - e0) + B * fx;
  float t = div_const_(num, d, rd);
  t = t > 0.0f ? t : 0.0f;
  t = t < 1.0f ? t : 1.0f;
  return t * (t * (3.0f - 2.0f * t));
}

typedef float xbrs_rgb_t __attribute__((ext_vector_type(4)));   // a colour as a vector value (a float4 struct behind a select goes through memory)
__device__ __forceinline__ xbrs_rgb_t xbrs_rgb(const float4 c) { return xbrs_rgb_t{c.x, c.y, c.z, 0.0f}; }
__device__ __forceinline__ xbrs_rgb_t xbrs_mix(const xbrs_rgb_t a, const xbrs_rgb_t b, float t) { return xbrs_rgb_t{a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), 0.0f}; }   // mix3
__device__ __forceinline__ float xbrs_cdf(const xbrs_rgb_t a, const xbrs_rgb_t b) { return (__builtin_fabsf(a.x - b.x) + __builtin_fabsf(a.y - b.y)) + __builtin_fabsf(a.z - b.z); }   // c_df
template <int K>
__device__ __forceinline__ void blend_rule(uint32_t rec, float fpy, float fpx, float& m, bool& any) {
  m = 0.0f;
  any = false;
  if (rec & (1u << K)) {
    const float fx45 = line_sstep_c<K, 0>(fpy, fpx);
    float f30 = 0.0f, f60 = 0.0f, f15 = 0.0f, f75 = 0.0f;
    if (rec & (1u << (4 + K))) {
      f30 = line_sstep_c<K, 1>(fpy, fpx);
      if (rec & (1u << (12 + K))) f15 = line_sstep_c<K, 3>(fpy, fpx);
    }
    if (rec & (1u << (8 + K))) {
      f60 = line_sstep_c<K, 2>(fpy, fpx);
      if (rec & (1u << (16 + K))) f75 = line_sstep_c<K, 4>(fpy, fpx);
    }
    // final = float(nc) * fx with nc = rule && (fx != 0): equals fx itself (0 stays 0)
    const float m1 = f15 > f75 ? f15 : f75, m2 = f30 > f60 ? f30 : f60;
    const float m3 = m1 > m2 ? m1 : m2;
    m = m3 > fx45 ? m3 : fx45;
    any = (f75 != 0.0f) || (f15 != 0.0f) || (f30 != 0.0f) || (f60 != 0.0f) || (fx45 != 0.0f);
  }
}

// plane[2] = TEX0.x, plane[7] = TEX0.y; RGBX8 nearest source, RGBA8 target
template <int IN_WRAP>
__global__ void __launch_bounds__(256) k_xbr_blend(const PassLaunch L) {
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  RC_TILE_LOOP_BEGIN
  const float ux = vary(L.plane[2], x, y, lo) * tsx, uy = vary(L.plane[7], x, y, lo) * tsy;
  const float fx0 = __builtin_floorf(ux), fy0 = __builtin_floorf(uy);
  const float fpx = ux - fx0, fpy = uy - fy0;
  const int sx = (int)fx0, sy = (int)fy0;
  const uint8_t* img = frame_ptr(L.in, z);
  const uint32_t rec =
      reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(L.scratch) + L.scratch_frame_stride * (uint64_t)z)[sy * L.in.w + sx];
  uint32_t* dst = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z) + ((size_t)y * L.out_w + x);
  if ((rec & 15u) == 0u) {
    // no rule fires: both blends are 0 and the result is E itself; unorm8(k/255) = k
    *dst = *reinterpret_cast<const uint32_t*>(img + ((size_t)sy * L.in.w + sx) * 4) | 0xff000000u;
    continue;
  }
  float m0, m1, m2, m3;
  bool n0, n1, n2, n3;
  blend_rule<0>(rec, fpy, fpx, m0, n0);
  blend_rule<1>(rec, fpy, fpx, m1, n1);
  blend_rule<2>(rec, fpy, fpx, m2, n2);
  blend_rule<3>(rec, fpy, fpx, m3, n3);
  const xbrs_rgb_t E = xbrs_rgb(texel<FMT_RGBX8>(L.in, img, sx, sy, nullptr));
  const xbrs_rgb_t B = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx, sy - 1, nullptr));
  const xbrs_rgb_t D = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx - 1, sy, nullptr));
  const xbrs_rgb_t F = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx + 1, sy, nullptr));
  const xbrs_rgb_t H = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx, sy + 1, nullptr));
  // (colours as vector values and the four maxima as scalars: float4 structs and arrays behind these selects went through scratch)
  const xbrs_rgb_t pk0 = (rec >> 20) & 1u ? F : H, pk1 = (rec >> 21) & 1u ? B : F, pk2 = (rec >> 22) & 1u ? D : B, pk3 = (rec >> 23) & 1u ? H : D;
  const xbrs_rgb_t pix1 = n0 ? pk0 : n1 ? pk1 : n2 ? pk2 : pk3;
  const float bl1 = n0 ? m0 : n1 ? m1 : n2 ? m2 : m3;
  const xbrs_rgb_t pix2 = n3 ? pk3 : n2 ? pk2 : n1 ? pk1 : pk0;
  const float bl2 = n3 ? m3 : n2 ? m2 : n1 ? m1 : m0;
  const xbrs_rgb_t res1 = xbrs_mix(E, pix1, bl1), res2 = xbrs_mix(E, pix2, bl2);
  const xbrs_rgb_t res = xbrs_cdf(E, res2) < xbrs_cdf(E, res1) ? res1 : res2;
  *dst = unorm8(res.x) | (unorm8(res.y) << 8) | (unorm8(res.z) << 16) | 0xff000000u;
  RC_TILE_LOOP_END
}


// ---- the blend, one SOURCE pixel per lane -----------------------------------------------------------------------
// k_xbr_blend renders one target pixel per lane: 64 consecutive target pixels of a row are four source pixels at fifteen
// different sub-pixel positions, so a wave runs every line test of every rule any of them fired.  Here a lane owns a source
// pixel - its record, its five texels and the four candidate colours stay in registers - and the wave walks the sub-pixel
// positions together: phase i of target row y is column X0[s] + i for source column s.  All lanes then sit at (nearly) the same
// point of the pixel, and a line test whose smoothstep is exactly 0 there for EVERY column of that phase - or exactly 1 - is
// known before it is computed: per (target row, phase) two 20-bit masks, built once per geometry from the line's value at the
// extreme fractional coordinates of the phase's columns (the test is monotone in the coordinate: every operation of
// line_sstep_c is).  At 256 x 224 -> 3840 x 2160 nine of the twenty tests are alive on average, 2.5 of them saturated.
// A wave assembles its 64 x 15 pixels of a target row in LDS and stores them 16 bytes per lane.  Same arithmetic per pixel as
// k_xbr_blend (the maxima of non-negative terms in another order), same bytes.
constexpr int kXsMaxPhases = 32;   // target columns per source column the form handles
struct XbrSrcTables {
  float* fx = nullptr;      // [out_w] fractional part of the centre coordinate of a target column ...
  float* fy = nullptr;      // [out_h] ... and row
  int* sx = nullptr;        // [out_w] its source column ...
  int* sy = nullptr;        // [out_h] ... and row
  int* x0 = nullptr;        // [in.w + 1] first target column of a source column (x0[in.w] = out_w)
  int* y0 = nullptr;        // [in.h + 1]
  uint32_t* lim = nullptr;  // [kXsMaxPhases][2] smallest / largest fx (float bits) over the columns of a phase
  uint2* masks = nullptr;   // [out_h][n_phases] {tests that are 0 for every column of the phase, tests that are 1}
  int n_phases = 0;
  bool usable = false;
  void release() {
    for (void* p : {(void*)fx, (void*)fy, (void*)sx, (void*)sy, (void*)x0, (void*)y0, (void*)lim, (void*)masks})
      if (p) (void)hipFree(p);
    *this = XbrSrcTables();
  }
};
__global__ void __launch_bounds__(256) k_xbrs_axes(const PassLaunch L, float* fx, float* fy, int* sx, int* sy, int* x0, int* y0, uint32_t* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  if (i < L.out_w) {
    const float u = vary(L.plane[2], i, 0, true) * tsx, f0 = __builtin_floorf(u);
    fx[i] = u - f0;
    sx[i] = (int)f0;
    const float up = i > 0 ? __builtin_floorf(vary(L.plane[2], i - 1, 0, true) * tsx) : -1.0f;
    if (!(f0 >= 0.0f && f0 < tsx) || f0 < up || f0 > up + 1.0f) atomicOr(bad, 1u);   // every source column in turn, none skipped
    else if (f0 != up) x0[(int)f0] = i;
    if (i == L.out_w - 1) {
      x0[L.in.w] = L.out_w;
      if ((int)f0 != L.in.w - 1) atomicOr(bad, 1u);
    }
  }
  if (i < L.out_h) {
    const float v = vary(L.plane[7], 0, i, true) * tsy, f0 = __builtin_floorf(v);
    fy[i] = v - f0;
    sy[i] = (int)f0;
    const float up = i > 0 ? __builtin_floorf(vary(L.plane[7], 0, i - 1, true) * tsy) : -1.0f;
    if (!(f0 >= 0.0f && f0 < tsy) || f0 < up || f0 > up + 1.0f) atomicOr(bad, 2u);
    else if (f0 != up) y0[(int)f0] = i;
    if (i == L.out_h - 1) {
      y0[L.in.h] = L.out_h;
      if ((int)f0 != L.in.h - 1) atomicOr(bad, 2u);
    }
  }
}
__global__ void __launch_bounds__(256) k_xbrs_limits(const PassLaunch L, const float* fx, const int* sx, const int* x0, uint32_t* lim) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= L.out_w) return;
  const int ph = x - x0[sx[x]];
  if (ph < 0 || ph >= kXsMaxPhases) return;
  atomicMin(&lim[2 * ph], f2bits(fx[x]));   // (non-negative floats order like their bits)
  atomicMax(&lim[2 * ph + 1], f2bits(fx[x]));
}
template <int K, int LINE>
__device__ __forceinline__ void xbrs_mask_bit(float fy, float lo, float hi, uint32_t* dead, uint32_t* sat) {
  constexpr float B_[5][4] = {{1.0f, 1.0f, -1.0f, -1.0f}, {0.5f, 2.0f, -0.5f, -2.0f}, {2.0f, 0.5f, -2.0f, -0.5f},
                              {2.0f, 6.0f, -2.0f, -6.0f}, {6.0f, 2.0f, -6.0f, -2.0f}};   // (line_sstep_c's B: the test rises with fx where B > 0)
  constexpr bool rising = B_[LINE][K] > 0.0f;
  if (line_sstep_c<K, LINE>(fy, rising ? hi : lo) == 0.0f) *dead |= 1u << (4 * LINE + K);
  if (line_sstep_c<K, LINE>(fy, rising ? lo : hi) == 1.0f) *sat |= 1u << (4 * LINE + K);
}
__global__ void __launch_bounds__(256) k_xbrs_masks(const PassLaunch L, const float* fy, const uint32_t* lim, int n_phases, uint2* masks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L.out_h * n_phases) return;
  const int y = i / n_phases, ph = i - y * n_phases;
  const float v = fy[y], lo = bits2f(lim[2 * ph]), hi = bits2f(lim[2 * ph + 1]);
  uint32_t dead = 0u, sat = 0u;
#define RC_XM(K) xbrs_mask_bit<K, 0>(v, lo, hi, &dead, &sat); xbrs_mask_bit<K, 1>(v, lo, hi, &dead, &sat); xbrs_mask_bit<K, 2>(v, lo, hi, &dead, &sat); \
                 xbrs_mask_bit<K, 3>(v, lo, hi, &dead, &sat); xbrs_mask_bit<K, 4>(v, lo, hi, &dead, &sat);
  RC_XM(0) RC_XM(1) RC_XM(2) RC_XM(3)
#undef RC_XM
  masks[i] = make_uint2(dead, sat);
}

// line_sstep_c once more, split the way the source-major kernel walks: the row's part of the numerator (A fy - e0; A fy for the
// two 45-degree lines whose numerator is (A fy + B fx) - e0) is formed once per target row, the column's part B fx once per phase
// for the four magnitudes of B (a sign is an operand modifier); the clamp is one v_med3 and 3 - 2 t one fma (2 t is exact).  The
// same IEEE operations on the same values as line_sstep_c.
struct XbrsLineConst {
  float A, B, e0, d;
  bool paired;   // (A, B) = (1, 1) or (-1, -1)
};
template <int K, int LINE>
__device__ __forceinline__ constexpr XbrsLineConst xbrs_const() {
  constexpr float A_[5][4] = {{1.0f, -1.0f, -1.0f, 1.0f}, {1.0f, -1.0f, -1.0f, 1.0f}, {1.0f, -1.0f, -1.0f, 1.0f},
                              {6.0f, -2.0f, -6.0f, 2.0f}, {2.0f, -6.0f, -2.0f, 6.0f}};
  constexpr float B_[5][4] = {{1.0f, 1.0f, -1.0f, -1.0f}, {0.5f, 2.0f, -0.5f, -2.0f}, {2.0f, 0.5f, -2.0f, -0.5f},
                              {2.0f, 6.0f, -2.0f, -6.0f}, {6.0f, 2.0f, -6.0f, -2.0f}};
  constexpr float C_[5][4] = {{1.5f, 0.5f, -0.5f, 0.5f}, {1.0f, 1.0f, -0.5f, 0.0f}, {2.0f, 0.0f, -1.0f, 0.5f},
                              {5.0f, 3.0f, -3.0f, -1.0f}, {5.0f, -1.0f, -3.0f, 3.0f}};
  constexpr float A = A_[LINE][K], B = B_[LINE][K], C = C_[LINE][K];
  return XbrsLineConst{A, B, C - 0.4f, (C + 0.4f) - (C - 0.4f), (A == -1.0f && B == -1.0f) || (A == 1.0f && B == 1.0f)};
}
template <int K, int LINE>
__device__ __forceinline__ float xbrs_row_term(float fy) {   // per target row
  constexpr XbrsLineConst c = xbrs_const<K, LINE>();
  return c.paired ? c.A * fy : c.A * fy - c.e0;
}
typedef float xbrs_row_t __attribute__((ext_vector_type(8)));   // the five row terms of one rule (a vector: stays in registers)
struct XbrsPhase {   // per phase and lane: |B| fx for the four magnitudes
  float q1, qh, q2, q6;
  // ... and, for the whole kernel, the four smoothstep widths that occur with their reciprocals, held in VECTOR registers: as
  // literals they would reach the fused operations of the division through scalar registers, and a scalar operand makes a VALU
  // instruction issue at the slow rate
  float d[4], rd[4];
};
__device__ __forceinline__ float xbrs_max(float a, float b) {   // v_max_f32 as it is (no operand is ever a NaN: no canonicalising copies)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int K, int LINE>
__device__ __forceinline__ float xbrs_line_value(float row_term, const XbrsPhase& q) {
  constexpr XbrsLineConst c = xbrs_const<K, LINE>();
  constexpr float mag = c.B < 0.0f ? -c.B : c.B;
  const float bq = mag == 1.0f ? q.q1 : (mag == 0.5f ? q.qh : (mag == 2.0f ? q.q2 : q.q6));
  const float bfx = c.B < 0.0f ? -bq : bq;
  const float num = c.paired ? (row_term + bfx) - c.e0 : row_term + bfx;
  constexpr int w = c.d == 0.79999995f ? 0 : (c.d == 0.8f ? 1 : (c.d == 0.80000007f ? 2 : 3));
  static_assert(c.d == 0.79999995f || c.d == 0.8f || c.d == 0.80000007f || c.d == 0.8000002f, "smoothstep widths");
  // div_const_(num, d, 1 / d)
  const float qt = num * q.rd[w];
  const float t = __builtin_amdgcn_fmed3f(fma_(fma_(-q.d[w], qt, num), q.rd[w], qt), 0.0f, 1.0f);
  return t * (t * fma_(-2.0f, t, 3.0f));
}

// one rule: the largest of its enabled line tests (k_xbr_blend's blend_rule; `any` there is m != 0: every term is >= 0)
template <int K, int LINE>
__device__ __forceinline__ float xbrs_line(float m, uint32_t dead, uint32_t sat, uint32_t rec, float row_term, const XbrsPhase& q) {
  constexpr uint32_t bit = 1u << (4 * LINE + K);
  if (dead & bit) return m;   // (uniform)
  const float t = (sat & bit) ? 1.0f : xbrs_line_value<K, LINE>(row_term, q);
  // the test counts where the source pixel's record enables it: t & (all ones or zero from the record's bit)
  return xbrs_max(m, bits2f(f2bits(t) & (uint32_t)__builtin_amdgcn_sbfe((int)rec, 4u * LINE + K, 1u)));
}
template <int K>
__device__ __forceinline__ float xbrs_rule(uint32_t dead, uint32_t sat, uint32_t rec, const xbrs_row_t row, const XbrsPhase& q) {
  if ((dead & (0x11111u << K)) == (0x11111u << K)) return 0.0f;   // (uniform) all five tests of the rule are 0 here
  float m = 0.0f;
  m = xbrs_line<K, 0>(m, dead, sat, rec, row[0], q);
  m = xbrs_line<K, 1>(m, dead, sat, rec, row[1], q);
  m = xbrs_line<K, 2>(m, dead, sat, rec, row[2], q);
  m = xbrs_line<K, 3>(m, dead, sat, rec, row[3], q);
  m = xbrs_line<K, 4>(m, dead, sat, rec, row[4], q);
  return m;
}
// One target pixel given the rules R (bit k: rule k has a test alive at this phase, uniform).  k_xbr_blend's selection - pix1 / bl1
// from the first rule in the order 0 1 2 3 whose maximum is not 0, pix2 / bl2 from the first in the order 3 2 1 0, rule 3 resp. 0 as
// they stand when none is - over the live rules only: a rule that is not alive has maximum 0, and a blend weight of 0 leaves E
// whatever colour it is paired with (E + 0 (pix - E) = E), so the chain may end on the last live rule instead.
template <uint32_t R>
__device__ __forceinline__ xbrs_rgb_t xbrs_pixel(uint32_t dead, uint32_t sat, uint32_t rec, const xbrs_row_t r0, const xbrs_row_t r1, const xbrs_row_t r2,
                                                 const xbrs_row_t r3, const XbrsPhase& q, const xbrs_rgb_t E, const xbrs_rgb_t pk0, const xbrs_rgb_t pk1,
                                                 const xbrs_rgb_t pk2, const xbrs_rgb_t pk3) {
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  if constexpr (R & 1u) m0 = xbrs_rule<0>(dead, sat, rec, r0, q);
  if constexpr (R & 2u) m1 = xbrs_rule<1>(dead, sat, rec, r1, q);
  if constexpr (R & 4u) m2 = xbrs_rule<2>(dead, sat, rec, r2, q);
  if constexpr (R & 8u) m3 = xbrs_rule<3>(dead, sat, rec, r3, q);
  constexpr int first = (R & 1u) ? 0 : (R & 2u) ? 1 : (R & 4u) ? 2 : 3, last = (R & 8u) ? 3 : (R & 4u) ? 2 : (R & 2u) ? 1 : 0;
  // ascending order, ending on the last live rule; descending order, ending on the first
  xbrs_rgb_t pix1 = last == 3 ? pk3 : last == 2 ? pk2 : last == 1 ? pk1 : pk0, pix2 = first == 0 ? pk0 : first == 1 ? pk1 : first == 2 ? pk2 : pk3;
  float bl1 = last == 3 ? m3 : last == 2 ? m2 : last == 1 ? m1 : m0, bl2 = first == 0 ? m0 : first == 1 ? m1 : first == 2 ? m2 : m3;
#define RC_XS1(k, mk, pkk) if constexpr (((R >> k) & 1u) && k != last) { const bool n = mk != 0.0f; pix1 = n ? pkk : pix1; bl1 = n ? mk : bl1; }
  RC_XS1(3, m3, pk3) RC_XS1(2, m2, pk2) RC_XS1(1, m1, pk1) RC_XS1(0, m0, pk0)
#undef RC_XS1
#define RC_XS2(k, mk, pkk) if constexpr (((R >> k) & 1u) && k != first) { const bool n = mk != 0.0f; pix2 = n ? pkk : pix2; bl2 = n ? mk : bl2; }
  RC_XS2(0, m0, pk0) RC_XS2(1, m1, pk1) RC_XS2(2, m2, pk2) RC_XS2(3, m3, pk3)
#undef RC_XS2
  if constexpr (first == last) return xbrs_mix(E, pix1, bl1);   // one live rule: both selections are it
  const xbrs_rgb_t res1 = xbrs_mix(E, pix1, bl1), res2 = xbrs_mix(E, pix2, bl2);
  return xbrs_cdf(E, res2) < xbrs_cdf(E, res1) ? res1 : res2;
}
// unorm8 of three channels into one texel: x * 255 converted with saturation and round-to-nearest-even in one instruction per
// channel (v_cvt_pk_u8_f32; NaN gives 0, as unorm8's)
__device__ __forceinline__ uint32_t xbrs_pack(float r, float g, float b) {
  uint32_t px = 0xff000000u;
  asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(px) : "v"(r * 255.0f));
  asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(px) : "v"(g * 255.0f));
  asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(px) : "v"(b * 255.0f));
  return px;
}

// one wave = 64 source pixels of one source row of one frame; a workgroup = four waves.  Dynamic LDS per wave: the row of target
// pixels being assembled and the fx of its columns (64 * n_phases words each), then the row's masks (2 * n_phases words)
template <int IN_WRAP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_xbr_blend_src(const PassLaunch L, const float* __restrict__ gfx, const float* __restrict__ gfy, const int* __restrict__ gx0,
                                                      const int* __restrict__ gy0, const uint2* __restrict__ gmasks, int n_phases) {
  extern __shared__ uint32_t rc_dyn_lds_[];
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int per_wave = 2 * 64 * n_phases + 2 * kXsMaxPhases;
  uint32_t* rowbuf = rc_dyn_lds_ + wave * per_wave;
  float* fxbuf = reinterpret_cast<float*>(rowbuf + 64 * n_phases);
  uint32_t* mbuf = rowbuf + 2 * 64 * n_phases;
  const int groups = (L.in.w + 63) >> 6;
  const long item = (long)blockIdx.x * 4 + wave, n_items = (long)L.n_frames * L.in.h * groups;
  if (item >= n_items) return;
  const int z = (int)(item / ((long)L.in.h * groups)), rem = (int)(item - (long)z * L.in.h * groups);
  const int sy = rem / groups, s0 = (rem - sy * groups) * 64;
  const int s = s0 + lane;
  const bool live = s < L.in.w;
  const int sc = live ? s : L.in.w - 1;
  const uint8_t* img = frame_ptr(L.in, z);
  const uint32_t rec = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(L.scratch) + L.scratch_frame_stride * (uint64_t)z)[sy * L.in.w + sc];
  const xbrs_rgb_t E = xbrs_rgb(texel<FMT_RGBX8>(L.in, img, sc, sy, nullptr));
  const xbrs_rgb_t B = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sc, sy - 1, nullptr));
  const xbrs_rgb_t D = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sc - 1, sy, nullptr));
  const xbrs_rgb_t F = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sc + 1, sy, nullptr));
  const xbrs_rgb_t H = xbrs_rgb(fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sc, sy + 1, nullptr));
  const xbrs_rgb_t pk0 = (rec >> 20) & 1u ? F : H, pk1 = (rec >> 21) & 1u ? B : F, pk2 = (rec >> 22) & 1u ? D : B, pk3 = (rec >> 23) & 1u ? H : D;
  const int xbase = gx0[s0], span = gx0[min(s0 + 64, L.in.w)] - xbase;   // (uniform) this wave's target columns
  const int xs = gx0[sc] - xbase, cnt = gx0[sc + 1] - gx0[sc];
  for (int j = lane; j < span; j += 64) fxbuf[j] = gfx[xbase + j];
  float wd0 = 0.79999995f, wd1 = 0.8f, wd2 = 0.80000007f, wd3 = 0.8000002f;   // (XbrsPhase::d, rd)
  float wr0 = 1.0f / 0.79999995f, wr1 = 1.0f / 0.8f, wr2 = 1.0f / 0.80000007f, wr3 = 1.0f / 0.8000002f;
  asm volatile("" : "+v"(wd0), "+v"(wd1), "+v"(wd2), "+v"(wd3), "+v"(wr0), "+v"(wr1), "+v"(wr2), "+v"(wr3));
  uint32_t* out_frame = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z);
  const int y_end = gy0[sy + 1];
  for (int y = gy0[sy]; y < y_end; ++y) {
    const float fy = gfy[y];
    xbrs_row_t r0, r1, r2, r3;   // the row's part of every test's numerator
#define RC_XR(K, r) r[0] = xbrs_row_term<K, 0>(fy); r[1] = xbrs_row_term<K, 1>(fy); r[2] = xbrs_row_term<K, 2>(fy); r[3] = xbrs_row_term<K, 3>(fy); r[4] = xbrs_row_term<K, 4>(fy);
    RC_XR(0, r0) RC_XR(1, r1) RC_XR(2, r2) RC_XR(3, r3)
#undef RC_XR
    if (lane < n_phases) {
      const uint2 m = gmasks[(size_t)y * n_phases + lane];
      mbuf[2 * lane] = m.x;
      mbuf[2 * lane + 1] = m.y;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < n_phases; ++i) {
      const uint32_t dead = __builtin_amdgcn_readfirstlane(mbuf[2 * i]), sat = __builtin_amdgcn_readfirstlane(mbuf[2 * i + 1]);
      const bool active = live && i < cnt;
      const float fx = fxbuf[active ? xs + i : 0];
      const XbrsPhase q = {fx, 0.5f * fx, 2.0f * fx, 6.0f * fx, {wd0, wd1, wd2, wd3}, {wr0, wr1, wr2, wr3}};
      // the rules with a test alive at this phase (uniform); the others' maxima are 0 and drop out of the selection below
      const uint32_t nd = ~dead;
      const uint32_t live_rules = ((nd | (nd >> 4) | (nd >> 8) | (nd >> 12) | (nd >> 16)) & 15u);
      xbrs_rgb_t res;
      switch (live_rules) {
#define RC_XT(R) case R: res = xbrs_pixel<R>(dead, sat, rec, r0, r1, r2, r3, q, E, pk0, pk1, pk2, pk3); break;
        RC_XT(1) RC_XT(2) RC_XT(3) RC_XT(4) RC_XT(5) RC_XT(6) RC_XT(7) RC_XT(8) RC_XT(9) RC_XT(10) RC_XT(11) RC_XT(12) RC_XT(13) RC_XT(14) RC_XT(15)
#undef RC_XT
        default: res = E; break;   // no test alive: both blends are 0
      }
      if (active) rowbuf[xs + i] = xbrs_pack(res.x, res.y, res.z);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    uint32_t* dst = out_frame + (size_t)y * L.out_w + xbase;
    if (((xbase | span) & 3) == 0) {   // (uniform) whole 16-byte groups
      for (int j = lane * 4; j < span; j += 256) *reinterpret_cast<uint4*>(dst + j) = *reinterpret_cast<const uint4*>(rowbuf + j);
    } else {
      for (int j = lane; j < span; j += 64) dst[j] = rowbuf[j];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

void buildXbrSrcTables(const PassLaunch& L, hipStream_t s, XbrSrcTables* T) {
  const Plane &pu = L.plane[2], &pv = L.plane[7];
  // the centre coordinate of a column must not depend on the row or the triangle, nor a row's on the column
  if (pu.dy_lo != 0.0f || pu.dy_up != 0.0f || pv.dx_lo != 0.0f || pv.dx_up != 0.0f || pu.a0_lo != pu.a0_up || pu.dx_lo != pu.dx_up ||
      pv.a0_lo != pv.a0_up || pv.dy_lo != pv.dy_up || L.in.w < 1 || L.in.h < 1)
    return;
  uint32_t* bad = nullptr;
  const size_t W = (size_t)L.out_w, H = (size_t)L.out_h;
  bool ok = hipMalloc((void**)&T->fx, W * 4) == hipSuccess && hipMalloc((void**)&T->fy, H * 4) == hipSuccess && hipMalloc((void**)&T->sx, W * 4) == hipSuccess &&
            hipMalloc((void**)&T->sy, H * 4) == hipSuccess && hipMalloc((void**)&T->x0, ((size_t)L.in.w + 1) * 4) == hipSuccess &&
            hipMalloc((void**)&T->y0, ((size_t)L.in.h + 1) * 4) == hipSuccess && hipMalloc((void**)&T->lim, kXsMaxPhases * 8) == hipSuccess &&
            hipMalloc((void**)&bad, 4) == hipSuccess;
  std::vector<int> hx0((size_t)L.in.w + 1), hy0((size_t)L.in.h + 1);
  uint32_t hbad = 1;
  if (ok)
    ok = hipMemsetAsync(bad, 0, 4, s) == hipSuccess && hipMemsetAsync(T->x0, 0xff, ((size_t)L.in.w + 1) * 4, s) == hipSuccess &&
         hipMemsetAsync(T->y0, 0xff, ((size_t)L.in.h + 1) * 4, s) == hipSuccess;
  if (ok) {
    const int n = std::max(L.out_w, L.out_h);
    hipLaunchKernelGGL(k_xbrs_axes, dim3((n + 255) / 256), dim3(256), 0, s, L, T->fx, T->fy, T->sx, T->sy, T->x0, T->y0, bad);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(hx0.data(), T->x0, hx0.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(hy0.data(), T->y0, hy0.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (bad) (void)hipFree(bad);
  int nph = 0;
  if (ok && hbad == 0) {
    for (size_t i = 0; i + 1 < hx0.size() && ok; ++i) {
      ok = hx0[i] >= 0 && hx0[i + 1] > hx0[i];
      nph = std::max(nph, hx0[i + 1] - hx0[i]);
    }
    for (size_t i = 0; i + 1 < hy0.size() && ok; ++i) ok = hy0[i] >= 0 && hy0[i + 1] > hy0[i];
    ok = ok && nph >= 1 && nph <= kXsMaxPhases;
  } else {
    ok = false;
  }
  if (ok) {
    std::vector<uint32_t> init((size_t)kXsMaxPhases * 2);
    for (int i = 0; i < kXsMaxPhases; ++i) {
      init[(size_t)2 * i] = 0x7f800000u;   // +inf: above every fx
      init[(size_t)2 * i + 1] = 0u;
    }
    ok = hipMalloc((void**)&T->masks, H * (size_t)nph * sizeof(uint2)) == hipSuccess &&
         hipMemcpyAsync(T->lim, init.data(), init.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
      hipLaunchKernelGGL(k_xbrs_limits, dim3((L.out_w + 255) / 256), dim3(256), 0, s, L, T->fx, T->sx, T->x0, T->lim);
      hipLaunchKernelGGL(k_xbrs_masks, dim3((unsigned)((H * (size_t)nph + 255) / 256)), dim3(256), 0, s, L, T->fy, T->lim, nph, T->masks);
      ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;   // (`init` must outlive the copy)
    }
  }
  if (!ok) {
    T->release();
    return;
  }
  T->n_phases = nph;
  T->usable = true;
}

}  // namespace

namespace rck {

hipError_t launch_xbr_lv2(const PassLaunch& L, hipStream_t s) {
  const bool fast = !L.in.linear && L.in.wrap == WRAP_EDGE && L.out_fmt == FMT_RGBA8 && !(L.flags & RC_FLAG_GENERAL_ONLY);
  const bool details = !(L.params[5] < 0.5f);   // "Preserve Small Details"
#define RC_XBR2(FMT, GEN)                                                                                                          \
  do {                                                                                                                             \
    if (details) hipLaunchKernelGGL((k_xbr_lv2<FMT, GEN, true>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);              \
    else hipLaunchKernelGGL((k_xbr_lv2<FMT, GEN, false>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);                     \
  } while (0)
  if (fast && L.in.fmt == FMT_RGBX8) RC_XBR2(FMT_RGBX8, false);
  else if (fast && L.in.fmt == FMT_RGBA8) RC_XBR2(FMT_RGBA8, false);
  else RC_XBR2(FMT_RGBA8, true);
#undef RC_XBR2
  return hipGetLastError();
}
hipError_t launch_xbr_lv3(const PassLaunch& L, hipStream_t s) {
  const bool shipped = L.in.fmt == FMT_RGBX8 && !L.in.linear && L.out_fmt == FMT_RGBA8 &&
                       (L.in.wrap == WRAP_EDGE || L.in.wrap == WRAP_BORDER);
  if (shipped && (L.flags & RC_FLAG_XBR_REGULAR) && !(L.flags & RC_FLAG_GENERAL_ONLY) && L.scratch) {
    const long n = (long)L.in.w * L.in.h * L.n_frames;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    int n_rows, n_cols;
    std::memcpy(&n_rows, &L.params[XBR_P_NROWS], 4);
    std::memcpy(&n_cols, &L.params[XBR_P_NCOLS], 4);
    const unsigned fix_blocks = (unsigned)(n_rows * ((L.out_w + 255) / 256) + n_cols * ((L.out_h + 255) / 256));
    // the blend with one source pixel per lane where the geometry allows it (XbrSrcTables), else one target pixel per lane
    static std::mutex mu;
    static std::map<rcstrip::GeoKey, rcstrip::GeoCached<XbrSrcTables>> cache;
    const auto T = rcstrip::geo_tables<XbrSrcTables>(L, s, mu, cache, buildXbrSrcTables);
    const long items = (long)L.n_frames * L.in.h * ((L.in.w + 63) / 64);
    const unsigned src_lds = T ? (unsigned)(4 * (2 * 64 * T->n_phases + 2 * kXsMaxPhases) * 4) : 0u;
    if (L.in.wrap == WRAP_EDGE) {
      hipLaunchKernelGGL((k_xbr_rules<WRAP_EDGE>), dim3(blocks ? blocks : 1), dim3(256), rcd::srgb_lds_bytes(L), s, L);
      if (T) hipLaunchKernelGGL((k_xbr_blend_src<WRAP_EDGE>), dim3((unsigned)((items + 3) / 4)), dim3(256), src_lds, s, L, T->fx, T->fy, T->x0, T->y0, T->masks, T->n_phases);
      else hipLaunchKernelGGL((k_xbr_blend<WRAP_EDGE>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
      if (fix_blocks) hipLaunchKernelGGL((k_xbr_fix<WRAP_EDGE>), dim3(fix_blocks, L.n_frames), dim3(256), rcd::srgb_lds_bytes(L), s, L);
    } else {
      hipLaunchKernelGGL((k_xbr_rules<WRAP_BORDER>), dim3(blocks ? blocks : 1), dim3(256), rcd::srgb_lds_bytes(L), s, L);
      if (T) hipLaunchKernelGGL((k_xbr_blend_src<WRAP_BORDER>), dim3((unsigned)((items + 3) / 4)), dim3(256), src_lds, s, L, T->fx, T->fy, T->x0, T->y0, T->masks, T->n_phases);
      else hipLaunchKernelGGL((k_xbr_blend<WRAP_BORDER>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
      if (fix_blocks) hipLaunchKernelGGL((k_xbr_fix<WRAP_BORDER>), dim3(fix_blocks, L.n_frames), dim3(256), rcd::srgb_lds_bytes(L), s, L);
    }
    return hipGetLastError();
  }
  // general form (any sampler state / target format, or an irregular sampling pattern): nearest on the RGB source frame, RGBA8 viewport-sized target
  if (L.in.fmt == FMT_RGBX8 && !L.in.linear && L.in.wrap == WRAP_EDGE && L.out_fmt == FMT_RGBA8)
    hipLaunchKernelGGL((k_xbr_lv3<FMT_RGBX8, WRAP_EDGE, FMT_RGBA8, false>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else if (L.in.fmt == FMT_RGBX8 && !L.in.linear && L.in.wrap == WRAP_BORDER && L.out_fmt == FMT_RGBA8)
    hipLaunchKernelGGL((k_xbr_lv3<FMT_RGBX8, WRAP_BORDER, FMT_RGBA8, false>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else
    hipLaunchKernelGGL((k_xbr_lv3<0, 0, 0, true>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}

}  // namespace rck
