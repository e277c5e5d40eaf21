// Pass kernel for xbr/shaders/xbr-lv3.glsl (arithmetic spec = the GLSL text: VS lines 77-100,
// FS lines 171-352).  Operation order is the one Mesa's GLSL compiler produces (measured, see
// oracle/rc_passes_ntsc_xbr.c): mat*vec accumulates columns left to right; the five-term
// weighted_distance sum is rebalanced to ((ab+ac)+(de+df))+4gh; the smoothstep numerators are
// (A*fp.y - e0) + B*fp.x; a mix whose weight is step() is a select; with no rule firing the
// undefined pix/blend resolve to the last arm of the if-chain (blend 0).
//
// plane[0..4]: TEX0.x + {-2dx,-dx,0,dx,2dx};  plane[5..9]: TEX0.y + {-2dy,-dy,0,dy,2dy}
// params: XBR_Y_WEIGHT, XBR_EQ_THRESHOLD, XBR_EQ_THRESHOLD2, XBR_LV2_COEFFICIENT, corner_type
#include <cstring>

#include "pass_launch.h"

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
  float maximo[4];
  bool nc[4];
  blend_rule<0>(rec, fpy, fpx, maximo[0], nc[0]);
  blend_rule<1>(rec, fpy, fpx, maximo[1], nc[1]);
  blend_rule<2>(rec, fpy, fpx, maximo[2], nc[2]);
  blend_rule<3>(rec, fpy, fpx, maximo[3], nc[3]);
  const float4 E = texel<FMT_RGBX8>(L.in, img, sx, sy, nullptr);
  const float4 B = fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx, sy - 1, nullptr);
  const float4 D = fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx - 1, sy, nullptr);
  const float4 F = fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx + 1, sy, nullptr);
  const float4 H = fetch_wrapped<FMT_RGBX8, IN_WRAP>(L.in, img, sx, sy + 1, nullptr);
  const float4 pk0 = (rec >> 20) & 1u ? F : H, pk1 = (rec >> 21) & 1u ? B : F, pk2 = (rec >> 22) & 1u ? D : B,
               pk3 = (rec >> 23) & 1u ? H : D;
  const float4 pix1 = nc[0] ? pk0 : nc[1] ? pk1 : nc[2] ? pk2 : pk3;
  const float bl1 = nc[0] ? maximo[0] : nc[1] ? maximo[1] : nc[2] ? maximo[2] : maximo[3];
  const float4 pix2 = nc[3] ? pk3 : nc[2] ? pk2 : nc[1] ? pk1 : pk0;
  const float bl2 = nc[3] ? maximo[3] : nc[2] ? maximo[2] : nc[1] ? maximo[1] : maximo[0];
  const float4 res1 = mix3(E, pix1, bl1), res2 = mix3(E, pix2, bl2);
  const float4 res = c_df(E, res2) < c_df(E, res1) ? res1 : res2;
  *dst = unorm8(res.x) | (unorm8(res.y) << 8) | (unorm8(res.z) << 16) | 0xff000000u;
  RC_TILE_LOOP_END
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
    if (L.in.wrap == WRAP_EDGE) {
      hipLaunchKernelGGL((k_xbr_rules<WRAP_EDGE>), dim3(blocks ? blocks : 1), dim3(256), rcd::srgb_lds_bytes(L), s, L);
      hipLaunchKernelGGL((k_xbr_blend<WRAP_EDGE>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
      if (fix_blocks) hipLaunchKernelGGL((k_xbr_fix<WRAP_EDGE>), dim3(fix_blocks, L.n_frames), dim3(256), rcd::srgb_lds_bytes(L), s, L);
    } else {
      hipLaunchKernelGGL((k_xbr_rules<WRAP_BORDER>), dim3(blocks ? blocks : 1), dim3(256), rcd::srgb_lds_bytes(L), s, L);
      hipLaunchKernelGGL((k_xbr_blend<WRAP_BORDER>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
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
