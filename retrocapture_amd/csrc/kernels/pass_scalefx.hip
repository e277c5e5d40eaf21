// scalefx/scalefx.glslp (5 passes): reference shaders/shaders_glsl/scalefx/shaders/scalefx-pass{0,1,2,3,4}.glsl,
// the non-GL_ES branch (textureOffset).  Pass 0: colour metric of a texel against four neighbours (RGBA32F);
// pass 1: corner strengths (RGBA32F; SFX_CLR, SFX_SAA); pass 2: junction resolution packed into four bits per
// channel (RGBA8; PassPrev2Texture = pass 0); pass 3: edge levels 1-6 -> sub-pixel tags (RGBA8; SFX_SCN); pass 4:
// 3x output, every sub-pixel a texel of the original frame (PassPrev5Texture) chosen by its tag.
// One thread per target pixel; all lookups are NEAREST texel fetches at integer offsets.  Measured on the GL for
// these files: textureOffset on a NEAREST sampler = texel (floor(coord * size) + offset) then the wrap;
// dist()'s dot(c*d, d) = cd.x*d.x + (cd.y*d.y + cd.z*d.z); comparison chains give exact 0 / 1 floats.
#include "pass_launch.h"

using namespace rcd;

namespace {

__device__ __forceinline__ float minps(float a, float b) { return a < b ? a : b; }   // NaN -> b
__device__ __forceinline__ float maxps(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float mod_glsl(float x, float y) { return x - y * __builtin_floorf(x / y); }

// textureOffset(tex, (u, v), ivec2(dx, dy))
__device__ __forceinline__ float4 tex_off(const Tex& t, const uint8_t* img, float u, float v, int dx, int dy, const SrgbLds* lds) {
  if (t.linear) return sample_rt(t, img, u + (float)dx * (1.0f / (float)t.w), v + (float)dy * (1.0f / (float)t.h), lds);  // (not pinned)
  float s = u, q = v;
  if (t.wrap == WRAP_REPEAT) {
    s = s - __builtin_floorf(s);
    q = q - __builtin_floorf(q);
  }
  int x = (int)__builtin_floorf(s * (float)t.w) + dx, y = (int)__builtin_floorf(q * (float)t.h) + dy;
  switch (t.wrap) {
    case WRAP_BORDER:
      if (x < 0 || y < 0 || x >= t.w || y >= t.h) return make_float4(0.f, 0.f, 0.f, 0.f);
      break;
    case WRAP_REPEAT: x = wrap_index<WRAP_REPEAT>(x, t.w); y = wrap_index<WRAP_REPEAT>(y, t.h); break;
    case WRAP_MIRROR: x = wrap_index<WRAP_MIRROR>(x, t.w); y = wrap_index<WRAP_MIRROR>(y, t.h); break;
    default: x = clampi(x, 0, t.w - 1); y = clampi(y, 0, t.h - 1); break;
  }
  switch (t.fmt) {
    case FMT_SRGB8: return texel<FMT_SRGB8>(t, img, x, y, lds);
    case FMT_RGBX8: return texel<FMT_RGBX8>(t, img, x, y, lds);
    case FMT_F32: return texel<FMT_F32>(t, img, x, y, lds);
    case FMT_F16: return texel<FMT_F16>(t, img, x, y, lds);
    default: return texel<FMT_RGBA8>(t, img, x, y, lds);
  }
}

// ---- pass 0 (FS 118-168) -------------------------------------------------------------------------------------
__device__ __forceinline__ float sfx_dist(float4 A, float4 B) {
  const float r = 0.5f * (A.x + B.x);
  const float dx = A.x - B.x, dy = A.y - B.y, dz = A.z - B.z;
  const float cx = 2.0f + r, cy = 4.0f, cz = 3.0f - r;
  return __builtin_sqrtf((cx * dx) * dx + ((cy * dy) * dy + (cz * dz) * dz)) / 3.0f;
}
__global__ void __launch_bounds__(256) k_scalefx0(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 A = tex_off(L.in, img, u, v, -1, -1, &lds), B = tex_off(L.in, img, u, v, 0, -1, &lds), C = tex_off(L.in, img, u, v, 1, -1, &lds);
  const float4 E = tex_off(L.in, img, u, v, 0, 0, &lds), F = tex_off(L.in, img, u, v, 1, 0, &lds);
  store_rt(L, z, x, y, make_float4(sfx_dist(E, A), sfx_dist(E, B), sfx_dist(E, C), sfx_dist(E, F)), &lds);
  RC_TILE_LOOP_END
}

// ---- pass 1 (FS 120-187): params SFX_CLR, SFX_SAA -------------------------------------------------------------
__device__ __forceinline__ float sfx_str(float d, float ax, float ay, float bx, float by, float clr, float saa) {
  const float diff = ax - ay;
  const float wght1 = maxps(clr - d, 0.0f) / clr;
  const float t = (1.0f - d) + ((minps(ax, bx) + ax > minps(ay, by) + ay) ? diff : -diff);
  const float wght2 = minps(maxps(t, 0.0f), 1.0f);
  return (saa == 1.0f || 2.0f * d < ax + ay) ? (wght1 * wght2) * (ax * ay) : 0.0f;
}
__global__ void __launch_bounds__(256) k_scalefx1(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float clr = L.params[0], saa = L.params[1];
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 A = tex_off(L.in, img, u, v, -1, -1, &lds), B = tex_off(L.in, img, u, v, 0, -1, &lds);
  const float4 D = tex_off(L.in, img, u, v, -1, 0, &lds), E = tex_off(L.in, img, u, v, 0, 0, &lds), F = tex_off(L.in, img, u, v, 1, 0, &lds);
  const float4 G = tex_off(L.in, img, u, v, -1, 1, &lds), H = tex_off(L.in, img, u, v, 0, 1, &lds), I = tex_off(L.in, img, u, v, 1, 1, &lds);
  float4 o;
  o.x = sfx_str(D.z, D.w, E.y, A.w, D.y, clr, saa);
  o.y = sfx_str(F.x, E.w, E.y, B.w, F.y, clr, saa);
  o.z = sfx_str(H.z, E.w, H.y, H.w, I.y, clr, saa);
  o.w = sfx_str(H.x, D.w, H.y, G.w, G.y, clr, saa);
  store_rt(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

// ---- pass 2 (FS 118-233): extra[0] = PassPrev2Texture (pass 0's metric) ------------------------------------------
__device__ __forceinline__ float LE(float x, float y) { return x < y ? 1.0f : 0.0f; }    // 1 - step(y, x)
__device__ __forceinline__ float GE(float x, float y) { return y < x ? 1.0f : 0.0f; }    // 1 - step(x, y)
__device__ __forceinline__ float LEQ(float x, float y) { return y < x ? 0.0f : 1.0f; }   // step(x, y)
__device__ __forceinline__ float NOT(float x) { return 1.0f - x; }
struct F4 { float v[4]; };
__device__ __forceinline__ F4 arr(float4 p) { return F4{{p.x, p.y, p.z, p.w}}; }
// dom(x, y, z, w) on swizzled triples, then the majority vote for ambiguous dominance junctions
__device__ __forceinline__ F4 sfx_vote(const float* x, const float* y, const float* z, const float* w) {
  const float jD[4] = {2.0f * x[1] - (x[0] + x[2]), 2.0f * y[1] - (y[0] + y[2]), 2.0f * z[1] - (z[0] + z[2]), 2.0f * w[1] - (w[0] + w[2])};
  F4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = jD[i], b = jD[(i + 1) & 3], c = jD[(i + 2) & 3], d = jD[(i + 3) & 3];
    r.v[i] = minps(GE(a, 0.0f) * (LEQ(b, 0.0f) * LEQ(d, 0.0f) + GE(a + c, b + d)), 1.0f);
  }
  return r;
}
__device__ __forceinline__ float sfx_clear(float cx, float cy, float ax, float ay, float bx, float by) {
  return (cx >= maxps(minps(ax, ay), minps(bx, by))) && (cy >= maxps(minps(ax, by), minps(bx, ay))) ? 1.0f : 0.0f;
}
__global__ void __launch_bounds__(256) k_scalefx2(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const Tex& M = L.extra[0];
  const uint8_t* mimg = frame_ptr(M, z);
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 A = tex_off(M, mimg, u, v, -1, -1, &lds), B = tex_off(M, mimg, u, v, 0, -1, &lds);
  const float4 D = tex_off(M, mimg, u, v, -1, 0, &lds), E = tex_off(M, mimg, u, v, 0, 0, &lds), F = tex_off(M, mimg, u, v, 1, 0, &lds);
  const float4 G = tex_off(M, mimg, u, v, -1, 1, &lds), H = tex_off(M, mimg, u, v, 0, 1, &lds), I = tex_off(M, mimg, u, v, 1, 1, &lds);
  const F4 As = arr(tex_off(L.in, img, u, v, -1, -1, &lds)), Bs = arr(tex_off(L.in, img, u, v, 0, -1, &lds)), Cs = arr(tex_off(L.in, img, u, v, 1, -1, &lds));
  const F4 Ds = arr(tex_off(L.in, img, u, v, -1, 0, &lds)), Es = arr(tex_off(L.in, img, u, v, 0, 0, &lds)), Fs = arr(tex_off(L.in, img, u, v, 1, 0, &lds));
  const F4 Gs = arr(tex_off(L.in, img, u, v, -1, 1, &lds)), Hs = arr(tex_off(L.in, img, u, v, 0, 1, &lds)), Is = arr(tex_off(L.in, img, u, v, 1, 1, &lds));
  // swizzles as index triples: yzw = 1,2,3; zwx = 2,3,0; wxy = 3,0,1; xyz = 0,1,2
#define SW3(p, i, j, k) {(p).v[i], (p).v[j], (p).v[k]}
  const float jSx[4] = {As.v[2], Bs.v[3], Es.v[0], Ds.v[1]}, jSy[4] = {Bs.v[2], Cs.v[3], Fs.v[0], Es.v[1]};
  const float jSz[4] = {Es.v[2], Fs.v[3], Is.v[0], Hs.v[1]}, jSw[4] = {Ds.v[2], Es.v[3], Hs.v[0], Gs.v[1]};
  const float ax_[3] = SW3(As, 1, 2, 3), bx_[3] = SW3(Bs, 2, 3, 0), ex_[3] = SW3(Es, 3, 0, 1), dx_[3] = SW3(Ds, 0, 1, 2);
  const float by_[3] = SW3(Bs, 1, 2, 3), cy_[3] = SW3(Cs, 2, 3, 0), fy_[3] = SW3(Fs, 3, 0, 1), ey_[3] = SW3(Es, 0, 1, 2);
  const float ez_[3] = SW3(Es, 1, 2, 3), fz_[3] = SW3(Fs, 2, 3, 0), iz_[3] = SW3(Is, 3, 0, 1), hz_[3] = SW3(Hs, 0, 1, 2);
  const float dw_[3] = SW3(Ds, 1, 2, 3), ew_[3] = SW3(Es, 2, 3, 0), hw_[3] = SW3(Hs, 3, 0, 1), gw_[3] = SW3(Gs, 0, 1, 2);
#undef SW3
  const F4 jx = sfx_vote(ax_, bx_, ex_, dx_), jy = sfx_vote(by_, cy_, fy_, ey_), jz = sfx_vote(ez_, fz_, iz_, hz_), jw = sfx_vote(dw_, ew_, hw_, gw_);
  float res[4];
  res[0] = minps(jx.v[2] + NOT(jx.v[1]) * NOT(jx.v[3]) * GE(jSx[2], 0.0f) * (jx.v[0] + GE(jSx[0] + jSx[2], jSx[1] + jSx[3])), 1.0f);
  res[1] = minps(jy.v[3] + NOT(jy.v[2]) * NOT(jy.v[0]) * GE(jSy[3], 0.0f) * (jy.v[1] + GE(jSy[1] + jSy[3], jSy[0] + jSy[2])), 1.0f);
  res[2] = minps(jz.v[0] + NOT(jz.v[3]) * NOT(jz.v[1]) * GE(jSz[0], 0.0f) * (jz.v[2] + GE(jSz[0] + jSz[2], jSz[1] + jSz[3])), 1.0f);
  res[3] = minps(jw.v[1] + NOT(jw.v[0]) * NOT(jw.v[2]) * GE(jSw[1], 0.0f) * (jw.v[3] + GE(jSw[1] + jSw[3], jSw[0] + jSw[2])), 1.0f);
  const float j4[4] = {jx.v[2], jy.v[3], jz.v[0], jw.v[1]};
  const float clr[4] = {sfx_clear(D.z, E.x, D.w, E.y, A.w, D.y), sfx_clear(F.x, E.z, E.w, E.y, B.w, F.y),
                        sfx_clear(H.z, I.x, E.w, H.y, H.w, I.y), sfx_clear(H.x, G.z, D.w, H.y, G.w, G.y)};
  const float h[4] = {minps(D.w, A.w), minps(E.w, B.w), minps(E.w, H.w), minps(D.w, G.w)};
  const float vv[4] = {minps(E.y, D.y), minps(E.y, F.y), minps(H.y, I.y), minps(H.y, G.y)};
  const float hadd[4] = {D.w, E.w, E.w, D.w}, vadd[4] = {E.y, E.y, H.y, H.y};
  float out[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float r2 = minps(res[i] * (j4[i] + NOT(res[(i + 3) & 3] * res[(i + 1) & 3])), 1.0f);   // single pixel & end of line detection
    const float orr = GE(h[i] + hadd[i], vv[i] + vadd[i]);
    const float hori = LE(h[i], vv[i]) * clr[i], vert = GE(h[i], vv[i]) * clr[i];
    out[i] = (r2 + 2.0f * hori + 4.0f * vert + 8.0f * orr) / 15.0f;
  }
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], out[3]), &lds);
  RC_TILE_LOOP_END
}

// ---- pass 3 (FS 119-256): param SFX_SCN ------------------------------------------------------------------------
// four flags of a texel: bit i = floor(mod(component i * mul + add, 2.)) != 0
__device__ __forceinline__ uint32_t sfx_bits(float4 t, float mul, float add) {
  const float c[4] = {t.x, t.y, t.z, t.w};
  uint32_t r = 0u;
#pragma unroll
  for (int i = 0; i < 4; ++i) r |= (__builtin_floorf(mod_glsl(c[i] * mul + add, 2.0f)) != 0.0f ? 1u : 0u) << i;
  return r;
}
#define CORN(t) sfx_bits(t, 15.0f, 0.5f)
#define HORI(t) sfx_bits(t, 7.5f, 0.25f)
#define VERT(t) sfx_bits(t, 3.75f, 0.125f)
#define ORIE(t) sfx_bits(t, 1.875f, 0.0625f)
__global__ void __launch_bounds__(256) k_scalefx3(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const bool scn = L.params[0] == 1.0f;
  enum { X = 0, Y = 1, Z = 2, W = 3 };
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
#define T(dx, dy) tex_off(L.in, img, u, v, dx, dy, &lds)
  const float4 E = T(0, 0), D = T(-1, 0), D0 = T(-2, 0), D1 = T(-3, 0), F = T(1, 0), F0 = T(2, 0), F1 = T(3, 0);
  const float4 B = T(0, -1), B0 = T(0, -2), B1 = T(0, -3), H = T(0, 1), H0 = T(0, 2), H1 = T(0, 3);
#undef T
  const uint32_t Ec = CORN(E), Eh = HORI(E), Ev = VERT(E), Eo = ORIE(E);
  const uint32_t Dc = CORN(D), Dh = HORI(D), Do = ORIE(D), D0c = CORN(D0), D0h = HORI(D0), D1h = HORI(D1);
  const uint32_t Fc = CORN(F), Fh = HORI(F), Fo = ORIE(F), F0c = CORN(F0), F0h = HORI(F0), F1h = HORI(F1);
  const uint32_t Bc = CORN(B), Bv = VERT(B), Bo = ORIE(B), B0c = CORN(B0), B0v = VERT(B0), B1v = VERT(B1);
  const uint32_t Hc = CORN(H), Hv = VERT(H), Ho = ORIE(H), H0c = CORN(H0), H0v = VERT(H0), H1v = VERT(H1);
#define c(b, i) ((((b) >> (i)) & 1u) != 0u)
  const bool lvl1x = c(Ec, X) && (c(Dc, Z) || c(Bc, Z) || scn), lvl1y = c(Ec, Y) && (c(Fc, W) || c(Bc, W) || scn);
  const bool lvl1z = c(Ec, Z) && (c(Fc, X) || c(Hc, X) || scn), lvl1w = c(Ec, W) && (c(Dc, Y) || c(Hc, Y) || scn);
  const bool l2x0 = (c(Ec, X) && c(Eh, Y)) && c(Dc, Z), l2x1 = (c(Ec, Y) && c(Eh, X)) && c(Fc, W);
  const bool l2y0 = (c(Ec, Y) && c(Ev, Z)) && c(Bc, W), l2y1 = (c(Ec, Z) && c(Ev, Y)) && c(Hc, X);
  const bool l2z0 = (c(Ec, W) && c(Eh, Z)) && c(Dc, Y), l2z1 = (c(Ec, Z) && c(Eh, W)) && c(Fc, X);
  const bool l2w0 = (c(Ec, X) && c(Ev, W)) && c(Bc, Z), l2w1 = (c(Ec, W) && c(Ev, X)) && c(Hc, Y);
  const bool l3x0 = l2x1 && (c(Dh, Y) && c(Dh, X)) && c(Fh, Z), l3x1 = l2w1 && (c(Bv, W) && c(Bv, X)) && c(Hv, Z);
  const bool l3y0 = l2x0 && (c(Fh, X) && c(Fh, Y)) && c(Dh, W), l3y1 = l2y1 && (c(Bv, Z) && c(Bv, Y)) && c(Hv, W);
  const bool l3z0 = l2z0 && (c(Fh, W) && c(Fh, Z)) && c(Dh, X), l3z1 = l2y0 && (c(Hv, Y) && c(Hv, Z)) && c(Bv, X);
  const bool l3w0 = l2z1 && (c(Dh, Z) && c(Dh, W)) && c(Fh, Y), l3w1 = l2w0 && (c(Hv, X) && c(Hv, W)) && c(Bv, Y);
  const bool l4x0 = (c(Dc, X) && c(Dh, Y) && c(Eh, X) && c(Eh, Y) && c(Fh, X) && c(Fh, Y)) && (c(D0c, Z) && c(D0h, W));
  const bool l4x1 = (c(Bc, X) && c(Bv, W) && c(Ev, X) && c(Ev, W) && c(Hv, X) && c(Hv, W)) && (c(B0c, Z) && c(B0v, Y));
  const bool l4y0 = (c(Fc, Y) && c(Fh, X) && c(Eh, Y) && c(Eh, X) && c(Dh, Y) && c(Dh, X)) && (c(F0c, W) && c(F0h, Z));
  const bool l4y1 = (c(Bc, Y) && c(Bv, Z) && c(Ev, Y) && c(Ev, Z) && c(Hv, Y) && c(Hv, Z)) && (c(B0c, W) && c(B0v, X));
  const bool l4z0 = (c(Fc, Z) && c(Fh, W) && c(Eh, Z) && c(Eh, W) && c(Dh, Z) && c(Dh, W)) && (c(F0c, X) && c(F0h, Y));
  const bool l4z1 = (c(Hc, Z) && c(Hv, Y) && c(Ev, Z) && c(Ev, Y) && c(Bv, Z) && c(Bv, Y)) && (c(H0c, X) && c(H0v, W));
  const bool l4w0 = (c(Dc, W) && c(Dh, Z) && c(Eh, W) && c(Eh, Z) && c(Fh, W) && c(Fh, Z)) && (c(D0c, Y) && c(D0h, X));
  const bool l4w1 = (c(Hc, W) && c(Hv, X) && c(Ev, W) && c(Ev, X) && c(Bv, W) && c(Bv, X)) && (c(H0c, Y) && c(H0v, Z));
  const bool l5x0 = l4x0 && (c(F0h, X) && c(F0h, Y)) && (c(D1h, Z) && c(D1h, W)), l5x1 = l4y0 && (c(D0h, Y) && c(D0h, X)) && (c(F1h, W) && c(F1h, Z));
  const bool l5y0 = l4y1 && (c(H0v, Y) && c(H0v, Z)) && (c(B1v, W) && c(B1v, X)), l5y1 = l4z1 && (c(B0v, Z) && c(B0v, Y)) && (c(H1v, X) && c(H1v, W));
  const bool l5z0 = l4w0 && (c(F0h, W) && c(F0h, Z)) && (c(D1h, Y) && c(D1h, X)), l5z1 = l4z0 && (c(D0h, Z) && c(D0h, W)) && (c(F1h, X) && c(F1h, Y));
  const bool l5w0 = l4x1 && (c(H0v, X) && c(H0v, W)) && (c(B1v, Z) && c(B1v, Y)), l5w1 = l4w1 && (c(B0v, W) && c(B0v, X)) && (c(H1v, Y) && c(H1v, Z));
  const bool l6x0 = l5x1 && (c(D1h, Y) && c(D1h, X)), l6x1 = l5w1 && (c(B1v, W) && c(B1v, X));
  const bool l6y0 = l5x0 && (c(F1h, X) && c(F1h, Y)), l6y1 = l5y1 && (c(B1v, Z) && c(B1v, Y));
  const bool l6z0 = l5z0 && (c(F1h, W) && c(F1h, Z)), l6z1 = l5y0 && (c(H1v, Y) && c(H1v, Z));
  const bool l6w0 = l5z1 && (c(D1h, Z) && c(D1h, W)), l6w1 = l5w0 && (c(H1v, X) && c(H1v, W));
  float crn[4], mid[4];
  // subpixels - 0 = E, 1 = D, 2 = D0, 3 = F, 4 = F0, 5 = B, 6 = B0, 7 = H, 8 = H0
  crn[0] = ((lvl1x && c(Eo, X)) || (l3x0 && c(Eo, Y)) || (l4x0 && c(Do, X)) || (l6x0 && c(Fo, Y))) ? 5.f : (lvl1x || (l3x1 && !c(Eo, W)) || (l4x1 && !c(Bo, X)) || (l6x1 && !c(Ho, W))) ? 1.f : l3x0 ? 3.f : l3x1 ? 7.f : l4x0 ? 2.f : l4x1 ? 6.f : l6x0 ? 4.f : l6x1 ? 8.f : 0.f;
  crn[1] = ((lvl1y && c(Eo, Y)) || (l3y0 && c(Eo, X)) || (l4y0 && c(Fo, Y)) || (l6y0 && c(Do, X))) ? 5.f : (lvl1y || (l3y1 && !c(Eo, Z)) || (l4y1 && !c(Bo, Y)) || (l6y1 && !c(Ho, Z))) ? 3.f : l3y0 ? 1.f : l3y1 ? 7.f : l4y0 ? 4.f : l4y1 ? 6.f : l6y0 ? 2.f : l6y1 ? 8.f : 0.f;
  crn[2] = ((lvl1z && c(Eo, Z)) || (l3z0 && c(Eo, W)) || (l4z0 && c(Fo, Z)) || (l6z0 && c(Do, W))) ? 7.f : (lvl1z || (l3z1 && !c(Eo, Y)) || (l4z1 && !c(Ho, Z)) || (l6z1 && !c(Bo, Y))) ? 3.f : l3z0 ? 1.f : l3z1 ? 5.f : l4z0 ? 4.f : l4z1 ? 8.f : l6z0 ? 2.f : l6z1 ? 6.f : 0.f;
  crn[3] = ((lvl1w && c(Eo, W)) || (l3w0 && c(Eo, Z)) || (l4w0 && c(Do, W)) || (l6w0 && c(Fo, Z))) ? 7.f : (lvl1w || (l3w1 && !c(Eo, X)) || (l4w1 && !c(Ho, W)) || (l6w1 && !c(Bo, X))) ? 1.f : l3w0 ? 3.f : l3w1 ? 5.f : l4w0 ? 2.f : l4w1 ? 8.f : l6w0 ? 4.f : l6w1 ? 6.f : 0.f;
  mid[0] = ((l2x0 && c(Eo, X)) || (l2x1 && c(Eo, Y)) || (l5x0 && c(Do, X)) || (l5x1 && c(Fo, Y))) ? 5.f : l2x0 ? 1.f : l2x1 ? 3.f : l5x0 ? 2.f : l5x1 ? 4.f : (c(Ec, X) && c(Dc, Z) && c(Ec, Y) && c(Fc, W)) ? (c(Eo, X) ? (c(Eo, Y) ? 5.f : 3.f) : 1.f) : 0.f;
  mid[1] = ((l2y0 && !c(Eo, Y)) || (l2y1 && !c(Eo, Z)) || (l5y0 && !c(Bo, Y)) || (l5y1 && !c(Ho, Z))) ? 3.f : l2y0 ? 5.f : l2y1 ? 7.f : l5y0 ? 6.f : l5y1 ? 8.f : (c(Ec, Y) && c(Bc, W) && c(Ec, Z) && c(Hc, X)) ? (!c(Eo, Y) ? (!c(Eo, Z) ? 3.f : 7.f) : 5.f) : 0.f;
  mid[2] = ((l2z0 && c(Eo, W)) || (l2z1 && c(Eo, Z)) || (l5z0 && c(Do, W)) || (l5z1 && c(Fo, Z))) ? 7.f : l2z0 ? 1.f : l2z1 ? 3.f : l5z0 ? 2.f : l5z1 ? 4.f : (c(Ec, Z) && c(Fc, X) && c(Ec, W) && c(Dc, Y)) ? (c(Eo, Z) ? (c(Eo, W) ? 7.f : 1.f) : 3.f) : 0.f;
  mid[3] = ((l2w0 && !c(Eo, X)) || (l2w1 && !c(Eo, W)) || (l5w0 && !c(Bo, X)) || (l5w1 && !c(Ho, W))) ? 1.f : l2w0 ? 5.f : l2w1 ? 7.f : l5w0 ? 6.f : l5w1 ? 8.f : (c(Ec, W) && c(Hc, Y) && c(Ec, X) && c(Bc, Z)) ? (!c(Eo, W) ? (!c(Eo, X) ? 1.f : 5.f) : 7.f) : 0.f;
#undef c
  store_rt(L, z, x, y, make_float4((crn[0] + 9.0f * mid[0]) / 80.0f, (crn[1] + 9.0f * mid[1]) / 80.0f, (crn[2] + 9.0f * mid[2]) / 80.0f,
                                  (crn[3] + 9.0f * mid[3]) / 80.0f), &lds);
  RC_TILE_LOOP_END
}

// ---- pass 4 (FS 116-177): extra[0] = PassPrev5Texture (the original frame) --------------------------------------
__global__ void __launch_bounds__(256) k_scalefx4(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float ssx = (float)L.in.w, ssy = (float)L.in.h;
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 E = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float e[4] = {E.x, E.y, E.z, E.w};
  float crn[4], mid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    crn[i] = __builtin_floorf(mod_glsl(e[i] * 80.0f + 0.5f, 9.0f));
    mid[i] = __builtin_floorf(mod_glsl(e[i] * 8.888888f + 0.055555f, 9.0f));
  }
  const float px = u * ssx, py = v * ssy;
  const float fx = __builtin_floorf(3.0f * (px - __builtin_floorf(px))), fy = __builtin_floorf(3.0f * (py - __builtin_floorf(py)));
  const float sp = fy == 0.f ? (fx == 0.f ? crn[0] : fx == 1.f ? mid[0] : crn[1])
                             : (fy == 1.f ? (fx == 0.f ? mid[3] : fx == 1.f ? 0.f : mid[1]) : (fx == 0.f ? crn[3] : fx == 1.f ? mid[2] : crn[2]));
  // output coordinate - 0 = E, 1 = D, 2 = D0, 3 = F, 4 = F0, 5 = B, 6 = B0, 7 = H, 8 = H0
  float rx = 0.f, ry = 0.f;
  if (sp == 0.f) { rx = 0.f; ry = 0.f; }
  else if (sp == 1.f) { rx = -1.f; }
  else if (sp == 2.f) { rx = -2.f; }
  else if (sp == 3.f) { rx = 1.f; }
  else if (sp == 4.f) { rx = 2.f; }
  else if (sp == 5.f) { ry = -1.f; }
  else if (sp == 6.f) { ry = -2.f; }
  else if (sp == 7.f) { ry = 1.f; }
  else { ry = 2.f; }
  store_rt(L, z, x, y, sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u + rx / ssx, v + ry / ssy, &lds), &lds);
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {
#define RC_LAUNCH(fn, kernel)                                                                 \
  hipError_t fn(const PassLaunch& L, hipStream_t s) {                                         \
    hipLaunchKernelGGL(kernel, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);         \
    return hipGetLastError();                                                                 \
  }
RC_LAUNCH(launch_scalefx0, k_scalefx0)
RC_LAUNCH(launch_scalefx1, k_scalefx1)
RC_LAUNCH(launch_scalefx2, k_scalefx2)
RC_LAUNCH(launch_scalefx3, k_scalefx3)
RC_LAUNCH(launch_scalefx4, k_scalefx4)
}  // namespace rck
