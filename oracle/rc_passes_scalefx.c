/* TEST INFRASTRUCTURE (oracle/): CPU restatement of scalefx/scalefx.glslp (5 passes), reference files
 * shaders/shaders_glsl/scalefx/shaders/scalefx-pass{0,1,2,3,4}.glsl (non-GL_ES branch: textureOffset).
 * Pinned against Mesa llvmpipe by tests/golden/scalefx_*.npz (8-bit) and f32_scalefx_*.npz (float targets).
 *
 * Facts measured on the GL for these files: textureOffset on a NEAREST sampler = texel (floor(coord * size) + offset),
 * then the wrap; dist()'s dot(c*d, d) is evaluated cd.x*d.x + (cd.y*d.y + cd.z*d.z) with separate multiplies and adds;
 * step / comparison chains produce exact 0 / 1 floats, so the remaining arithmetic is exact whatever its order. */
#include <math.h>

#include "rc_oracle.h"

#define ENTER unsigned csr_ = o_fp_enter()
#define LEAVE o_fp_leave(csr_)

static inline float minf_(float a, float b) { return a < b ? a : b; }   /* SSE min/max operand order: NaN -> b */
static inline float maxf_(float a, float b) { return a > b ? a : b; }
static inline float modf_(float x, float y) { return x - y * floorf(x / y); }

/* textureOffset(tex, (u, v), ivec2(dx, dy)) */
static o_vec4 tex_off(const o_tex* t, float u, float v, int dx, int dy) {
  if (t->linear) return o_sample(t, u + (float)dx * (1.0f / (float)t->w), v + (float)dy * (1.0f / (float)t->h)); /* not pinned */
  float s = u, q = v;
  if (t->wrap == O_WRAP_REPEAT) {
    s = s - floorf(s);
    q = q - floorf(q);
  }
  int x = (int)floorf(s * (float)t->w) + dx, y = (int)floorf(q * (float)t->h) + dy;
  if (t->wrap == O_WRAP_BORDER) {
    if (x < 0 || y < 0 || x >= t->w || y >= t->h) { o_vec4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
  } else if (t->wrap == O_WRAP_REPEAT) {
    x = ((x % t->w) + t->w) % t->w;
    y = ((y % t->h) + t->h) % t->h;
  } else if (t->wrap == O_WRAP_MIRROR) {
    int px = ((x % (2 * t->w)) + 2 * t->w) % (2 * t->w), py = ((y % (2 * t->h)) + 2 * t->h) % (2 * t->h);
    x = px < t->w ? px : 2 * t->w - 1 - px;
    y = py < t->h ? py : 2 * t->h - 1 - py;
  } else {
    x = x < 0 ? 0 : (x > t->w - 1 ? t->w - 1 : x);
    y = y < 0 ? 0 : (y > t->h - 1 ? t->h - 1 : y);
  }
  return o_texel(t, x, y);
}

typedef struct { o_varying u, v; } uvp;
static uvp texcoord(const o_pass_args* a) {
  uvp p = {o_varying_setup(0.f, 1.f, 1.f, 0.f, a->out_w, a->out_h, a->out_fmt), o_varying_setup(0.f, 0.f, 1.f, 1.f, a->out_w, a->out_h, a->out_fmt)};
  return p;
}

/* ---- pass 0 (FS 118-168): colour metric of E against A, B, C, F -------------------------------------------- */
static float sfx_dist(o_vec4 A, o_vec4 B) {
  const float r = 0.5f * (A.x + B.x);
  const float dx = A.x - B.x, dy = A.y - B.y, dz = A.z - B.z;
  const float cx = 2.0f + r, cy = 4.0f, cz = 3.0f - r;
  return sqrtf((cx * dx) * dx + ((cy * dy) * dy + (cz * dz) * dz)) / 3.0f;
}
void o_pass_scalefx0(const o_pass_args* a) {
  ENTER;
  const uvp tc = texcoord(a);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < a->out_w; ++x) {
      const int lo = o_lower_tri(x, y, a->out_w, a->out_h);
      const float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      const o_vec4 A = tex_off(a->in, u, v, -1, -1), B = tex_off(a->in, u, v, 0, -1), C = tex_off(a->in, u, v, 1, -1);
      const o_vec4 E = tex_off(a->in, u, v, 0, 0), F = tex_off(a->in, u, v, 1, 0);
      const o_vec4 o = {sfx_dist(E, A), sfx_dist(E, B), sfx_dist(E, C), sfx_dist(E, F)};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* ---- pass 1 (FS 120-187): corner strength; params SFX_CLR, SFX_SAA ----------------------------------------- */
static float sfx_str(float d, float ax, float ay, float bx, float by, float clr, float saa) {
  const float diff = ax - ay;
  const float wght1 = maxf_(clr - d, 0.0f) / clr;
  const float t = (1.0f - d) + ((minf_(ax, bx) + ax > minf_(ay, by) + ay) ? diff : -diff);
  const float wght2 = minf_(maxf_(t, 0.0f), 1.0f);
  return (saa == 1.0f || 2.0f * d < ax + ay) ? (wght1 * wght2) * (ax * ay) : 0.0f;
}
void o_pass_scalefx1(const o_pass_args* a) {
  ENTER;
  const uvp tc = texcoord(a);
  const float clr = a->params[0], saa = a->params[1];
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < a->out_w; ++x) {
      const int lo = o_lower_tri(x, y, a->out_w, a->out_h);
      const float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      const o_vec4 A = tex_off(a->in, u, v, -1, -1), B = tex_off(a->in, u, v, 0, -1);
      const o_vec4 D = tex_off(a->in, u, v, -1, 0), E = tex_off(a->in, u, v, 0, 0), F = tex_off(a->in, u, v, 1, 0);
      const o_vec4 G = tex_off(a->in, u, v, -1, 1), H = tex_off(a->in, u, v, 0, 1), I = tex_off(a->in, u, v, 1, 1);
      o_vec4 o;
      o.x = sfx_str(D.z, D.w, E.y, A.w, D.y, clr, saa);
      o.y = sfx_str(F.x, E.w, E.y, B.w, F.y, clr, saa);
      o.z = sfx_str(H.z, E.w, H.y, H.w, I.y, clr, saa);
      o.w = sfx_str(H.x, D.w, H.y, G.w, G.y, clr, saa);
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* ---- pass 2 (FS 118-233): junction resolution; extra[0] = PassPrev2Texture (pass 0's metric) ----------------- */
static inline float LE(float x, float y) { return x < y ? 1.0f : 0.0f; }    /* 1 - step(y, x) */
static inline float GE(float x, float y) { return y < x ? 1.0f : 0.0f; }    /* 1 - step(x, y) */
static inline float LEQ(float x, float y) { return y < x ? 0.0f : 1.0f; }   /* step(x, y) */
static inline float NOT(float x) { return 1.0f - x; }
typedef struct { float v[4]; } f4;
static f4 sfx_dom(const float* x, const float* y, const float* z, const float* w) { /* each: 3 floats */
  f4 r = {{2.0f * x[1] - (x[0] + x[2]), 2.0f * y[1] - (y[0] + y[2]), 2.0f * z[1] - (z[0] + z[2]), 2.0f * w[1] - (w[0] + w[2])}};
  return r;
}
static f4 sfx_vote(f4 jD) { /* majority vote for ambiguous dominance junctions */
  f4 r;
  for (int i = 0; i < 4; ++i) {
    const float a = jD.v[i], b = jD.v[(i + 1) & 3], c = jD.v[(i + 2) & 3], d = jD.v[(i + 3) & 3]; /* .xyzw, .yzwx, .zwxy, .wxyz */
    r.v[i] = minf_(GE(a, 0.0f) * (LEQ(b, 0.0f) * LEQ(d, 0.0f) + GE(a + c, b + d)), 1.0f);
  }
  return r;
}
static float sfx_clear(float cx, float cy, float ax, float ay, float bx, float by) {
  return (cx >= maxf_(minf_(ax, ay), minf_(bx, by))) && (cy >= maxf_(minf_(ax, by), minf_(bx, ay))) ? 1.0f : 0.0f;
}
#define V4(p) {(p).x, (p).y, (p).z, (p).w}
void o_pass_scalefx2(const o_pass_args* a) {
  ENTER;
  const uvp tc = texcoord(a);
  const o_tex* M = a->extra[0];
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < a->out_w; ++x) {
      const int lo = o_lower_tri(x, y, a->out_w, a->out_h);
      const float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      const o_vec4 A = tex_off(M, u, v, -1, -1), B = tex_off(M, u, v, 0, -1);
      const o_vec4 D = tex_off(M, u, v, -1, 0), E = tex_off(M, u, v, 0, 0), F = tex_off(M, u, v, 1, 0);
      const o_vec4 G = tex_off(M, u, v, -1, 1), H = tex_off(M, u, v, 0, 1), I = tex_off(M, u, v, 1, 1);
      const o_vec4 As_ = tex_off(a->in, u, v, -1, -1), Bs_ = tex_off(a->in, u, v, 0, -1), Cs_ = tex_off(a->in, u, v, 1, -1);
      const o_vec4 Ds_ = tex_off(a->in, u, v, -1, 0), Es_ = tex_off(a->in, u, v, 0, 0), Fs_ = tex_off(a->in, u, v, 1, 0);
      const o_vec4 Gs_ = tex_off(a->in, u, v, -1, 1), Hs_ = tex_off(a->in, u, v, 0, 1), Is_ = tex_off(a->in, u, v, 1, 1);
      const float As[4] = V4(As_), Bs[4] = V4(Bs_), Cs[4] = V4(Cs_), Ds[4] = V4(Ds_), Es[4] = V4(Es_), Fs[4] = V4(Fs_);
      const float Gs[4] = V4(Gs_), Hs[4] = V4(Hs_), Is[4] = V4(Is_);
      /* swizzles as index triples: yzw = 1,2,3; zwx = 2,3,0; wxy = 3,0,1; xyz = 0,1,2 */
#define SW3(p, i, j, k) {(p)[i], (p)[j], (p)[k]}
      const float jSx[4] = {As[2], Bs[3], Es[0], Ds[1]}, jSy[4] = {Bs[2], Cs[3], Fs[0], Es[1]};
      const float jSz[4] = {Es[2], Fs[3], Is[0], Hs[1]}, jSw[4] = {Ds[2], Es[3], Hs[0], Gs[1]};
      const float ax_[3] = SW3(As, 1, 2, 3), bx_[3] = SW3(Bs, 2, 3, 0), ex_[3] = SW3(Es, 3, 0, 1), dx_[3] = SW3(Ds, 0, 1, 2);
      const float by_[3] = SW3(Bs, 1, 2, 3), cy_[3] = SW3(Cs, 2, 3, 0), fy_[3] = SW3(Fs, 3, 0, 1), ey_[3] = SW3(Es, 0, 1, 2);
      const float ez_[3] = SW3(Es, 1, 2, 3), fz_[3] = SW3(Fs, 2, 3, 0), iz_[3] = SW3(Is, 3, 0, 1), hz_[3] = SW3(Hs, 0, 1, 2);
      const float dw_[3] = SW3(Ds, 1, 2, 3), ew_[3] = SW3(Es, 2, 3, 0), hw_[3] = SW3(Hs, 3, 0, 1), gw_[3] = SW3(Gs, 0, 1, 2);
      const f4 jx = sfx_vote(sfx_dom(ax_, bx_, ex_, dx_)), jy = sfx_vote(sfx_dom(by_, cy_, fy_, ey_));
      const f4 jz = sfx_vote(sfx_dom(ez_, fz_, iz_, hz_)), jw = sfx_vote(sfx_dom(dw_, ew_, hw_, gw_));
      float res[4];
      res[0] = minf_(jx.v[2] + NOT(jx.v[1]) * NOT(jx.v[3]) * GE(jSx[2], 0.0f) * (jx.v[0] + GE(jSx[0] + jSx[2], jSx[1] + jSx[3])), 1.0f);
      res[1] = minf_(jy.v[3] + NOT(jy.v[2]) * NOT(jy.v[0]) * GE(jSy[3], 0.0f) * (jy.v[1] + GE(jSy[1] + jSy[3], jSy[0] + jSy[2])), 1.0f);
      res[2] = minf_(jz.v[0] + NOT(jz.v[3]) * NOT(jz.v[1]) * GE(jSz[0], 0.0f) * (jz.v[2] + GE(jSz[0] + jSz[2], jSz[1] + jSz[3])), 1.0f);
      res[3] = minf_(jw.v[1] + NOT(jw.v[0]) * NOT(jw.v[2]) * GE(jSw[1], 0.0f) * (jw.v[3] + GE(jSw[1] + jSw[3], jSw[0] + jSw[2])), 1.0f);
      /* single pixel & end of line detection: res * (vec4(jx.z, jy.w, jz.x, jw.y) + NOT(res.wxyz * res.yzwx)) */
      const float j4[4] = {jx.v[2], jy.v[3], jz.v[0], jw.v[1]};
      float res2[4];
      for (int i = 0; i < 4; ++i) res2[i] = minf_(res[i] * (j4[i] + NOT(res[(i + 3) & 3] * res[(i + 1) & 3])), 1.0f);
      const float clr[4] = {sfx_clear(D.z, E.x, D.w, E.y, A.w, D.y), sfx_clear(F.x, E.z, E.w, E.y, B.w, F.y),
                            sfx_clear(H.z, I.x, E.w, H.y, H.w, I.y), sfx_clear(H.x, G.z, D.w, H.y, G.w, G.y)};
      const float h[4] = {minf_(D.w, A.w), minf_(E.w, B.w), minf_(E.w, H.w), minf_(D.w, G.w)};
      const float vv[4] = {minf_(E.y, D.y), minf_(E.y, F.y), minf_(H.y, I.y), minf_(H.y, G.y)};
      const float hadd[4] = {D.w, E.w, E.w, D.w}, vadd[4] = {E.y, E.y, H.y, H.y};
      float out[4];
      for (int i = 0; i < 4; ++i) {
        const float orr = GE(h[i] + hadd[i], vv[i] + vadd[i]);
        const float hori = LE(h[i], vv[i]) * clr[i], vert = GE(h[i], vv[i]) * clr[i];
        out[i] = (res2[i] + 2.0f * hori + 4.0f * vert + 8.0f * orr) / 15.0f;
      }
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* ---- pass 3 (FS 119-256): edge level determination; param SFX_SCN ------------------------------------------- */
typedef struct { int v[4]; } b4;
static b4 sfx_bits(o_vec4 x, float mul, float add) { /* bvec4(floor(mod(x * mul + add, 2.))) */
  const float t[4] = {x.x, x.y, x.z, x.w};
  b4 r;
  for (int i = 0; i < 4; ++i) r.v[i] = floorf(modf_(t[i] * mul + add, 2.0f)) != 0.0f;
  return r;
}
#define CORN(t) sfx_bits(t, 15.0f, 0.5f)
#define HORI(t) sfx_bits(t, 7.5f, 0.25f)
#define VERT(t) sfx_bits(t, 3.75f, 0.125f)
#define ORIE(t) sfx_bits(t, 1.875f, 0.0625f)
enum { X = 0, Y = 1, Z = 2, Wc = 3 };
void o_pass_scalefx3(const o_pass_args* a) {
  ENTER;
  const uvp tc = texcoord(a);
  const int scn = a->params[0] == 1.0f;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < a->out_w; ++x) {
      const int lo = o_lower_tri(x, y, a->out_w, a->out_h);
      const float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      const o_vec4 E = tex_off(a->in, u, v, 0, 0);
      const o_vec4 D = tex_off(a->in, u, v, -1, 0), D0 = tex_off(a->in, u, v, -2, 0), D1 = tex_off(a->in, u, v, -3, 0);
      const o_vec4 F = tex_off(a->in, u, v, 1, 0), F0 = tex_off(a->in, u, v, 2, 0), F1 = tex_off(a->in, u, v, 3, 0);
      const o_vec4 B = tex_off(a->in, u, v, 0, -1), B0 = tex_off(a->in, u, v, 0, -2), B1 = tex_off(a->in, u, v, 0, -3);
      const o_vec4 H = tex_off(a->in, u, v, 0, 1), H0 = tex_off(a->in, u, v, 0, 2), H1 = tex_off(a->in, u, v, 0, 3);
      const b4 Ec = CORN(E), Eh = HORI(E), Ev = VERT(E), Eo = ORIE(E);
      const b4 Dc = CORN(D), Dh = HORI(D), Do = ORIE(D), D0c = CORN(D0), D0h = HORI(D0), D1h = HORI(D1);
      const b4 Fc = CORN(F), Fh = HORI(F), Fo = ORIE(F), F0c = CORN(F0), F0h = HORI(F0), F1h = HORI(F1);
      const b4 Bc = CORN(B), Bv = VERT(B), Bo = ORIE(B), B0c = CORN(B0), B0v = VERT(B0), B1v = VERT(B1);
      const b4 Hc = CORN(H), Hv = VERT(H), Ho = ORIE(H), H0c = CORN(H0), H0v = VERT(H0), H1v = VERT(H1);
#define c(b, i) ((b).v[i])
      const int lvl1x = c(Ec, X) && (c(Dc, Z) || c(Bc, Z) || scn), lvl1y = c(Ec, Y) && (c(Fc, Wc) || c(Bc, Wc) || scn);
      const int lvl1z = c(Ec, Z) && (c(Fc, X) || c(Hc, X) || scn), lvl1w = c(Ec, Wc) && (c(Dc, Y) || c(Hc, Y) || scn);
      const int l2x0 = (c(Ec, X) && c(Eh, Y)) && c(Dc, Z), l2x1 = (c(Ec, Y) && c(Eh, X)) && c(Fc, Wc);
      const int l2y0 = (c(Ec, Y) && c(Ev, Z)) && c(Bc, Wc), l2y1 = (c(Ec, Z) && c(Ev, Y)) && c(Hc, X);
      const int l2z0 = (c(Ec, Wc) && c(Eh, Z)) && c(Dc, Y), l2z1 = (c(Ec, Z) && c(Eh, Wc)) && c(Fc, X);
      const int l2w0 = (c(Ec, X) && c(Ev, Wc)) && c(Bc, Z), l2w1 = (c(Ec, Wc) && c(Ev, X)) && c(Hc, Y);
      const int l3x0 = l2x1 && (c(Dh, Y) && c(Dh, X)) && c(Fh, Z), l3x1 = l2w1 && (c(Bv, Wc) && c(Bv, X)) && c(Hv, Z);
      const int l3y0 = l2x0 && (c(Fh, X) && c(Fh, Y)) && c(Dh, Wc), l3y1 = l2y1 && (c(Bv, Z) && c(Bv, Y)) && c(Hv, Wc);
      const int l3z0 = l2z0 && (c(Fh, Wc) && c(Fh, Z)) && c(Dh, X), l3z1 = l2y0 && (c(Hv, Y) && c(Hv, Z)) && c(Bv, X);
      const int l3w0 = l2z1 && (c(Dh, Z) && c(Dh, Wc)) && c(Fh, Y), l3w1 = l2w0 && (c(Hv, X) && c(Hv, Wc)) && c(Bv, Y);
      const int l4x0 = (c(Dc, X) && c(Dh, Y) && c(Eh, X) && c(Eh, Y) && c(Fh, X) && c(Fh, Y)) && (c(D0c, Z) && c(D0h, Wc));
      const int l4x1 = (c(Bc, X) && c(Bv, Wc) && c(Ev, X) && c(Ev, Wc) && c(Hv, X) && c(Hv, Wc)) && (c(B0c, Z) && c(B0v, Y));
      const int l4y0 = (c(Fc, Y) && c(Fh, X) && c(Eh, Y) && c(Eh, X) && c(Dh, Y) && c(Dh, X)) && (c(F0c, Wc) && c(F0h, Z));
      const int l4y1 = (c(Bc, Y) && c(Bv, Z) && c(Ev, Y) && c(Ev, Z) && c(Hv, Y) && c(Hv, Z)) && (c(B0c, Wc) && c(B0v, X));
      const int l4z0 = (c(Fc, Z) && c(Fh, Wc) && c(Eh, Z) && c(Eh, Wc) && c(Dh, Z) && c(Dh, Wc)) && (c(F0c, X) && c(F0h, Y));
      const int l4z1 = (c(Hc, Z) && c(Hv, Y) && c(Ev, Z) && c(Ev, Y) && c(Bv, Z) && c(Bv, Y)) && (c(H0c, X) && c(H0v, Wc));
      const int l4w0 = (c(Dc, Wc) && c(Dh, Z) && c(Eh, Wc) && c(Eh, Z) && c(Fh, Wc) && c(Fh, Z)) && (c(D0c, Y) && c(D0h, X));
      const int l4w1 = (c(Hc, Wc) && c(Hv, X) && c(Ev, Wc) && c(Ev, X) && c(Bv, Wc) && c(Bv, X)) && (c(H0c, Y) && c(H0v, Z));
      const int l5x0 = l4x0 && (c(F0h, X) && c(F0h, Y)) && (c(D1h, Z) && c(D1h, Wc)), l5x1 = l4y0 && (c(D0h, Y) && c(D0h, X)) && (c(F1h, Wc) && c(F1h, Z));
      const int l5y0 = l4y1 && (c(H0v, Y) && c(H0v, Z)) && (c(B1v, Wc) && c(B1v, X)), l5y1 = l4z1 && (c(B0v, Z) && c(B0v, Y)) && (c(H1v, X) && c(H1v, Wc));
      const int l5z0 = l4w0 && (c(F0h, Wc) && c(F0h, Z)) && (c(D1h, Y) && c(D1h, X)), l5z1 = l4z0 && (c(D0h, Z) && c(D0h, Wc)) && (c(F1h, X) && c(F1h, Y));
      const int l5w0 = l4x1 && (c(H0v, X) && c(H0v, Wc)) && (c(B1v, Z) && c(B1v, Y)), l5w1 = l4w1 && (c(B0v, Wc) && c(B0v, X)) && (c(H1v, Y) && c(H1v, Z));
      const int l6x0 = l5x1 && (c(D1h, Y) && c(D1h, X)), l6x1 = l5w1 && (c(B1v, Wc) && c(B1v, X));
      const int l6y0 = l5x0 && (c(F1h, X) && c(F1h, Y)), l6y1 = l5y1 && (c(B1v, Z) && c(B1v, Y));
      const int l6z0 = l5z0 && (c(F1h, Wc) && c(F1h, Z)), l6z1 = l5y0 && (c(H1v, Y) && c(H1v, Z));
      const int l6w0 = l5z1 && (c(D1h, Z) && c(D1h, Wc)), l6w1 = l5w0 && (c(H1v, X) && c(H1v, Wc));
      float crn[4], mid[4];
      crn[0] = ((lvl1x && c(Eo, X)) || (l3x0 && c(Eo, Y)) || (l4x0 && c(Do, X)) || (l6x0 && c(Fo, Y))) ? 5.f : (lvl1x || (l3x1 && !c(Eo, Wc)) || (l4x1 && !c(Bo, X)) || (l6x1 && !c(Ho, Wc))) ? 1.f : l3x0 ? 3.f : l3x1 ? 7.f : l4x0 ? 2.f : l4x1 ? 6.f : l6x0 ? 4.f : l6x1 ? 8.f : 0.f;
      crn[1] = ((lvl1y && c(Eo, Y)) || (l3y0 && c(Eo, X)) || (l4y0 && c(Fo, Y)) || (l6y0 && c(Do, X))) ? 5.f : (lvl1y || (l3y1 && !c(Eo, Z)) || (l4y1 && !c(Bo, Y)) || (l6y1 && !c(Ho, Z))) ? 3.f : l3y0 ? 1.f : l3y1 ? 7.f : l4y0 ? 4.f : l4y1 ? 6.f : l6y0 ? 2.f : l6y1 ? 8.f : 0.f;
      crn[2] = ((lvl1z && c(Eo, Z)) || (l3z0 && c(Eo, Wc)) || (l4z0 && c(Fo, Z)) || (l6z0 && c(Do, Wc))) ? 7.f : (lvl1z || (l3z1 && !c(Eo, Y)) || (l4z1 && !c(Ho, Z)) || (l6z1 && !c(Bo, Y))) ? 3.f : l3z0 ? 1.f : l3z1 ? 5.f : l4z0 ? 4.f : l4z1 ? 8.f : l6z0 ? 2.f : l6z1 ? 6.f : 0.f;
      crn[3] = ((lvl1w && c(Eo, Wc)) || (l3w0 && c(Eo, Z)) || (l4w0 && c(Do, Wc)) || (l6w0 && c(Fo, Z))) ? 7.f : (lvl1w || (l3w1 && !c(Eo, X)) || (l4w1 && !c(Ho, Wc)) || (l6w1 && !c(Bo, X))) ? 1.f : l3w0 ? 3.f : l3w1 ? 5.f : l4w0 ? 2.f : l4w1 ? 8.f : l6w0 ? 4.f : l6w1 ? 6.f : 0.f;
      mid[0] = ((l2x0 && c(Eo, X)) || (l2x1 && c(Eo, Y)) || (l5x0 && c(Do, X)) || (l5x1 && c(Fo, Y))) ? 5.f : l2x0 ? 1.f : l2x1 ? 3.f : l5x0 ? 2.f : l5x1 ? 4.f : (c(Ec, X) && c(Dc, Z) && c(Ec, Y) && c(Fc, Wc)) ? (c(Eo, X) ? (c(Eo, Y) ? 5.f : 3.f) : 1.f) : 0.f;
      mid[1] = ((l2y0 && !c(Eo, Y)) || (l2y1 && !c(Eo, Z)) || (l5y0 && !c(Bo, Y)) || (l5y1 && !c(Ho, Z))) ? 3.f : l2y0 ? 5.f : l2y1 ? 7.f : l5y0 ? 6.f : l5y1 ? 8.f : (c(Ec, Y) && c(Bc, Wc) && c(Ec, Z) && c(Hc, X)) ? (!c(Eo, Y) ? (!c(Eo, Z) ? 3.f : 7.f) : 5.f) : 0.f;
      mid[2] = ((l2z0 && c(Eo, Wc)) || (l2z1 && c(Eo, Z)) || (l5z0 && c(Do, Wc)) || (l5z1 && c(Fo, Z))) ? 7.f : l2z0 ? 1.f : l2z1 ? 3.f : l5z0 ? 2.f : l5z1 ? 4.f : (c(Ec, Z) && c(Fc, X) && c(Ec, Wc) && c(Dc, Y)) ? (c(Eo, Z) ? (c(Eo, Wc) ? 7.f : 1.f) : 3.f) : 0.f;
      mid[3] = ((l2w0 && !c(Eo, X)) || (l2w1 && !c(Eo, Wc)) || (l5w0 && !c(Bo, X)) || (l5w1 && !c(Ho, Wc))) ? 1.f : l2w0 ? 5.f : l2w1 ? 7.f : l5w0 ? 6.f : l5w1 ? 8.f : (c(Ec, Wc) && c(Hc, Y) && c(Ec, X) && c(Bc, Z)) ? (!c(Eo, Wc) ? (!c(Eo, X) ? 1.f : 5.f) : 7.f) : 0.f;
#undef c
      const o_vec4 o = {(crn[0] + 9.0f * mid[0]) / 80.0f, (crn[1] + 9.0f * mid[1]) / 80.0f, (crn[2] + 9.0f * mid[2]) / 80.0f, (crn[3] + 9.0f * mid[3]) / 80.0f};
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}

/* ---- pass 4 (FS 116-177): subpixel lookup; extra[0] = PassPrev5Texture (the original frame) ------------------ */
void o_pass_scalefx4(const o_pass_args* a) {
  ENTER;
  const uvp tc = texcoord(a);
  const float ssx = (float)a->in->w, ssy = (float)a->in->h;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < a->out_w; ++x) {
      const int lo = o_lower_tri(x, y, a->out_w, a->out_h);
      const float u = o_varying_at(&tc.u, x, y, lo), v = o_varying_at(&tc.v, x, y, lo);
      const o_vec4 E = o_sample(a->in, u, v);
      const float e[4] = {E.x, E.y, E.z, E.w};
      float crn[4], mid[4];
      for (int i = 0; i < 4; ++i) {
        crn[i] = floorf(modf_(e[i] * 80.0f + 0.5f, 9.0f));
        mid[i] = floorf(modf_(e[i] * 8.888888f + 0.055555f, 9.0f));
      }
      const float px = u * ssx, py = v * ssy;
      const float fx = floorf(3.0f * (px - floorf(px))), fy = floorf(3.0f * (py - floorf(py)));
      const float sp = fy == 0.f ? (fx == 0.f ? crn[0] : fx == 1.f ? mid[0] : crn[1])
                                 : (fy == 1.f ? (fx == 0.f ? mid[3] : fx == 1.f ? 0.f : mid[1]) : (fx == 0.f ? crn[3] : fx == 1.f ? mid[2] : crn[2]));
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
      const o_vec4 o = o_sample(a->extra[0], u + rx / ssx, v + ry / ssy);
      o_store_pixel(a, x, y, o);
    }
  LEAVE;
}
