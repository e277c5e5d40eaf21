/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * crt/crt-geom.glslp: one pass, reference shaders/shaders_glsl/crt/shaders/crt-geom.glsl
 * (VS main 155-207, helpers 117-153; FS helpers 293-374, main 377-504).
 * 17 params in pragma order: CRTgamma, monitorgamma, d, CURVATURE, R, cornersize, cornersmooth, x_tilt, y_tilt,
 * overscan_x, overscan_y, DOTMASK, SHARPER, scanline_weight, lum, interlace_detect, SATURATION.
 *
 * The vertex shader computes sinangle / cosangle / stretch (maxscale(): bkwtrans + four fwtrans) from
 * uniforms only, so all four vertices carry the same value and the plane equations hand every pixel
 * that value unchanged; TEX0 (= TexCoord * 1.0001) and mod_factor are real planes.
 * acos is Mesa's polynomial (nir_builtin_builder build_asin with the acos coefficients):
 *   asin(x) = sign(x) * (pi/2 - sqrt(1 - |x|) * (pi/2 + |x| * (pi/4 - 1 + |x| * (p0 + |x| * p1))))
 *   acos(x) = pi/2 - asin(x), p0 = 0.08132463, p1 = -0.02363318, no fused operations
 * (oracle/probes: 65536 random arguments, bit-identical).
 */
#include <math.h>

#include "rc_oracle.h"

static inline float minps(float a, float b) { return a < b ? a : b; } /* SSE minps: NaN -> b */
static inline float maxps(float a, float b) { return a > b ? a : b; }
static inline float clampf(float x, float lo, float hi) { return minps(maxps(x, lo), hi); }
static inline float fix(float c) { return maxps(fabsf(c), 1e-5f); }
static inline float modf_glsl(float x, float y) { return x - y * floorf(x / y); }

float o_acos(float x) {
  const float half_pi = 1.57079637f, p0 = 0.08132463f, p1 = -0.02363318f;
  const float ax = fabsf(x);
  float e = ax * p1 + p0;
  e = ax * e + (0.785398185f - 1.0f);
  e = ax * e + half_pi;
  const float sg = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
  const float as = sg * (half_pi - sqrtf(1.0f - ax) * e);
  return half_pi - as;
}

typedef struct { float R, d, sx, sy, cx, cy; } geom;
typedef struct { float x, y; } v2;

static float g_intersect(const geom* g, v2 p) {
  const float R = g->R, d = g->d;
  const float A = (p.x * p.x + p.y * p.y) + d * d;
  const float B = 2.0f * (R * ((p.x * g->sx + p.y * g->sy) - (d * g->cx) * g->cy) - d * d);
  const float C = d * d + 2.0f * (((d * g->cx) * g->cy) * R);   /* the product shares d*cos*cos with B */
  return (-B - sqrtf(B * B - (4.0f * A) * C)) / (2.0f * A);
}
static v2 g_bkwtrans(const geom* g, v2 p, int fixed) {
  const float R = g->R;
  const float c = g_intersect(g, p);
  v2 pt = {c * p.x, c * p.y};
  pt.x = pt.x - (-R) * g->sx; pt.y = pt.y - (-R) * g->sy;
  pt.x = pt.x / R; pt.y = pt.y / R;
  const v2 tang = {g->sx / g->cx, g->sy / g->cy};
  const v2 poc = {pt.x / g->cx, pt.y / g->cy};
  const float A = tang.x * tang.x + (tang.y * tang.y + 1.0f);   /* dot + constant: the constant joins the inner addend */
  const float B = -2.0f * (poc.x * tang.x + poc.y * tang.y);
  const float C = poc.x * poc.x + (poc.y * poc.y - 1.0f);
  const float a = (-B + sqrtf(B * B - (4.0f * A) * C)) / (2.0f * A);
  const v2 uv = {(pt.x - a * g->sx) / g->cx, (pt.y - a * g->sy) / g->cy};
  float r = R * o_acos(a);
  if (fixed) r = fix(r);
  const float s = o_sin(r / R);
  const v2 o = {(uv.x * r) / s, (uv.y * r) / s};
  return o;
}
static v2 g_fwtrans(const geom* g, v2 uv) {
  const float R = g->R, d = g->d;
  const float r = fix(sqrtf(uv.x * uv.x + uv.y * uv.y));
  const float k = o_sin(r / R) / r;
  uv.x *= k; uv.y *= k;
  const float x = 1.0f - o_cos(r / R);
  const float D = uv.x * g->sx + (uv.y * g->sy + (d / R + (x * g->cx) * g->cy));   /* addend + dot: joins the dot's inner term */
  const v2 o = {(d * (uv.x * g->cx - x * g->sx)) / D, (d * (uv.y * g->cy - x * g->sy)) / D};
  return o;
}
static void g_maxscale(const geom* g, float ax, float ay, float* stretch) {
  const float den = 1.0f + ((g->R / g->d) * g->cx) * g->cy;
  const v2 c0 = {(-g->R * g->sx) / den, (-g->R * g->sy) / den};
  const v2 c = g_bkwtrans(g, c0, 0);
  const float hx = 0.5f * ax, hy = 0.5f * ay;
  const v2 p0 = {-hx, c.y}, p1 = {c.x, -hy}, p2 = {hx, c.y}, p3 = {c.x, hy};
  const float lox = g_fwtrans(g, p0).x / ax, loy = g_fwtrans(g, p1).y / ay;
  const float hix = g_fwtrans(g, p2).x / ax, hiy = g_fwtrans(g, p3).y / ay;
  stretch[0] = ((hix + lox) * ax) * 0.5f;
  stretch[1] = ((hiy + loy) * ay) * 0.5f;
  stretch[2] = maxps(hix - lox, hiy - loy);
}

/* what the vertex shader hands the fragment shader (identical at all four vertices):
 * out = sinangle.xy, cosangle.xy, stretch.xyz */
static void geom_vertex(const float* P, geom* g, float* stretch) {
  g->R = P[4]; g->d = P[2];
  g->sx = o_sin(P[7]) + 0.001f; g->sy = o_sin(P[8]) + 0.001f;
  g->cx = o_cos(P[7]) + 0.001f; g->cy = o_cos(P[8]) + 0.001f;
  g_maxscale(g, 1.0f, 0.75f, stretch);
}
void o_crt_geom_vertex(const float* params, float* out) {
  unsigned csr = o_fp_enter();
  geom g;
  geom_vertex(params, &g, out + 4);
  out[0] = g.sx; out[1] = g.sy; out[2] = g.cx; out[3] = g.cy;
  o_fp_leave(csr);
}

static void scanline_weights(float distance, const float* col, float sw, float lum, float* out) {
  for (int c = 0; c < 3; ++c) {
    const float c2 = col[c] * col[c];
    const float wid = 2.0f + 2.0f * (c2 * c2);   /* pow(color, 4.0) is lowered to two squarings */
    const float w = distance / sw;
    const float p = o_pow(w * (1.0f / sqrtf(0.5f * wid)), wid);
    out[c] = ((lum + 1.4f) * o_exp(-p)) / (0.6f + 0.2f * wid);
  }
}

static void o_pass_crt_geom_body(const o_pass_args* a) {
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  const float crt_gamma = P[0], mon_gamma = P[1], curvature = P[3], cornersize = P[5], cornersmooth = P[6];
  const float ovx = P[9] / 100.0f, ovy = P[10] / 100.0f, dotmask = P[11], sharper = P[12], sw = P[13], lum = P[14];
  const float interlace = P[15], satur = P[16];
  const float tsx = (float)a->in->w, tsy = (float)a->in->h; /* TextureSize == InputSize */
  const float aspx = 1.0f, aspy = 0.75f;
  geom g;
  float stretch[3];
  geom_vertex(P, &g, stretch);
  const float ilfac_y = clampf(floorf(tsy / 200.0f), 1.0f, 2.0f);
  const float one_x = 1.0f / (sharper * tsx), one_y = ilfac_y / tsy;
  const float k1 = 1.0001f;
  o_varying tu = o_varying_setup(0.f * k1, 1.f * k1, 1.f * k1, 0.f * k1, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * k1, 0.f * k1, 1.f * k1, 1.f * k1, W, H, a->out_fmt);
  const float mf1 = ((1.0f * tsx) * (float)W) / tsx;
  o_varying vm = o_varying_setup(0.f, mf1, mf1, 0.f, W, H, a->out_fmt);
  const float ilvec_y = ilfac_y * interlace > 1.5f ? modf_glsl((float)a->frame_count, 2.0f) : 0.0f;
  const float filter_ = tsy / (float)H;
  const float pi = 3.141592653589f;
  const float inv_mon = 1.0f / mon_gamma;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float u = o_varying_at(&tu, x, y, lo), v = o_varying_at(&tv, x, y, lo);
      float xx = u, xy = v;
      if (curvature > 0.5f) {
        v2 c = {u * (tsx / tsx), v * (tsy / tsy)};
        c.x = ((c.x - 0.5f) * aspx) * stretch[2] + stretch[0];
        c.y = ((c.y - 0.5f) * aspy) * stretch[2] + stretch[1];
        const v2 b = g_bkwtrans(&g, c, 1);
        xx = ((((b.x / ovx) / aspx) + 0.5f) * tsx) / tsx;
        xy = ((((b.y / ovy) / aspy) + 0.5f) * tsy) / tsy;
      }
      /* corner() */
      float cval;
      {
        float cx = xx * (tsx / tsx), cy = xy * (tsy / tsy);
        /* 1 - (t + 0.5) is re-associated to 0.5 - t */
        const float tx = (cx - 0.5f) * ovx, ty = (cy - 0.5f) * ovy;
        cx = minps(tx + 0.5f, 0.5f - tx) * aspx; cy = minps(ty + 0.5f, 0.5f - ty) * aspy;
        cx = cornersize - minps(cx, cornersize); cy = cornersize - minps(cy, cornersize);
        const float dist = sqrtf(cx * cx + cy * cy);
        cval = clampf((cornersize - dist) * cornersmooth, 0.0f, 1.0f) * 1.0001f;
      }
      const float rsx = (xx * tsx - 0.5f) / 1.0f, rsy = (xy * tsy + (ilvec_y - 0.5f)) / ilfac_y;
      const float uvx = rsx - floorf(rsx);
      float uvy = rsy - floorf(rsy);
      const float px = (floorf(rsx) * 1.0f + 0.5f) / tsx, py = (floorf(rsy) * ilfac_y + (0.5f - ilvec_y)) / tsy;
      float k[4] = {pi * (1.0f + uvx), pi * uvx, pi * (1.0f - uvx), pi * (2.0f - uvx)};
      for (int q = 0; q < 4; ++q) {
        const float c = fix(k[q]);
        k[q] = ((2.0f * o_sin(c)) * o_sin(c * 0.5f)) / (c * c);
      }
      const float ksum = k[0] + (k[1] + (k[2] + k[3]));
      for (int q = 0; q < 4; ++q) k[q] = k[q] / ksum;
      float col[3], col2[3];
      for (int row = 0; row < 2; ++row) {
        const float ty = row ? py + one_y : py;
        const float tx[4] = {px + -one_x, px, px + one_x, px + 2.0f * one_x};
        float m[4][3];
        for (int q = 0; q < 4; ++q) {
          const o_vec4 t = o_sample(a->in, tx[q], ty);
          m[q][0] = o_pow(t.x, crt_gamma); m[q][1] = o_pow(t.y, crt_gamma); m[q][2] = o_pow(t.z, crt_gamma);
        }
        float* dst = row ? col2 : col;
        for (int c = 0; c < 3; ++c) dst[c] = clampf(((m[0][c] * k[0] + m[1][c] * k[1]) + m[2][c] * k[2]) + m[3][c] * k[3], 0.0f, 1.0f);
      }
      float w1[3], w2[3], t1[3], t2[3];
      scanline_weights(uvy, col, sw, lum, w1);
      scanline_weights(1.0f - uvy, col2, sw, lum, w2);
      uvy = uvy + 0.333333343f * filter_;
      scanline_weights(uvy, col, sw, lum, t1);
      scanline_weights(fabsf(1.0f - uvy), col2, sw, lum, t2);
      for (int c = 0; c < 3; ++c) { w1[c] = (w1[c] + t1[c]) / 3.0f; w2[c] = (w2[c] + t2[c]) / 3.0f; }
      uvy = uvy - 0.666666687f * filter_;
      scanline_weights(fabsf(uvy), col, sw, lum, t1);
      scanline_weights(fabsf(1.0f - uvy), col2, sw, lum, t2);
      for (int c = 0; c < 3; ++c) { w1[c] = w1[c] + t1[c] / 3.0f; w2[c] = w2[c] + t2[c] / 3.0f; }
      const float mf = o_varying_at(&vm, x, y, lo);
      const float t = floorf(modf_glsl(mf, 2.0f));
      const float ma[3] = {1.0f, 1.0f - dotmask, 1.0f}, mb[3] = {1.0f - dotmask, 1.0f, 1.0f - dotmask};
      float res[3];
      for (int c = 0; c < 3; ++c) {
        float r = (col[c] * w1[c] + col2[c] * w2[c]) * cval;
        r = r * (ma[c] + t * (mb[c] - ma[c]));
        res[c] = o_pow(r, inv_mon);
      }
      const float len = sqrtf(res[0] * res[0] + (res[1] * res[1] + res[2] * res[2])) * 0.5775f;
      const float l3[3] = {len < 0.5f ? 0.3f * 0.3f + 0.3f * 0.3f : 0.3f, len < 0.5f ? 0.6f * 0.6f + 0.6f * 0.6f : 0.6f,
                           len < 0.5f ? 0.1f * 0.1f + 0.1f * 0.1f : 0.1f};
      const float grey = res[0] * l3[0] + (res[1] * l3[1] + res[2] * l3[2]);
      const float gs = grey * (1.0f - satur);
      const o_vec4 o = {gs + res[0] * satur, gs + res[1] * satur, gs + res[2] * satur, 1.0f};
      o_store_pixel(a, x, y, o);
    }
}
void o_pass_crt_geom(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  o_pass_crt_geom_body(a);
  o_fp_leave(csr);
}
