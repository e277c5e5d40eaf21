/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * handheld/shaders/lcd-cgwg/lcd-grid-v2.glsl (handheld/lcd-grid-v2.glslp and the lcd-grid-v2-<colour>[-motionblur] chains): every
 * output pixel integrates the subpixel profile of the four source texels around it (intsmear, FS 150-166: a 13th-degree odd
 * polynomial per edge, six times horizontally and twice vertically), on texels fetched with texelFetchOffset and passed
 * through pow(gain * t + blacklevel, gamma) + ambient, then a 3x3 subpixel colour matrix and the output gamma.
 * ~580 scalar operations that the GL's compiler reassociates (0.5 * dx folded into the constants, the last polynomial term as
 * (zn * c) * z2, shared clamps): the body is the GL's own instruction list (oracle/glrun/nir2c.py, recipe gen_lists.sh),
 * pinned by tests/golden/lcd_grid_v2_* (llvmpipe, 8-bit and float).  The vertex stage is TEX0 = TexCoord.
 * params: the shader's 15 #pragma parameters in declaration order. */
#include <math.h>
#include <string.h>

#include "rc_oracle.h"

#define RCN_FN static
static inline float RCN_BITS(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define RCN_RCP(x) (1.0f / (x))
#define RCN_DIV(a, b) ((a) / (b))
#define RCN_FLOOR(x) floorf(x)
static inline int rcn_f2i(float x) { return x != x || x >= 2147483648.0f || x < -2147483648.0f ? (-2147483647 - 1) : (int)x; }   /* cvttps2dq */
#define RCN_F2I(x) rcn_f2i(x)
static inline float rcn_min(float a, float b) { return b != b ? a : (a < b ? a : b); }
static inline float rcn_max(float a, float b) { return b != b ? a : (a > b ? a : b); }
#define RCN_MIN(a, b) rcn_min(a, b)
#define RCN_MAX(a, b) rcn_max(a, b)
#define RCN_POW(a, b) o_pow(a, b)
/* texelFetch: the decoded texel, zeros outside the image */
static void rcn_txf(void* ctx, int x, int y, float* dst) {
  const o_tex* t = (const o_tex*)ctx;
  o_vec4 r = {0.f, 0.f, 0.f, 0.f};
  if (x >= 0 && y >= 0 && x < t->w && y < t->h) r = o_texel(t, x, y);
  dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; dst[3] = r.w;
}
#define RCN_TXF(ctx, unit, x, y, dst) rcn_txf(ctx, x, y, dst)
static void rcn_tex(void* ctx, float u, float v, float* dst) {
  const o_vec4 r = o_sample((const o_tex*)ctx, u, v);
  dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; dst[3] = r.w;
}
#define RCN_TEX(ctx, unit, u, v, dst) rcn_tex(ctx, u, v, dst)

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wunused-but-set-variable"
#include "gen/lcd_grid_v2_fs.inc"
#include "gen/lcd_grid_fs.inc"
#pragma GCC diagnostic pop

void o_pass_lcd_grid_v2(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float U[24] = {0};
  U[0] = (float)W; U[1] = (float)H;
  U[2] = (float)a->in->w; U[3] = (float)a->in->h; U[4] = U[2]; U[5] = U[3];
  for (int k = 0; k < 15; ++k) U[6 + k] = a->params[k];
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)};
      float out[4];
      lcd_grid_v2_fs(U, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}

/* handheld/shaders/lcd-cgwg/lcd-grid.glsl (handheld/lcd-grid.glslp, nds.glslp and twenty console-border presets): the first version
 * of the same idea - four sampled texels (texture(), not texelFetch) under the subpixel integrals, one GRID_STRENGTH and an input
 * gamma; ~560 operations, the GL's own instruction list again (gen/lcd_grid_fs.inc), pinned by tests/golden/lcd_grid_*.
 * params: GRID_STRENGTH, gamma. */
void o_pass_lcd_grid(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float U[8] = {(float)W, (float)H, (float)a->in->w, (float)a->in->h, (float)a->in->w, (float)a->in->h, a->params[0], a->params[1]};
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)};
      float out[4];
      lcd_grid_fs(U, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}
