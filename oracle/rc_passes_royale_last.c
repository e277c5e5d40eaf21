/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * crt-royale's last pass with curved geometry (geom_mode_runtime 1..3) or overscan != 1: geometry-aa-last-pass.glsl
 * FS 5451-5531 takes the tex2Daa12x branch (3985-4063) and, when curved, the screen-space ray casts (2527-3010); its VS
 * (5337-5400) builds the eye position and the global-to-local matrix from sin/cos of the tilt angles.
 *
 * That is ~2400 scalar operations whose association the GL's compiler rearranges freely (constant folding of the
 * sample grid, factored weight sums, CSE across the branches), so instead of a hand restatement the body is the GL's own
 * final instruction list: oracle/glrun/nir2c.py turns the NIR llvmpipe compiles for this shader (LP_DEBUG=fs,
 * GALLIVM_DEBUG=tgsi; recipe oracle/glrun/gen_lists.sh) into straight C, one statement per instruction, and the
 * float built-ins map to the llvmpipe-exact primitives of rc_math.c.  The vertex stage runs at the quad's four vertices
 * and every varying goes through the rasteriser's plane setup (rc_varying.c), as everywhere else.
 * Pinned by tests/golden/crt_royale_geom_*.npz and f32_crt_royale_geom_*.npz (llvmpipe outputs, 8-bit and float).
 */
#include <math.h>
#include <string.h>

#include "rc_oracle.h"

#define RCN_FN static
static inline float RCN_BITS(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define RCN_ABS(x) fabsf(x)
#define RCN_RSQ(x) (1.0f / sqrtf(x))
#define RCN_RCP(x) (1.0f / (x))
#define RCN_SQRT(x) sqrtf(x)
static inline float rcn_sign(float x) { return x == 0.0f ? 0.0f : copysignf(1.0f, x); }
#define RCN_SIGN(x) rcn_sign(x)
#define RCN_SIN(x) o_sin(x)
#define RCN_COS(x) o_cos(x)
#define RCN_DIV(a, b) ((a) / (b))
/* fmin / fmax as gallivm builds them (lp_build_min_simple, GALLIVM_NAN_RETURN_OTHER): MINPS / MAXPS - the second operand
 * when either is NaN - and then the first operand where the second is NaN: the operand that is not NaN */
static inline float rcn_min(float a, float b) { return b != b ? a : (a < b ? a : b); }
static inline float rcn_max(float a, float b) { return b != b ? a : (a > b ? a : b); }
#define RCN_MIN(a, b) rcn_min(a, b)
#define RCN_MAX(a, b) rcn_max(a, b)
#define RCN_POW(a, b) o_pow(a, b)
/* texture(): plain, or - mipmap_input, crt-royale-fake-bloom - with the LOD llvmpipe derives per tex instruction from the
 * coordinate differences inside the 2x2 quad: the coordinates of every tap are first recorded at the quad's pixels
 * (TEX_RECORD), then each tap of this pixel is filtered with its own differences (TEX_MIP; rc_sampler.c o_sample_quad) */
enum { TEX_PLAIN = 0, TEX_RECORD = 1, TEX_MIP = 2, MAX_TAPS = 16 };
typedef struct {
  const o_tex* t;
  int mode, n;
  float (*rec)[2];            /* TEX_RECORD: where the coordinates go */
  const float (*q[4])[2];     /* TEX_MIP: recorded coordinates at (x0, y0), (x0 + 1, y0), (x0, y0), (x0, y0 + 1) */
} tex_ctx;
static void rcn_tex(void* vctx, float u, float v, float* dst) {
  tex_ctx* c = (tex_ctx*)vctx;
  o_vec4 r = {0.f, 0.f, 0.f, 0.f};
  if (c->mode == TEX_RECORD) {
    if (c->n < MAX_TAPS) { c->rec[c->n][0] = u; c->rec[c->n][1] = v; }
    c->n++;
  } else if (c->mode == TEX_MIP) {
    const int k = c->n++;
    r = o_sample_quad(c->t, u, v, c->q[0][k][0], c->q[1][k][0], c->q[0][k][1], c->q[1][k][1], c->q[2][k][0], c->q[3][k][0], c->q[2][k][1],
                      c->q[3][k][1]);
  } else {
    r = o_sample(c->t, u, v);
  }
  dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; dst[3] = r.w;
}
#define RCN_TEX(ctx, unit, u, v, dst) rcn_tex(ctx, u, v, dst)

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wunused-but-set-variable"
#include "gen/royale_last_vs.inc"
#include "gen/royale_last_fs.inc"
#pragma GCC diagnostic pop

static void set_uniform(float* U, const char* name, const float* v, int n, const void* table) {
  const struct { const char* name; int off, n, flat; }* t = table;
  for (; t->name; ++t)
    if (!strcmp(t->name, name)) {
      for (int k = 0; k < n && k < t->n; ++k) U[t->off + k] = v[k];
      return;
    }
}

int o_royale_last_is_general(const float* P) { return P[30] > 0.5f || P[37] != 1.0f || P[38] != 1.0f; }

void o_pass_royale_last_general(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  const float* P = a->params;
  const float tex_size[2] = {(float)a->in->w, (float)a->in->h}, out_size[2] = {(float)W, (float)H};
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  static const struct { const char* name; int idx; } pn[] = {
      {"lcd_gamma", 1}, {"aa_cubic_c", 28}, {"geom_mode_runtime", 30}, {"geom_radius", 31}, {"geom_view_dist", 32},
      {"geom_tilt_angle_x", 33}, {"geom_tilt_angle_y", 34}, {"geom_aspect_ratio_x", 35}, {"geom_aspect_ratio_y", 36},
      {"geom_overscan_x", 37}, {"geom_overscan_y", 38}, {"border_size", 39}, {"border_darkness", 40}, {"border_compress", 41}, {0, 0}};
  float Uv[64] = {0}, Uf[64] = {0};
  for (int k = 0; pn[k].name; ++k) {
    set_uniform(Uv, pn[k].name, &P[pn[k].idx], 1, royale_last_vs_uniforms);
    set_uniform(Uf, pn[k].name, &P[pn[k].idx], 1, royale_last_fs_uniforms);
  }
  set_uniform(Uv, "MVPMatrix", ident, 16, royale_last_vs_uniforms);
  set_uniform(Uv, "OutputSize", out_size, 2, royale_last_vs_uniforms);
  set_uniform(Uv, "TextureSize", tex_size, 2, royale_last_vs_uniforms);
  set_uniform(Uv, "InputSize", tex_size, 2, royale_last_vs_uniforms);
  set_uniform(Uf, "OutputSize", out_size, 2, royale_last_fs_uniforms);
  set_uniform(Uf, "TextureSize", tex_size, 2, royale_last_fs_uniforms);
  set_uniform(Uf, "InputSize", tex_size, 2, royale_last_fs_uniforms);
  /* the quad: BL, BR, TR, TL (ShaderEngine.cpp:2945-2960) */
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  float vout[4][48];
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    memset(vout[v], 0, sizeof vout[v]);
    royale_last_vs(Uv, in, vout[v], 0);
  }
  /* eye_pos_local etc. make the quad take the two-triangle path whatever the target format (see o_pass_royale_last) */
  o_varying pl[32];
  int n_in = 0;
  for (const void* t0 = royale_last_fs_inputs; royale_last_fs_inputs[n_in].name; ++n_in) (void)t0;
  for (int k = 0; k < n_in; ++k) {
    const int s = royale_last_fs_inputs[k].off;
    pl[k] = o_varying_setup(vout[0][s], vout[1][s], vout[2][s], vout[3][s], W, H, O_FMT_SRGB8);
  }
  const int mip = a->in->n_levels > 1;
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      float in[32] = {0}, out[4], rec[4][MAX_TAPS][2];
      tex_ctx ctx = {a->in, TEX_PLAIN, 0, 0, {0, 0, 0, 0}};
      /* one set of differences per quad, at its top-left pixel: (x0, y0), (x0 + 1, y0) and (x0, y0 + 1) as THIS pixel's
       * triangle extrapolates them (rc_sampler.c; f32_crt_royale_fake_bloom_geom_cylinder_tilt tells this from differences
       * along the pixel's own row and column) */
      const int x0 = x & ~1, y0 = y & ~1;
      const int qx[5] = {x0, x0 + 1, x0, x0, x}, qy[5] = {y0, y0, y0, y0 + 1, y};
      for (int e = mip ? 0 : 4; e < 5; ++e) {
        for (int k = 0; k < n_in; ++k) {
          const int s = royale_last_fs_inputs[k].off;
          /* flat varyings take the provoking vertex: every one of this shader is uniform-only, the same at all four */
          in[s] = royale_last_fs_inputs[k].flat ? vout[0][s] : o_varying_at(&pl[k], qx[e], qy[e], lo);
        }
        ctx.n = 0;
        if (e < 4) {
          ctx.mode = TEX_RECORD;
          ctx.rec = rec[e];
        } else if (mip) {
          ctx.mode = TEX_MIP;
          for (int q = 0; q < 4; ++q) ctx.q[q] = (const float (*)[2])rec[q];
        }
        royale_last_fs(Uf, in, out, &ctx);
      }
      o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}
