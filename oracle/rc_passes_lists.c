/* TEST INFRASTRUCTURE - see rc_oracle.h.
 *
 * Two more passes whose bodies are the GL's own instruction lists (oracle/glrun/nir2c.py, recipe gen_lists.sh; see rc_passes_lcd.c
 * and rc_passes_royale_last.c for the idea), both hubs of the reference's preset tree:
 *   crt/shaders/tvout-tweaks.glsl  (39 presets) - FS 99-214: per-pixel sinc resampling of Y, I and Q at three signal bandwidths
 *       (32 sin per pixel), composite cross-talk and TV colour levels behind run-time switches; ~840 operations, 17 branches.
 *   misc/image-adjustment.glsl     (48 presets) - gamma, saturation / contrast / luminance, channel gains, overscan masks,
 *       film grain seeded by FrameCount, sharpen; its vertex stage zooms and shifts the coordinates (~40 operations).  ia_FLIP_HORZ / _VERT
 *       are NOT restated: the shader flips the quad's clip-space position (1 - x puts it at [0, 2]), i.e. clipped geometry over half the target.
 * Uniforms are filled by name from the tables the generator emits; parameters arrive in #pragma order.
 * Pinned by tests/golden/tvout_* and image_adjustment_* (llvmpipe, 8-bit and float). */
#include <math.h>
#include <string.h>

#include "rc_oracle.h"

#define RCN_FN static
static inline float RCN_BITS(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define RCN_ABS(x) fabsf(x)
#define RCN_RCP(x) (1.0f / (x))
#define RCN_DIV(a, b) ((a) / (b))
#define RCN_FLOOR(x) floorf(x)
static inline float rcn_fract(float x) { return x - floorf(x); }
#define RCN_FRACT(x) rcn_fract(x)
static inline float rcn_min(float a, float b) { return b != b ? a : (a < b ? a : b); }   /* gallivm's fmin / fmax: the operand that is not NaN */
static inline float rcn_max(float a, float b) { return b != b ? a : (a > b ? a : b); }
#define RCN_MIN(a, b) rcn_min(a, b)
#define RCN_MAX(a, b) rcn_max(a, b)
#define RCN_POW(a, b) o_pow(a, b)
#define RCN_SIN(x) o_sin(x)
#define RCN_SQRT(x) sqrtf(x)
#define RCN_EXP2(x) o_exp2(x)
static void rcn_tex(void* ctx, float u, float v, float* dst) {
  const o_vec4 r = o_sample((const o_tex*)ctx, u, v);
  dst[0] = r.x; dst[1] = r.y; dst[2] = r.z; dst[3] = r.w;
}
#define RCN_TEX(ctx, unit, u, v, dst) rcn_tex(ctx, u, v, dst)

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wunused-but-set-variable"
#include "gen/tvout_tweaks_fs.inc"
#include "gen/jinc2_sharper_fs.inc"
#include "gen/crt_lottes_fs.inc"
#include "gen/fakelottes_fs.inc"
#include "gen/side_by_side_vs.inc"
#include "gen/side_by_side_fs.inc"
#include "gen/sameboy_lcd_fs.inc"
#include "gen/crt_consumer_fs.inc"
#include "gen/reverse_aa_fs.inc"
#include "gen/advanced_aa_vs.inc"
#include "gen/advanced_aa_fs.inc"
#include "gen/image_adjustment_vs.inc"
#include "gen/image_adjustment_fs.inc"
#pragma GCC diagnostic pop

typedef struct { const char* name; int off, n, flat; } utab;
static void put(float* U, const void* table, const char* name, const float* v, int n) {
  for (const utab* t = table; t->name; ++t)
    if (!strcmp(t->name, name)) {
      for (int k = 0; k < n && k < t->n; ++k) U[t->off + k] = v[k];
      return;
    }
}
static void put_sizes(float* U, const void* table, const o_pass_args* a, int pass3_rule) {
  const int H = a->out_h;
  const float os[2] = {(float)a->out_w, (float)H}, is[2] = {(float)a->in->w, (float)a->in->h};
  /* the reference hands pass index 3 TextureSize.y = the TARGET's height when the pass scales its height (ShaderEngine.cpp:2418-2421) */
  const float ts[2] = {is[0], (pass3_rule && a->pass_index == 3 && H != a->in->h) ? (float)H : is[1]};
  const float fc = (float)a->frame_count;
  put(U, table, "OutputSize", os, 2);
  put(U, table, "InputSize", is, 2);
  put(U, table, "TextureSize", ts, 2);
  put(U, table, "FrameCount", &fc, 1);   /* an int uniform: the lists take it as a float holding its value (nir2c.py, i2f32) */
}

void o_pass_tvout_tweaks(const o_pass_args* a) {
  static const char* const names[6] = {"TVOUT_RESOLUTION", "TVOUT_COMPOSITE_CONNECTION", "TVOUT_TV_COLOR_LEVELS", "TVOUT_RESOLUTION_Y", "TVOUT_RESOLUTION_I",
                                       "TVOUT_RESOLUTION_Q"};
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float U[32] = {0};
  put_sizes(U, tvout_tweaks_fs_uniforms, a, 1);
  for (int k = 0; k < 6; ++k) put(U, tvout_tweaks_fs_uniforms, names[k], &a->params[k], 1);
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)};
      float out[4];
      tvout_tweaks_fs(U, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}

void o_pass_image_adjustment(const o_pass_args* a) {
  static const char* const names[23] = {"ia_target_gamma", "ia_monitor_gamma", "ia_overscan_percent_x", "ia_overscan_percent_y", "ia_saturation", "ia_contrast",
                                        "ia_luminance", "ia_black_level", "ia_bright_boost", "ia_R", "ia_G", "ia_B", "ia_ZOOM", "ia_XPOS", "ia_YPOS", "ia_TOPMASK",
                                        "ia_BOTMASK", "ia_LMASK", "ia_RMASK", "ia_GRAIN_STR", "ia_SHARPEN", "ia_FLIP_HORZ", "ia_FLIP_VERT"};
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float Uv[64] = {0}, Uf[64] = {0};
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  put(Uv, image_adjustment_vs_uniforms, "MVPMatrix", ident, 16);
  put_sizes(Uv, image_adjustment_vs_uniforms, a, 1);
  put_sizes(Uf, image_adjustment_fs_uniforms, a, 1);
  for (int k = 0; k < 23; ++k) {
    put(Uv, image_adjustment_vs_uniforms, names[k], &a->params[k], 1);
    put(Uf, image_adjustment_fs_uniforms, names[k], &a->params[k], 1);
  }
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};   /* BL, BR, TR, TL */
  float vout[4][48];
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    memset(vout[v], 0, sizeof vout[v]);
    image_adjustment_vs(Uv, in, vout[v], 0);
  }
  o_varying pl[2];
  for (int c = 0; c < 2; ++c) pl[c] = o_varying_setup(vout[0][c], vout[1][c], vout[2][c], vout[3][c], W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&pl[0], x, y, lo), o_varying_at(&pl[1], x, y, lo)};
      float out[4];
      image_adjustment_fs(Uf, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}

/* windowed/shaders/jinc2-sharper.glsl (windowed/jinc2-sharper.glslp, the tvout-jinc-sharpen presets, 10 in all): a 4x4 jinc-windowed-jinc
 * resampler - 16 taps, each weight two sin of a sqrt distance - with anti-ringing clamp; ~430 operations, the GL's instruction list.
 * VS: TEX0 = TexCoord * 1.0001.  No parameters. */
void o_pass_jinc2_sharper(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float U[8] = {0};
  put_sizes(U, jinc2_sharper_fs_uniforms, a, 1);
  o_varying tu = o_varying_setup(0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, 0.f * 1.0001f, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * 1.0001f, 0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)};
      float out[4] = {0.f, 0.f, 0.f, 0.f};   /* the shader never writes alpha: the GL stores 0 */
      jinc2_sharper_fs(U, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}

/* crt/shaders/crt-lottes.glsl (crt/crt-lottes.glslp; ~2 500 operations: 31 taps under gaussian pixel / scanline / bloom kernels, tube warp, four
 * shadow masks on gl_FragCoord, 48 branches) and crt/shaders/fakelottes.glsl (crt/fakelottes.glslp: its one-tap cousin).  Both read
 * gl_FragCoord (pixel + 0.5; handed over in the list's input slots 32..35) through gl_FbWposYTransform = (1, 0, -1, height).
 * params in #pragma order. */
static void run_fragcoord_list_k(const o_pass_args* a, void (*fs)(const float*, const float*, float*, void*), const void* table, const char* const* names, int n,
                                 float k);
static void run_fragcoord_list(const o_pass_args* a, void (*fs)(const float*, const float*, float*, void*), const void* table, const char* const* names, int n) {
  run_fragcoord_list_k(a, fs, table, names, n, 1.0f);
}
/* k: the vertex stage's TEX0 = TexCoord * k */
static void run_fragcoord_list_k(const o_pass_args* a, void (*fs)(const float*, const float*, float*, void*), const void* table, const char* const* names, int n,
                                 float k) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float U[64] = {0};
  const float ytr[4] = {1.0f, 0.0f, -1.0f, (float)H};
  put_sizes(U, table, a, 1);
  put(U, table, "gl_FbWposYTransform", ytr, 4);
  for (int k = 0; k < n; ++k) put(U, table, names[k], &a->params[k], 1);
  o_varying tu = o_varying_setup(0.f * k, 1.f * k, 1.f * k, 0.f * k, W, H, a->out_fmt), tv = o_varying_setup(0.f * k, 0.f * k, 1.f * k, 1.f * k, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      float in[36] = {0}, out[4] = {0.f, 0.f, 0.f, 0.f};
      in[0] = o_varying_at(&tu, x, y, lo);
      in[1] = o_varying_at(&tv, x, y, lo);
      in[32] = (float)x + 0.5f; in[33] = (float)y + 0.5f; in[34] = 0.5f; in[35] = 1.0f;
      fs(U, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}
void o_pass_crt_lottes(const o_pass_args* a) {
  static const char* const names[13] = {"hardScan", "hardPix", "warpX", "warpY", "maskDark", "maskLight", "scaleInLinearGamma", "shadowMask", "brightBoost", "hardBloomPix", "hardBloomScan", "bloomAmount", "shape"};
  run_fragcoord_list(a, crt_lottes_fs, crt_lottes_fs_uniforms, names, 13);
}
void o_pass_fakelottes(const o_pass_args* a) {
  static const char* const names[10] = {"shadowMask", "SCANLINE_SINE_COMP_B", "warpX", "warpY", "maskDark", "maskLight", "crt_gamma", "monitor_gamma", "SCANLINE_SINE_COMP_A", "SCANLINE_BASE_BRIGHTNESS"};
  run_fragcoord_list(a, fakelottes_fs, fakelottes_fs_uniforms, names, 10);
}

/* stereoscopic-3d/shaders/side-by-side-simple.glsl (stereoscopic-3d/side-by-side.glslp, sbs-{flat,warp}-mobile-16x9.glslp): the frame twice, one copy
 * per eye, each placed and lens-warped; an eye's sample is taken only where its coordinate falls inside the frame (texture() inside branches),
 * the two are added.  VS (zoom, width / height, horizontal placement) and FS are the GL's instruction lists.  params in #pragma order. */
void o_pass_side_by_side(const o_pass_args* a) {
  static const char* const names[9] = {"eye_sep", "y_loc", "BOTH", "ana_zoom", "WIDTH", "HEIGHT", "warpX", "warpY", "pulfrich"};
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float Uv[64] = {0}, Uf[32] = {0};
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  put(Uv, side_by_side_vs_uniforms, "MVPMatrix", ident, 16);
  put_sizes(Uv, side_by_side_vs_uniforms, a, 1);
  put_sizes(Uf, side_by_side_fs_uniforms, a, 1);
  for (int k = 0; k < 9; ++k) {
    put(Uv, side_by_side_vs_uniforms, names[k], &a->params[k], 1);
    put(Uf, side_by_side_fs_uniforms, names[k], &a->params[k], 1);
  }
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  float vout[4][48];
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    memset(vout[v], 0, sizeof vout[v]);
    side_by_side_vs(Uv, in, vout[v], 0);
  }
  o_varying pl[2];
  for (int c = 0; c < 2; ++c) pl[c] = o_varying_setup(vout[0][c], vout[1][c], vout[2][c], vout[3][c], W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&pl[0], x, y, lo), o_varying_at(&pl[1], x, y, lo)};
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      side_by_side_fs(Uf, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}

/* handheld/shaders/sameboy-lcd.glsl (handheld/sameboy-lcd.glslp, sameboy-lcd-gbc-color-motionblur.glslp): SameBoy's LCD filter - nine taps around
 * the pixel, the sub-pixel position picking which neighbours and in what proportion, a scanline term; ~265 operations, 7 branches with taps inside.
 * params: COLOR_LOW, COLOR_HIGH, SCANLINE_DEPTH.  VS: TEX0 = TexCoord. */
void o_pass_sameboy_lcd(const o_pass_args* a) {
  static const char* const names[3] = {"COLOR_LOW", "COLOR_HIGH", "SCANLINE_DEPTH"};
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float U[8] = {0};
  put_sizes(U, sameboy_lcd_fs_uniforms, a, 1);
  for (int k = 0; k < 3; ++k) put(U, sameboy_lcd_fs_uniforms, names[k], &a->params[k], 1);
  o_varying tu = o_varying_setup(0.f, 1.f, 1.f, 0.f, W, H, a->out_fmt), tv = o_varying_setup(0.f, 0.f, 1.f, 1.f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)};
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      sameboy_lcd_fs(U, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}

/* crt/shaders/crt-consumer.glsl (crt/crt-consumer.glslp; ~970 operations, 24 taps, 19 branches: blur, warp, corner, beam / scanline profiles, three
 * mask families on gl_FragCoord, glow, noise seeded by FrameCount, vignette).  VS: TEX0 = TexCoord * 1.0001.  33 params in #pragma order. */
void o_pass_crt_consumer(const o_pass_args* a) {
  static const char* const names[33] = {"blurx", "blury", "warpx", "warpy", "corner", "smoothness", "scanlow", "scanhigh", "beamlow", "beamhigh", "brightboost1", "brightboost2", "Shadowmask", "masksize", "MaskDark", "MaskLight", "slotmask", "slotwidth", "double_slot", "slotms", "GAMMA_IN", "GAMMA_OUT", "glow", "Size", "sat", "contrast", "nois", "WP", "inter", "vignette", "vpower", "vstr", "alloff"};
  run_fragcoord_list_k(a, crt_consumer_fs, crt_consumer_fs_uniforms, names, 33, 1.0001f);
}

/* anti-aliasing/shaders/reverse-aa.glsl (anti-aliasing/reverse-aa.glslp): Christoph Feck's reverse anti-aliasing - a 3x3 neighbourhood, tilt
 * estimates clamped by the local range, two sub-pixel corrections; ~230 operations.  VS: TEX0 = TexCoord * 1.0001.  params: REVERSEAA_SHARPNESS. */
void o_pass_reverse_aa(const o_pass_args* a) {
  static const char* const names[1] = {"REVERSEAA_SHARPNESS"};
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float U[8] = {0};
  put_sizes(U, reverse_aa_fs_uniforms, a, 1);
  put(U, reverse_aa_fs_uniforms, names[0], &a->params[0], 1);
  o_varying tu = o_varying_setup(0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, 0.f * 1.0001f, W, H, a->out_fmt);
  o_varying tv = o_varying_setup(0.f * 1.0001f, 0.f * 1.0001f, 1.f * 1.0001f, 1.f * 1.0001f, W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      const float in[2] = {o_varying_at(&tu, x, y, lo), o_varying_at(&tv, x, y, lo)};
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      reverse_aa_fs(U, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}

/* anti-aliasing/shaders/advanced-aa.glsl (anti-aliasing/advanced-aa.glslp): nine taps at coordinates its vertex stage prepares (six varyings: the
 * pixel and its neighbours at 1 / AA_RESOLUTION, or 1 / TextureSize when the parameters are 0), edge-directed blend.  Both stages are the GL's lists.
 * params: AA_RESOLUTION_X, AA_RESOLUTION_Y. */
void o_pass_advanced_aa(const o_pass_args* a) {
  unsigned csr = o_fp_enter();
  const int W = a->out_w, H = a->out_h;
  float Uv[32] = {0}, Uf[4] = {0};
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  put(Uv, advanced_aa_vs_uniforms, "MVPMatrix", ident, 16);
  put_sizes(Uv, advanced_aa_vs_uniforms, a, 1);
  put(Uv, advanced_aa_vs_uniforms, "AA_RESOLUTION_X", &a->params[0], 1);
  put(Uv, advanced_aa_vs_uniforms, "AA_RESOLUTION_Y", &a->params[1], 1);
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  float vout[4][48];
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    memset(vout[v], 0, sizeof vout[v]);
    advanced_aa_vs(Uv, in, vout[v], 0);
  }
  o_varying pl[6];
  for (int c = 0; c < 6; ++c) pl[c] = o_varying_setup(vout[0][c], vout[1][c], vout[2][c], vout[3][c], W, H, a->out_fmt);
  for (int y = a->y0; y < a->y1; ++y)
    for (int x = 0; x < W; ++x) {
      const int lo = o_lower_tri(x, y, W, H);
      float in[6], out[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < 6; ++c) in[c] = o_varying_at(&pl[c], x, y, lo);
      advanced_aa_fs(Uf, in, out, (void*)a->in);
      const o_vec4 o = {out[0], out[1], out[2], out[3]};
      o_store_pixel(a, x, y, o);
    }
  o_fp_leave(csr);
}
