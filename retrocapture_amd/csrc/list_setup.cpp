#include "list_setup.h"

#include <cmath>
#include <cstring>

#include "kernels/list_params.h"
#include "rc_log.h"
#include "varying.h"

namespace rc {
namespace {

// the generated tables (uniform name -> dword) and image-adjustment's vertex stage, with host primitives
#define RCN_FN static
#define RCN_BITS(u) rcd::bits2f(u)
#define RCN_DIV(a, b) ((a) / (b))
#define RCN_RCP(x) (1.0f / (x))
#define RCN_TEX(ctx, unit, u, v, dst) ((void)0)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#pragma clang diagnostic ignored "-Wunused-variable"
#include "kernels/gen/image_adjustment_vs.inc"
#include "kernels/gen/side_by_side_vs.inc"
#include "kernels/gen/advanced_aa_vs.inc"
#define RCN_TABLES_ONLY
#include "kernels/gen/image_adjustment_fs.inc"
#include "kernels/gen/tvout_tweaks_fs.inc"
#include "kernels/gen/jinc2_sharper_fs.inc"
#include "kernels/gen/crt_lottes_fs.inc"
#include "kernels/gen/fakelottes_fs.inc"
#include "kernels/gen/side_by_side_fs.inc"
#include "kernels/gen/sameboy_lcd_fs.inc"
#include "kernels/gen/crt_consumer_fs.inc"
#include "kernels/gen/reverse_aa_fs.inc"
#undef RCN_TABLES_ONLY
#pragma clang diagnostic pop

template <class T>
void put(float* U, const T* table, const char* name, const float* v, int n, int cap) {
  for (; table->name; ++table)
    if (!std::strcmp(table->name, name)) {
      for (int k = 0; k < n && k < table->n && table->off + k < cap; ++k) U[table->off + k] = v[k];
      return;
    }
}
// TextureSize / InputSize / OutputSize as the reference hands them over: pass index 3 gets TextureSize.y = the TARGET's height when it
// scales its height (ShaderEngine.cpp:2418-2421)
template <class T>
void putSizes(float* U, const T* table, const PassGeometry& g, int cap) {
  const float os[2] = {(float)g.out_w, (float)g.out_h}, is[2] = {(float)g.in_w, (float)g.in_h};
  const float ts[2] = {is[0], (g.pass_index == 3 && g.out_h != g.in_h) ? (float)g.out_h : is[1]};
  put(U, table, "OutputSize", os, 2, cap);
  put(U, table, "InputSize", is, 2, cap);
  put(U, table, "TextureSize", ts, 2, cap);
}

const char* const kTvoutNames[6] = {"TVOUT_RESOLUTION", "TVOUT_COMPOSITE_CONNECTION", "TVOUT_TV_COLOR_LEVELS", "TVOUT_RESOLUTION_Y", "TVOUT_RESOLUTION_I",
                                    "TVOUT_RESOLUTION_Q"};
const char* const kImageAdjNames[23] = {"ia_target_gamma", "ia_monitor_gamma", "ia_overscan_percent_x", "ia_overscan_percent_y", "ia_saturation", "ia_contrast",
                                        "ia_luminance", "ia_black_level", "ia_bright_boost", "ia_R", "ia_G", "ia_B", "ia_ZOOM", "ia_XPOS", "ia_YPOS", "ia_TOPMASK",
                                        "ia_BOTMASK", "ia_LMASK", "ia_RMASK", "ia_GRAIN_STR", "ia_SHARPEN", "ia_FLIP_HORZ", "ia_FLIP_VERT"};

}  // namespace

void setupTvoutTweaks(const PassGeometry& g, rcd::PassLaunch& L) {
  L.plane[0] = planeU(1.0f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, g.out_fmt);
  float* U = L.params + kListU0;
  for (int k = 0; k < kTvoutU; ++k) U[k] = 0.0f;
  putSizes(U, tvout_tweaks_fs_uniforms, g, kTvoutU);
  for (int k = 0; k < 6; ++k) put(U, tvout_tweaks_fs_uniforms, kTvoutNames[k], &L.params[k], 1, kTvoutU);
}

namespace {
const char* const kLottesNames[13] = {"hardScan", "hardPix", "warpX", "warpY", "maskDark", "maskLight", "scaleInLinearGamma", "shadowMask", "brightBoost", "hardBloomPix", "hardBloomScan", "bloomAmount", "shape"};
const char* const kFakeLottesNames[10] = {"shadowMask", "SCANLINE_SINE_COMP_B", "warpX", "warpY", "maskDark", "maskLight", "crt_gamma", "monitor_gamma", "SCANLINE_SINE_COMP_A", "SCANLINE_BASE_BRIGHTNESS"};
// a list that reads gl_FragCoord: TEX0 = TexCoord, gl_FbWposYTransform = (1, 0, -1, height)
template <class T>
void setupFragCoordList(const PassGeometry& g, rcd::PassLaunch& L, const T* table, const char* const* names, int n, int nu) {
  L.plane[0] = planeU(1.0f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, g.out_fmt);
  float* U = L.params + kListU0;
  for (int k = 0; k < nu; ++k) U[k] = 0.0f;
  putSizes(U, table, g, nu);
  const float ytr[4] = {1.0f, 0.0f, -1.0f, (float)g.out_h};
  put(U, table, "gl_FbWposYTransform", ytr, 4, nu);
  for (int k = 0; k < n; ++k) put(U, table, names[k], &L.params[k], 1, nu);
}
}  // namespace
void setupCrtConsumer(const PassGeometry& g, rcd::PassLaunch& L) {
  static const char* const names[33] = {"blurx", "blury", "warpx", "warpy", "corner", "smoothness", "scanlow", "scanhigh", "beamlow", "beamhigh", "brightboost1", "brightboost2", "Shadowmask", "masksize", "MaskDark", "MaskLight", "slotmask", "slotwidth", "double_slot", "slotms", "GAMMA_IN", "GAMMA_OUT", "glow", "Size", "sat", "contrast", "nois", "WP", "inter", "vignette", "vpower", "vstr", "alloff"};
  setupFragCoordList(g, L, crt_consumer_fs_uniforms, names, 33, kConsumerU);
  L.plane[0] = planeU(1.0001f, g.out_w, g.out_h, g.out_fmt);   // VS: TEX0 = TexCoord * 1.0001
  L.plane[1] = planeV(1.0001f, g.out_w, g.out_h, g.out_fmt);
}
void setupCrtLottes(const PassGeometry& g, rcd::PassLaunch& L) { setupFragCoordList(g, L, crt_lottes_fs_uniforms, kLottesNames, 13, kLottesU); }
void setupFakeLottes(const PassGeometry& g, rcd::PassLaunch& L) { setupFragCoordList(g, L, fakelottes_fs_uniforms, kFakeLottesNames, 10, kFakeLottesU); }

// side-by-side-simple.glsl: uniform block by name, vertex stage (zoom, width / height, placement) at the quad's vertices
void setupSideBySide(const PassGeometry& g, rcd::PassLaunch& L) {
  static const char* const names[9] = {"eye_sep", "y_loc", "BOTH", "ana_zoom", "WIDTH", "HEIGHT", "warpX", "warpY", "pulfrich"};
  float* U = L.params + kListU0;
  for (int k = 0; k < kSbsU; ++k) U[k] = 0.0f;
  putSizes(U, side_by_side_fs_uniforms, g, kSbsU);
  float Uv[64] = {};
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  put(Uv, side_by_side_vs_uniforms, "MVPMatrix", ident, 16, 64);
  putSizes(Uv, side_by_side_vs_uniforms, g, 64);
  for (int k = 0; k < 9; ++k) {
    put(U, side_by_side_fs_uniforms, names[k], &L.params[k], 1, kSbsU);
    put(Uv, side_by_side_vs_uniforms, names[k], &L.params[k], 1, 64);
  }
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  float out[4][48] = {};
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    side_by_side_vs(Uv, in, out[v], nullptr);
  }
  for (int c = 0; c < 2; ++c) L.plane[c] = makePlane(out[0][c], out[1][c], out[2][c], out[3][c], g.out_w, g.out_h, g.out_fmt);
}

void setupSameboyLcd(const PassGeometry& g, rcd::PassLaunch& L) {
  static const char* const names[3] = {"COLOR_LOW", "COLOR_HIGH", "SCANLINE_DEPTH"};
  L.plane[0] = planeU(1.0f, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, g.out_fmt);
  float* U = L.params + kListU0;
  for (int k = 0; k < kSameboyLcdU; ++k) U[k] = 0.0f;
  putSizes(U, sameboy_lcd_fs_uniforms, g, kSameboyLcdU);
  for (int k = 0; k < 3; ++k) put(U, sameboy_lcd_fs_uniforms, names[k], &L.params[k], 1, kSameboyLcdU);
}

// advanced-aa.glsl: the vertex stage (the pixel's and its neighbours' coordinates, six varyings) at the quad's vertices
void setupAdvancedAa(const PassGeometry& g, rcd::PassLaunch& L) {
  float Uv[32] = {};
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  put(Uv, advanced_aa_vs_uniforms, "MVPMatrix", ident, 16, 32);
  putSizes(Uv, advanced_aa_vs_uniforms, g, 32);
  put(Uv, advanced_aa_vs_uniforms, "AA_RESOLUTION_X", &L.params[0], 1, 32);
  put(Uv, advanced_aa_vs_uniforms, "AA_RESOLUTION_Y", &L.params[1], 1, 32);
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  float out[4][48] = {};
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    advanced_aa_vs(Uv, in, out[v], nullptr);
  }
  for (int c = 0; c < 6; ++c) L.plane[c] = makePlane(out[0][c], out[1][c], out[2][c], out[3][c], g.out_w, g.out_h, g.out_fmt);
}

void setupReverseAa(const PassGeometry& g, rcd::PassLaunch& L) {
  L.plane[0] = planeU(1.0001f, g.out_w, g.out_h, g.out_fmt);   // VS: TEX0 = TexCoord * 1.0001
  L.plane[1] = planeV(1.0001f, g.out_w, g.out_h, g.out_fmt);
  float* U = L.params + kListU0;
  for (int k = 0; k < kReverseAaU; ++k) U[k] = 0.0f;
  putSizes(U, reverse_aa_fs_uniforms, g, kReverseAaU);
  put(U, reverse_aa_fs_uniforms, "REVERSEAA_SHARPNESS", &L.params[0], 1, kReverseAaU);
}

void setupJinc2Sharper(const PassGeometry& g, rcd::PassLaunch& L) {
  L.plane[0] = planeU(1.0001f, g.out_w, g.out_h, g.out_fmt);   // VS: TEX0 = TexCoord * 1.0001
  L.plane[1] = planeV(1.0001f, g.out_w, g.out_h, g.out_fmt);
  float* U = L.params + kListU0;
  for (int k = 0; k < kJinc2U; ++k) U[k] = 0.0f;
  putSizes(U, jinc2_sharper_fs_uniforms, g, kJinc2U);
}

void setupImageAdjustment(const PassGeometry& g, rcd::PassLaunch& L) {
  float* U = L.params + kListU0;
  for (int k = 0; k < kImageAdjU; ++k) U[k] = 0.0f;
  putSizes(U, image_adjustment_fs_uniforms, g, kImageAdjU);
  for (int k = 0; k < 23; ++k) put(U, image_adjustment_fs_uniforms, kImageAdjNames[k], &L.params[k], 1, kImageAdjU);
  // vertex stage (image-adjustment.glsl VS: overscan, zoom, shift, flip of TexCoord) at the quad's vertices BL, BR, TR, TL
  float Uv[64] = {};
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  put(Uv, image_adjustment_vs_uniforms, "MVPMatrix", ident, 16, 64);
  putSizes(Uv, image_adjustment_vs_uniforms, g, 64);
  for (int k = 0; k < 23; ++k) put(Uv, image_adjustment_vs_uniforms, kImageAdjNames[k], &L.params[k], 1, 64);
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  float out[4][48] = {};
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    image_adjustment_vs(Uv, in, out[v], nullptr);
  }
  for (int c = 0; c < 2; ++c) L.plane[c] = makePlane(out[0][c], out[1][c], out[2][c], out[3][c], g.out_w, g.out_h, g.out_fmt);
}

}  // namespace rc
