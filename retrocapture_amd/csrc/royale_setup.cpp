#include "royale_setup.h"

#include <cmath>

#include <cstring>
#include <string>

#include "kernels/royale_params.h"
#include "rc_log.h"
#include "varying.h"

namespace rc {
namespace {

using rcd::PassLaunch;

inline float minps(float a, float b) { return a < b ? a : b; }  // SSE min/max: NaN -> second operand
inline float maxps(float a, float b) { return a > b ? a : b; }
inline float clampf(float x, float lo, float hi) { return minps(maxps(x, lo), hi); }

// is_interlaced() of the royale shaders (first-pass file 4723-4750): interlace_detect = true,
// interlace_1080i = false
bool isInterlaced(float lines) { return lines > 288.5f && lines < 576.5f; }

// exp() of a compile-time constant as the GL's compiler folds it: exp2f(x * log2(e)) in float
float constExp(float x) { return exp2f(x * 1.4426950408889634f); }

struct V2 { float x, y; };

// get_resized_mask_tile_size (mask-resize-vertical.glsl 2995-3046) for mask_sample_mode 0,
// mask_specify_num_triads 0, mask_triad_size_desired 3, square 64x64 source tile
V2 resizedMaskTileSize(float out_x, float out_y) {
  const float temp = minps(8.0f * 3.0f, 64.0f);
  const float min_tile = 16.0f;
  const float max_x = out_x / 2.0f, max_y = out_y / 2.0f;  // mask_resize_num_tiles = 2
  const float cx = clampf(temp * 1.0f, min_tile * 1.0f, max_x), cy = clampf(temp * 1.0f, min_tile * 1.0f, max_y);
  const float x_from_y = cy * 1.0f, y_from_x = cy;
  const float fix_zero = 0.0000152587890625f;
  return {std::floor(minps(cx, x_from_y) + fix_zero), std::floor(minps(cy, y_from_x) + fix_zero)};
}

float bloomSigmaRuntime(float ox, float oy) {  // brightpass.glsl 6616-6649
  V2 tile = resizedMaskTileSize(ox * 0.0625f, oy * 0.0625f);
  const float triad = tile.x / 8.0f;
  const float thresh = 1.0f / 256.0f;
  return -0.05168f + 0.6113f * triad - 1.122f * triad * std::sqrt(0.000416f + thresh);
}

// get_fast_gaussian_weight_sum_inv (bloom-vertical.glsl 6624-6628) - a run-time expression in
// the fragment shader, so it uses the GL's exp, not libm's
float centerWeight(float sigma) {
  return minps(rcd::exp_(rcd::exp_(0.348348412457428f / (sigma - 0.0860587260734721f))), 0.399334576340352f / sigma);
}

const float kMaskAmplify = 1.0f / (46.0f / 255.0f);  // 1 / mask_slot_avg_color

// PassPrev<n>{Input,Texture}Size as the reference engine sets them (ShaderEngine.cpp:1191-1227)
struct PrevSizes { float in_w, in_h, tex_w, tex_h; };
PrevSizes prevSizes(const PassGeometry& g, int n) {
  const int p = g.pass_index - n;
  PrevSizes s{0, 0, 0, 0};
  if (p < 0 || p >= g.n_passes) return s;
  s.tex_w = (float)g.chain_w[p];
  s.tex_h = (float)g.chain_h[p];
  s.in_w = p == 0 ? (float)g.src_w : (float)g.chain_w[p - 1];
  s.in_h = p == 0 ? (float)g.src_h : (float)g.chain_h[p - 1];
  return s;
}

rcd::Plane planeU01(float at0, float at1, const PassGeometry& g, int fmt) { return makePlane(at0, at1, at1, at0, g.out_w, g.out_h, fmt); }
rcd::Plane planeV01(float at0, float at1, const PassGeometry& g, int fmt) { return makePlane(at0, at0, at1, at1, g.out_w, g.out_h, fmt); }

void setupTexCoord(const PassGeometry& g, PassLaunch& L, float k) {
  L.plane[0] = planeU(k, g.out_w, g.out_h, g.out_fmt);
  L.plane[1] = planeV(k, g.out_w, g.out_h, g.out_fmt);
}

// ---- pass 0: VS 4801-4811
void setupFirst(const PassGeometry& g, PassLaunch& L) {
  setupTexCoord(g, L, 1.00001f);
  L.params[RP0_INTERLACED] = isInterlaced((float)g.in_h) ? 1.0f : 0.0f;
}

// ---- pass 1: VS 5912-5938
void setupScanV(const PassGeometry& g, PassLaunch& L) {
  setupTexCoord(g, L, 1.0f);
  // texture_size = TextureSize, video_size = InputSize (glsl :38-39); the reference hands pass index 3
  // TextureSize.y = the target's height when that differs from the input's (ShaderEngine.cpp:2418-2421),
  // which is where this shader sits in crt/crt-royale-ntsc-*.glslp
  const float vsy = (float)g.in_h;
  const float tsy = (g.pass_index == 3 && g.out_h != g.in_h) ? (float)g.out_h : vsy;
  const float y_step = 1.0f + (isInterlaced(vsy) ? 1.0f : 0.0f);
  L.params[RP1_Y_STEP] = y_step;
  L.params[RP1_UV_STEP_Y] = y_step / tsy;
  L.params[RP1_PH] = (vsy / (float)g.out_h) / y_step;
  L.params[RP1_TSY] = tsy;
}

// ---- pass 2: VS 5926-5932
void setupBloomApprox(const PassGeometry& g, PassLaunch& L) {
  const float tsx = (float)g.in_w, tsy = (float)g.in_h;
  PrevSizes p = prevSizes(g, 2);
  const float u1 = ((1.0f * tsx) / tsx) * p.in_w / p.tex_w, v1 = ((1.0f * tsy) / tsy) * p.in_h / p.tex_h;
  const float u0 = ((0.0f * tsx) / tsx) * p.in_w / p.tex_w, v0 = ((0.0f * tsy) / tsy) * p.in_h / p.tex_h;
  L.plane[0] = planeU01(u0, u1, g, g.out_fmt);
  L.plane[1] = planeV01(v0, v1, g, g.out_fmt);
}

// ---- passes 3/4: blur9fast VS 2040-2048, weights 1496-1510 with blur9_std_dev (line 394)
void setupBlur9(const PassGeometry& g, PassLaunch& L, bool horizontal) {
  setupTexCoord(g, L, 1.0f);
  const float sigma = 1.7533203125f;
  const float denom_inv = 0.5f / (sigma * sigma);
  const float w0 = 1.0f, w1 = constExp(-1.0f * denom_inv), w2 = constExp(-4.0f * denom_inv);
  const float w3 = constExp(-9.0f * denom_inv), w4 = constExp(-16.0f * denom_inv);
  const float w12 = w1 + w2, w34 = w3 + w4;
  L.params[RPB_W12] = w12;
  L.params[RPB_W34] = w34;
  L.params[RPB_K12] = 1.0f + w2 / w12;
  L.params[RPB_K34] = 3.0f + w4 / w34;
  // the GLSL compiler rebalances the four-term sum (value read back from the GL)
  L.params[RPB_SUM_INV] = 1.0f / (w0 + 2.0f * ((w1 + w2) + (w3 + w4)));
  const float tsx = (float)g.in_w, tsy = (float)g.in_h;
  L.params[RPB_DX] = horizontal ? (tsx / (float)g.out_w) / tsx : 0.0f;
  L.params[RPB_DY] = horizontal ? 0.0f : (tsy / (float)g.out_h) / tsy;
}
void setupBlur9V(const PassGeometry& g, PassLaunch& L) { setupBlur9(g, L, false); }
void setupBlur9H(const PassGeometry& g, PassLaunch& L) { setupBlur9(g, L, true); }

// ---- pass 5: mask-resize-vertical VS 3280-3305
void setupMaskV(const PassGeometry& g, PassLaunch& L) {
  const float ox = (float)g.out_w, oy = (float)g.out_h, tsx = (float)g.in_w, tsy = (float)g.in_h;
  const float aspect_ratio = 1.313069909f / 1.0f;
  V2 tile = resizedMaskTileSize(oy * aspect_ratio, oy);
  const float pots_x = minps(64.0f, ox), pots_y = tile.y;
  const float tiles_x = ox / pots_x, tiles_y = oy / pots_y;
  L.plane[0] = planeU01(((0.0f * tsx) / tsx) * tiles_x, ((1.0f * tsx) / tsx) * tiles_x, g, g.out_fmt);
  L.plane[1] = planeV01(((0.0f * tsy) / tsy) * tiles_y, ((1.0f * tsy) / tsy) * tiles_y, g, g.out_fmt);
  L.params[RP5_MAG_Y] = pots_y / 64.0f;
}

// ---- pass 6: mask-resize-horizontal VS 3276-3300
void setupMaskH(const PassGeometry& g, PassLaunch& L) {
  const float ox = (float)g.out_w, oy = (float)g.out_h, tsx = (float)g.in_w, tsy = (float)g.in_h;
  V2 tile = resizedMaskTileSize(ox, oy);
  const float tiles_x = ox / tile.x, tiles_y = oy / tile.y;
  const float its_x = minps(64.0f, tsx), its_y = tile.y;
  const float tsuv_x = its_x / tsx, tsuv_y = its_y / tsy;
  L.plane[0] = planeU01((((0.0f * tsx) / tsx) * tiles_x) * tsuv_x, (((1.0f * tsx) / tsx) * tiles_x) * tsuv_x, g, g.out_fmt);
  L.plane[1] = planeV01((((0.0f * tsy) / tsy) * tiles_y) * tsuv_y, (((1.0f * tsy) / tsy) * tiles_y) * tsuv_y, g, g.out_fmt);
  L.params[RP6_MAG_X] = tile.x / its_x;
  L.params[RP6_SRC_DX] = 1.0f / tsx;
  L.params[RP6_TILE_SIZE_UV_X] = tsuv_x;
}

// ---- pass 7: scanlines-horizontal-apply-mask VS 6103-6135
void setupScanH(const PassGeometry& g, PassLaunch& L) {
  const float ox = (float)g.out_w, oy = (float)g.out_h, tsx = (float)g.in_w, tsy = (float)g.in_h;
  PrevSizes p6 = prevSizes(g, 6);
  const float stix = 1.0f / p6.tex_w, stiy = 1.0f / p6.tex_h;
  const float vu1 = (1.0f * tsx) / tsx, vv1 = (1.0f * tsy) / tsy, vu0 = (0.0f * tsx) / tsx, vv0 = (0.0f * tsy) / tsy;
  L.plane[0] = planeU01(vu0, vu1, g, g.out_fmt);
  L.plane[1] = planeV01(vv0, vv1, g, g.out_fmt);
  L.plane[2] = planeU01(vu0 * p6.in_w * stix, vu1 * p6.in_w * stix, g, g.out_fmt);
  L.plane[3] = planeV01(vv0 * p6.in_h * stiy, vv1 * p6.in_h * stiy, g, g.out_fmt);
  V2 tile = resizedMaskTileSize(tsx, tsy);
  const float uvs_x = tile.x / tsx, uvs_y = tile.y / tsy;
  L.params[RP7_TPS_X] = ox / tile.x;
  L.params[RP7_TPS_Y] = oy / tile.y;
  L.params[RP7_START_X] = (0.0f / tile.x) * uvs_x;
  L.params[RP7_START_Y] = (0.0f / tile.y) * uvs_y;
  L.params[RP7_UVS_X] = uvs_x;
  L.params[RP7_UVS_Y] = uvs_y;
  L.params[RP7_SCAN_TW] = p6.tex_w;
  L.params[RP7_SCAN_TH] = p6.tex_h;
  L.params[RP7_SCAN_TIX] = stix;
  L.params[RP7_SCAN_TIY] = stiy;
}

// ---- pass 7 of crt-royale-fake-bloom: the same file with PHOSPHOR_BLOOM_FAKE; blur3x3_tex_uv / halation_tex_uv
// = video_uv * <pass>video_size / <pass>texture_size (VS 6117-6120), BLOOM_APPROX = PassPrev5, HALATION_BLUR = PassPrev3
void setupScanHFake(const PassGeometry& g, PassLaunch& L) {
  setupScanH(g, L);
  const float tsx = (float)g.in_w, tsy = (float)g.in_h;
  const float vu1 = (1.0f * tsx) / tsx, vv1 = (1.0f * tsy) / tsy, vu0 = (0.0f * tsx) / tsx, vv0 = (0.0f * tsy) / tsy;
  PrevSizes b = prevSizes(g, 5), h = prevSizes(g, 3);
  L.plane[4] = planeU01(vu0 * b.in_w / b.tex_w, vu1 * b.in_w / b.tex_w, g, g.out_fmt);
  L.plane[5] = planeV01(vv0 * b.in_h / b.tex_h, vv1 * b.in_h / b.tex_h, g, g.out_fmt);
  L.plane[6] = planeU01(vu0 * h.in_w / h.tex_w, vu1 * h.in_w / h.tex_w, g, g.out_fmt);
  L.plane[7] = planeV01(vv0 * h.in_h / h.tex_h, vv1 * h.in_h / h.tex_h, g, g.out_fmt);
}

// ---- pass 8: brightpass VS 6630-6649
void setupBrightpass(const PassGeometry& g, PassLaunch& L) {
  const float tsx = (float)g.in_w, tsy = (float)g.in_h;
  PrevSizes q = prevSizes(g, 4);
  const float vu1 = (1.0f * tsx) / tsx, vv1 = (1.0f * tsy) / tsy, vu0 = (0.0f * tsx) / tsx, vv0 = (0.0f * tsy) / tsy;
  L.plane[0] = planeU01(vu0 * tsx / tsx, vu1 * tsx / tsx, g, g.out_fmt);
  L.plane[1] = planeV01(vv0 * tsy / tsy, vv1 * tsy / tsy, g, g.out_fmt);
  L.plane[2] = planeU01(vu0 * q.in_w / q.tex_w, vu1 * q.in_w / q.tex_w, g, g.out_fmt);
  L.plane[3] = planeV01(vv0 * q.in_h / q.tex_h, vv1 * q.in_h / q.tex_h, g, g.out_fmt);
  L.params[RP8_CENTER_WEIGHT] = centerWeight(bloomSigmaRuntime((float)g.out_w, (float)g.out_h));
  L.params[RP8_MASK_AMPLIFY] = kMaskAmplify;
}

// tex2Dblur17fast weights (bloom-vertical.glsl 7132-7160) for the run-time sigma
void blur17Params(float sigma, PassLaunch& L) {
  const float denom_inv = 0.5f / (sigma * sigma);
  const float w1 = rcd::exp_(-1.0f * denom_inv), w2 = rcd::exp_(-4.0f * denom_inv), w3 = rcd::exp_(-9.0f * denom_inv);
  const float w4 = rcd::exp_(-16.0f * denom_inv), w5 = rcd::exp_(-25.0f * denom_inv), w6 = rcd::exp_(-36.0f * denom_inv);
  const float w7 = rcd::exp_(-49.0f * denom_inv), w8 = rcd::exp_(-64.0f * denom_inv);
  const float w12 = w1 + w2, w34 = w3 + w4, w56 = w5 + w6, w78 = w7 + w8;
  L.params[RPG_W12] = w12;
  L.params[RPG_W34] = w34;
  L.params[RPG_W56] = w56;
  L.params[RPG_W78] = w78;
  L.params[RPG_K12] = 1.0f + w2 / w12;
  L.params[RPG_K34] = 3.0f + w4 / w34;
  L.params[RPG_K56] = 5.0f + w6 / w56;
  L.params[RPG_K78] = 7.0f + w8 / w78;
  L.params[RPG_SUM_INV] = centerWeight(sigma);
  L.params[RPG_MASK_AMPLIFY] = kMaskAmplify;
}

// ---- pass 9: bloom-vertical VS 3851-3861
void setupBloomV(const PassGeometry& g, PassLaunch& L) {
  setupTexCoord(g, L, 1.0001f);
  blur17Params(bloomSigmaRuntime((float)g.out_w, (float)g.out_h), L);
  const float tsy = (float)g.in_h;
  L.params[RPG_DXY] = (tsy / (float)g.out_h) / tsy;
}

// ---- pass 10: bloom-horizontal-reconstitute VS 6641-6660
void setupBloomH(const PassGeometry& g, PassLaunch& L) {
  const float tsx = (float)g.in_w, tsy = (float)g.in_h;
  setupTexCoord(g, L, 1.0f);
  blur17Params(bloomSigmaRuntime((float)g.out_w, (float)g.out_h), L);
  L.params[RPG_DXY] = 1.0f / tsx;
  PrevSizes m = prevSizes(g, 3), b = prevSizes(g, 2), h = prevSizes(g, 6);
  const float vu1 = (1.0f * tsx) / tsx, vv1 = (1.0f * tsy) / tsy, vu0 = (0.0f * tsx) / tsx, vv0 = (0.0f * tsy) / tsy;
  L.plane[2] = planeU01(vu0 * m.in_w / m.tex_w, vu1 * m.in_w / m.tex_w, g, g.out_fmt);
  L.plane[3] = planeV01(vv0 * m.in_h / m.tex_h, vv1 * m.in_h / m.tex_h, g, g.out_fmt);
  L.plane[4] = planeU01(vu0 * b.in_w / b.tex_w, vu1 * b.in_w / b.tex_w, g, g.out_fmt);
  L.plane[5] = planeV01(vv0 * b.in_h / b.tex_h, vv1 * b.in_h / b.tex_h, g, g.out_fmt);
  L.plane[6] = planeU01(vu0 * h.in_w / h.tex_w, vu1 * h.in_w / h.tex_w, g, g.out_fmt);
  L.plane[7] = planeV01(vv0 * h.in_h / h.tex_h, vv1 * h.in_h / h.tex_h, g, g.out_fmt);
}

// ---- pass 11, general form: the vertex stage (geometry-aa-last-pass.glsl VS 5337-5400 with get_ideal_global_eye_pos
// 4843-5020: eye position and global-to-local matrix from sin / cos of the tilt angles) in the GL's own instruction
// order - kernels/gen/royale_last_vs.inc, generated from Mesa's NIR listing (oracle/glrun/nir2c.py)
namespace lastvs {
#define RCN_FN static
#define RCN_BITS(u) rcd::bits2f(u)
#define RCN_ABS(x) std::fabs(x)
#define RCN_RSQ(x) (1.0f / std::sqrt(x))
#define RCN_RCP(x) (1.0f / (x))
#define RCN_SQRT(x) std::sqrt(x)
#define RCN_SIGN(x) ((x) == 0.0f ? 0.0f : std::copysign(1.0f, (x)))
#define RCN_SIN(x) rcd::sin_(x)
#define RCN_COS(x) rcd::cos_(x)
#define RCN_DIV(a, b) ((a) / (b))
#define RCN_MIN(a, b) vmin(a, b)
#define RCN_MAX(a, b) vmax(a, b)
#define RCN_POW(a, b) ((a) != (a) ? 0.0f : rcd::pow_(a, b))
#define RCN_TEX(ctx, unit, u, v, dst) ((void)0)
inline float vmin(float a, float b) { return b != b ? a : (a < b ? a : b); }   // gallivm's fmin / fmax: the operand that is not NaN
inline float vmax(float a, float b) { return b != b ? a : (a > b ? a : b); }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#pragma clang diagnostic ignored "-Wunused-variable"
#include "kernels/gen/royale_last_vs.inc"
#define RCN_TABLES_ONLY
#include "kernels/gen/royale_last_fs.inc"
#undef RCN_TABLES_ONLY
#pragma clang diagnostic pop
template <class T>
void setUniform(float* U, const T* table, const char* name, const float* v, int n) {
  for (; table->name; ++table)
    if (!std::strcmp(table->name, name)) {
      for (int k = 0; k < n && k < table->n; ++k) U[table->off + k] = v[k];
      return;
    }
}
}  // namespace lastvs

void setupLastGeneral(const PassGeometry& g, PassLaunch& L) {
  using namespace lastvs;
  // the kernel addresses the fragment stage's uniform block by position (pass_royale_last_general.hip): keep the two in step
  static const char* const kFsUniforms[] = {"lcd_gamma", "aa_cubic_c", "geom_mode_runtime", "geom_radius", "geom_view_dist", "border_size",
                                            "border_darkness", "border_compress", "TextureSize", "InputSize"};
  static const int kFsOffsets[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10};
  for (int k = 0; k < 10; ++k)
    if (!royale_last_fs_uniforms[k].name || std::strcmp(royale_last_fs_uniforms[k].name, kFsUniforms[k]) || royale_last_fs_uniforms[k].off != kFsOffsets[k])
      RC_LOG_ERROR("crt-royale last pass: generated uniform layout changed (entry " + std::to_string(k) + ")");
  const float* P = L.params;
  static const struct { const char* name; int idx; } pn[] = {{"geom_mode_runtime", 30}, {"geom_radius", 31}, {"geom_view_dist", 32},
                                                             {"geom_tilt_angle_x", 33}, {"geom_tilt_angle_y", 34}, {"geom_aspect_ratio_x", 35},
                                                             {"geom_aspect_ratio_y", 36}, {"geom_overscan_x", 37}, {"geom_overscan_y", 38}};
  float U[64] = {};
  for (const auto& e : pn) setUniform(U, royale_last_vs_uniforms, e.name, &P[e.idx], 1);
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float tex_size[2] = {(float)g.in_w, (float)g.in_h}, out_size[2] = {(float)g.out_w, (float)g.out_h};
  setUniform(U, royale_last_vs_uniforms, "MVPMatrix", ident, 16);
  setUniform(U, royale_last_vs_uniforms, "OutputSize", out_size, 2);
  setUniform(U, royale_last_vs_uniforms, "TextureSize", tex_size, 2);
  setUniform(U, royale_last_vs_uniforms, "InputSize", tex_size, 2);
  // the quad's vertices: BL, BR, TR, TL (reference ShaderEngine.cpp:2945-2960)
  static const float pos[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}}, uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  float out[4][48] = {};
  for (int v = 0; v < 4; ++v) {
    const float in[8] = {pos[v][0], pos[v][1], 0.0f, 1.0f, uv[v][0], uv[v][1], 0.0f, 1.0f};
    royale_last_vs(U, in, out[v], nullptr);
  }
  L.plane[0] = makePlane(out[0][0], out[1][0], out[2][0], out[3][0], g.out_w, g.out_h, rcd::FMT_SRGB8);   // tex_uv: two-triangle planes (setupLast)
  L.plane[1] = makePlane(out[0][1], out[1][1], out[2][1], out[3][1], g.out_w, g.out_h, rcd::FMT_SRGB8);
  for (int k = 2; k < kLastVaryings; ++k) {
    // everything else is built from uniforms alone: the same bits at the four vertices, a constant plane
    for (int v = 1; v < 4; ++v)
      if (std::memcmp(&out[v][k], &out[0][k], 4)) RC_LOG_ERROR("crt-royale last pass: varying slot " + std::to_string(k) + " differs across the quad");
    L.params[RP11_VARYING0 + k] = out[0][k];
  }
}

// ---- pass 11: geometry-aa-last-pass VS 5337-5400 (flat path); get_aspect_vector 2512-2520
void setupLast(const PassGeometry& g, PassLaunch& L) {
  if (lastIsGeneral(L.params)) {
    setupLastGeneral(g, L);
    return;
  }
  // This vertex shader also emits eye_pos_local, which is NaN in the flat geometry mode; the GL
  // then rasterises the quad as two triangles even on an RGBA8 target (measured), so the planes
  // are the two-triangle ones regardless of the target format.
  L.plane[0] = planeU(1.0f, g.out_w, g.out_h, rcd::FMT_SRGB8);
  L.plane[1] = planeV(1.0f, g.out_w, g.out_h, rcd::FMT_SRGB8);
  const float ar = (float)g.out_w / (float)g.out_h;
  const float gx = minps(ar, 4.0f / 3.0f), gy = 1.0f;
  const float rs = 1.0f / std::sqrt(gx * gx + gy * gy);
  L.params[RP11_ASPECT_X] = gx * rs;
  L.params[RP11_ASPECT_Y] = gy * rs;
}

}  // namespace

void registerRoyaleKernels(std::vector<KernelEntry>& r) {
  const char* R = "crt/shaders/crt-royale/src/crt-royale-";
  static std::vector<std::string> ids;
  auto id = [&](const char* tail) -> const char* {
    ids.push_back(std::string(R) + tail);
    return ids.back().c_str();
  };
  ids.reserve(16);
  r.push_back({id("first-pass-linearize-crt-gamma-bob-fields.glsl"), "royale-first", {}, {}, rck::launch_royale_first, setupFirst, false});
  {
    KernelEntry e{id("scanlines-vertical-interlacing.glsl"), "royale-scanlines-v", {}, {}, rck::launch_royale_scan_v, setupScanV, false};
    // 1:1 geometry runs the expansion-table form (kernels/pass_royale_scan.hip), which lists the pixels it leaves to the
    // general form in this scratch: 256 bytes of counters and one 4-byte entry per pixel, per frame
    e.scratch_bytes = [](const PassGeometry& g) -> uint64_t {
      return (g.in_w == g.out_w && g.in_h == g.out_h) ? 256u + (uint64_t)g.out_w * (uint64_t)g.out_h * 4u : 0u;
    };
    r.push_back(e);
  }
  r.back().texture_height_override = true;
  r.push_back({id("bloom-approx.glsl"), "royale-bloom-approx", {}, {"PassPrev2Texture"}, rck::launch_royale_bloom_approx, setupBloomApprox, false});
  r.push_back({"blurs/blur9fast-vertical.glsl", "blur9fast-v", {}, {}, rck::launch_blur9, setupBlur9V, false});
  r.push_back({"blurs/blur9fast-horizontal.glsl", "blur9fast-h", {}, {}, rck::launch_blur9, setupBlur9H, false});
  {
    // the vertical mask resize samples the mask LUT only (its `Texture` input, pass 4's output, is never read)
    KernelEntry e{id("mask-resize-vertical.glsl"), "royale-mask-v", {}, {"mask_slot_texture_small"}, rck::launch_royale_mask_v, setupMaskV, true};
    e.reads_input = false;
    r.push_back(e);
  }
  r.push_back({id("mask-resize-horizontal.glsl"), "royale-mask-h", {}, {}, rck::launch_royale_mask_h, setupMaskH, true});
  r.push_back({id("scanlines-horizontal-apply-mask.glsl"), "royale-scanlines-h", {}, {"PassPrev6Texture", "PassPrev3Texture"},
               rck::launch_royale_scan_h, setupScanH, false});
  // crt-royale-fake-bloom.glslp: the two files that differ from crt-royale's by `#define PHOSPHOR_BLOOM_FAKE`
  r.push_back({id("bloom-approx-fake-bloom.glsl"), "royale-bloom-approx", {}, {"PassPrev2Texture"}, rck::launch_royale_bloom_approx, setupBloomApprox, false});
  r.push_back({id("scanlines-horizontal-apply-mask-fake-bloom.glsl"), "royale-scanlines-h-fake-bloom", {},
               {"PassPrev6Texture", "PassPrev5Texture", "PassPrev3Texture"}, rck::launch_royale_scan_h_fake, setupScanHFake, false});
  r.push_back({id("brightpass.glsl"), "royale-brightpass", {}, {"PassPrev4Texture"}, rck::launch_royale_brightpass, setupBrightpass, false});
  r.push_back({id("bloom-vertical.glsl"), "royale-bloom-v", {}, {}, rck::launch_royale_bloom_v, setupBloomV, false});
  r.push_back({id("bloom-horizontal-reconstitute.glsl"), "royale-bloom-h", {},
               {"PassPrev3Texture", "PassPrev2Texture", "PassPrev6Texture"}, rck::launch_royale_bloom_h, setupBloomH, false});
  r.push_back({id("geometry-aa-last-pass.glsl"), "royale-last",
               {{"crt_gamma", 2.5f, 1.0f, 5.0f, 0.025f, "Simulated CRT Gamma"},
                {"lcd_gamma", 2.2f, 1.0f, 5.0f, 0.025f, "Your Display Gamma"},
                {"levels_contrast", 1.0f, 0.0f, 4.0f, 0.015625f, "Contrast"},
                {"halation_weight", 0.0f, 0.0f, 1.0f, 0.005f, "Halation Weight"},
                {"diffusion_weight", 0.075f, 0.0f, 1.0f, 0.005f, "Diffusion Weight"},
                {"bloom_underestimate_levels", 0.8f, 0.0f, 5.0f, 0.01f, "Bloom - Underestimate Levels"},
                {"bloom_excess", 0.0f, 0.0f, 1.0f, 0.005f, "Bloom - Excess"},
                {"beam_min_sigma", 0.02f, 0.005f, 1.0f, 0.005f, "Beam - Min Sigma"},
                {"beam_max_sigma", 0.3f, 0.005f, 1.0f, 0.005f, "Beam - Max Sigma"},
                {"beam_spot_power", 0.33f, 0.01f, 16.0f, 0.01f, "Beam - Spot Power"},
                {"beam_min_shape", 2.0f, 2.0f, 32.0f, 0.1f, "Beam - Min Shape"},
                {"beam_max_shape", 4.0f, 2.0f, 32.0f, 0.1f, "Beam - Max Shape"},
                {"beam_shape_power", 0.25f, 0.01f, 16.0f, 0.01f, "Beam - Shape Power"},
                {"beam_horiz_filter", 0.0f, 0.0f, 2.0f, 1.0f, "Beam - Horiz Filter"},
                {"beam_horiz_sigma", 0.35f, 0.0f, 0.67f, 0.005f, "Beam - Horiz Sigma"},
                {"beam_horiz_linear_rgb_weight", 1.0f, 0.0f, 1.0f, 0.01f, "Beam - Horiz Linear RGB Weight"},
                {"convergence_offset_x_r", 0.0f, -4.0f, 4.0f, 0.05f, "Convergence - Offset X Red"},
                {"convergence_offset_x_g", 0.0f, -4.0f, 4.0f, 0.05f, "Convergence - Offset X Green"},
                {"convergence_offset_x_b", 0.0f, -4.0f, 4.0f, 0.05f, "Convergence - Offset X Blue"},
                {"convergence_offset_y_r", 0.0f, -2.0f, 2.0f, 0.05f, "Convergence - Offset Y Red"},
                {"convergence_offset_y_g", 0.0f, -2.0f, 2.0f, 0.05f, "Convergence - Offset Y Green"},
                {"convergence_offset_y_b", 0.0f, -2.0f, 2.0f, 0.05f, "Convergence - Offset Y Blue"},
                {"mask_type", 1.0f, 0.0f, 2.0f, 1.0f, "Mask - Type"},
                {"mask_sample_mode_desired", 0.0f, 0.0f, 2.0f, 1.0f, "Mask - Sample Mode"},
                {"mask_specify_num_triads", 0.0f, 0.0f, 1.0f, 1.0f, "Mask - Specify Number of Triads"},
                {"mask_triad_size_desired", 3.0f, 1.0f, 18.0f, 0.125f, "Mask - Triad Size Desired"},
                {"mask_num_triads_desired", 480.0f, 342.0f, 1920.0f, 1.0f, "Mask - Number of Triads Desired"},
                {"aa_subpixel_r_offset_y_runtime", 0.0f, -0.333333333f, 0.333333333f, 0.333333333f, "AA - Subpixel R Offset Y"},
                {"aa_cubic_c", 0.5f, 0.0f, 4.0f, 0.015625f, "AA - Cubic Sharpness"},
                {"aa_gauss_sigma", 0.5f, 0.0625f, 1.0f, 0.015625f, "AA - Gaussian Sigma"},
                {"geom_mode_runtime", 0.0f, 0.0f, 3.0f, 1.0f, "Geometry - Mode"},
                {"geom_radius", 2.0f, 0.16f, 1024.0f, 0.1f, "Geometry - Radius"},
                {"geom_view_dist", 2.0f, 0.5f, 1024.0f, 0.25f, "Geometry - View Distance"},
                {"geom_tilt_angle_x", 0.0f, -3.14159265f, 3.14159265f, 0.017453292519943295f, "Geometry - Tilt Angle X"},
                {"geom_tilt_angle_y", 0.0f, -3.14159265f, 3.14159265f, 0.017453292519943295f, "Geometry - Tilt Angle Y"},
                {"geom_aspect_ratio_x", 432.0f, 1.0f, 512.0f, 1.0f, "Geometry - Aspect Ratio X"},
                {"geom_aspect_ratio_y", 329.0f, 1.0f, 512.0f, 1.0f, "Geometry - Aspect Ratio Y"},
                {"geom_overscan_x", 1.0f, 0.00390625f, 4.0f, 0.00390625f, "Geometry - Overscan X"},
                {"geom_overscan_y", 1.0f, 0.00390625f, 4.0f, 0.00390625f, "Geometry - Overscan Y"},
                {"border_size", 0.015f, 0.0000001f, 0.5f, 0.005f, "Border - Size"},
                {"border_darkness", 2.0f, 0.0f, 16.0f, 0.0625f, "Border - Darkness"},
                {"border_compress", 2.5f, 1.0f, 64.0f, 0.0625f, "Border - Compression"},
                {"interlace_bff", 0.0f, 0.0f, 1.0f, 1.0f, "Interlacing - Bottom Field First"},
                {"interlace_1080i", 0.0f, 0.0f, 1.0f, 1.0f, "Interlace - Detect 1080i"}},
               {}, rck::launch_royale_last, setupLast, false});
  r.back().mip_aware = true;   // mipmap_input (crt-royale-fake-bloom): LOD from the pixel quad, chain built by the engine
  for (auto& e : r) {
    const std::string n = e.name;
    if (n == "royale-bloom-approx" || n == "royale-mask-v") e.reads_input = false;
    // pass 0 at 1:1 is a byte map: its two consumers read the source frame through the composed decode table instead
    if (n == "royale-first") e.byte_map = rck::royale_first_byte_map;
    if (n == "royale-scanlines-v") e.decode_table_inputs = 1u;        // Texture
    if (n == "royale-bloom-approx") e.decode_table_inputs = 1u << 1;  // PassPrev2Texture
  }
}

}  // namespace rc
