// crt-royale pass 1 (scanlines-vertical-interlacing.glsl): the general per-pixel kernels, and the table form
// that runs the shipped 1:1 configuration.
//
// Per pixel the shader evaluates nine generalized-Gaussian beam profiles (3 scanlines x 3 channels), each
// a function K(colour, distance) of ~280 float operations (5 log2, 9 exp2, 4 divisions of the GL's
// polynomial / IEEE kind), sums three per channel, halves, and stores to an sRGB8 target.  At 1:1 geometry
// (source lines = target lines, progressive) the structure is far simpler than the arithmetic:
//   * the three scanlines of target row y are source rows y, y+1, y-1, sampled at texel centres up to float
//     rounding: the bilinear weights are 0 or a few 2^-13 (or 1 minus that), `dist` is 0 or +-2^-14 at most;
//   * so every colour that reaches K is a decoded sRGB8 byte plus a perturbation delta of at most 2.6e-4,
//     and every distance is one of nine constants plus `dist`;
//   * the pass output is a BYTE: encode(((K0 + K1) + K2) * 0.5).
// k_royale_scan_v_tab therefore evaluates K from a table: the exact float value T of K at the node (computed by
// the very beam_k code of the general kernel, on the device), corrected by a second-order expansion in delta and
// a first-order one in dist whose coefficients come from the closed-form K in double precision, together with a
// bound on everything the expansion leaves out.  That bound is MEASURED EXHAUSTIVELY, not sampled: when the tables of a
// geometry are built, k_scan_tab_bounds evaluates the exact float K (the general kernel's beam_k) at EVERY float colour a
// node can be selected for (all 2^20 floats of a log bucket; every float within kMaxDelta of a byte's decoded value, above
// 2^-8) and every distance the geometry's rows actually produce (a handful of values: k_scan_geometry), and records the
// largest difference to the very expression the table kernel evaluates - about 2 * 10^9 exact evaluations per distance.
// If the sRGB8 byte is the same over the whole interval [S - B, S + B] the byte is certain and is stored;
// otherwise (about 0.6 % of the pixels on uniform noise) the pixel goes to its wave's list and from there to the pass's fix
// list, which k_royale_scan_v_fix renders with the general form.  The result is bit-identical to the general kernel whenever
// the bound holds - which it does by construction for every input the geometry can produce; tests/test_royale_fullsize.py
// compares both forms on full-size noise and natural-like frames, and tests/test_royale_scan_table.py checks the
// measured bound against the oracle's exact K (every float of a node's range for some nodes, random ones for all).
// Colours below 2^-8 (where K is strongly non-linear in colour) use log-spaced nodes, 8 per octave, instead of
// the byte nodes, so the relative perturbation stays below 4.6 %.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../rc_log.h"
#include "royale_common.h"
#include "royale_strip2.h"

using namespace rcd;
using namespace rcroyale;

namespace {

// ------------------------------------------------------------------------------- P1 ------
// scanlines-vertical-interlacing.glsl FS 5982-6141; beam functions 4775-4998; gamma_impl 3907.
// The nine (scanline, channel) beam evaluations of a pixel are independent and identical, 250
// float operations each: they run as four packed pairs plus one scalar (rc_vecmath.h).
template <class F, bool SAFE>
__device__ __forceinline__ F div_sel_(F n, F d) { return SAFE ? div_safe_v<F>(n, d) : n / d; }

template <class F, bool SAFE>
__device__ __forceinline__ F gamma_impl1(F s, F s_inv) {
  const float g = 1.12906830989f, c0 = 0.8109119309638332633713423362694399653724431f;
  const float c1 = 0.4808354605142681877121661197951496120000040f, e = 2.71828182845904523536028747135266249775724709f;
  const F sph = s + 0.5f;
  const F lanczos_sum = c0 + div_sel_<F, SAFE>(F(c1), s + 1.0f);  // s + 1 in [1.25, 1.5]
  // base is in [0.69, 0.78] for s = 1/beta in [1/4, 1/2]: a positive normal, no log2 edge cases
  // (s + 0.5 + g) / e: the GL folds the two additive constants, (s + (0.5 + g)) / e (oracle gamma_impl1)
  const F base = div_const_v<F>(s + (0.5f + g), e, 1.0f / e);
  return (exp2_v<F, true>(log2_core_v<F>(base) * sph) * lanczos_sum) * s_inv;  // finite argument
}

// One scanline's contribution to one channel: scanline_contrib(dist, color, ...) of the GLSL, with
// the three sub-pixel samples at dist, dist + off, |dist - off|.
template <class F, bool SAFE>
__device__ __forceinline__ F beam_k(F color, F dist, float off, float sigma_range, float shape_range) {
  // SAFE also means: color is a non-negative number, so no exp2 argument below can be a NaN (log2 of
  // 0 is -inf, every product with it stays -inf) and exp2's two clamps fold into one v_med3_f32
  const F lg = log2_v(color);  // pow(color, p) = exp2(log2(color) * p) for both exponents
  const F sigma = 0.02f + sigma_range * exp2_v<F, SAFE>(lg * (1.0f / 3.0f));
  const F alpha = 1.41421356237309504880f * sigma;  // sqrtf(2.0f)
  const F beta = 2.0f + shape_range * exp2_v<F, SAFE>(lg * (1.0f / 4.0f));
  // SAFE (colour sampled from an 8-bit texture): operand ranges for div_safe_: alpha in [0.028, 0.43],
  // beta in [2, 4], gamma_impl1 in [0.88, 3.7]; color is 0 or >= 2^-40, so the numerator is 0 or >= 2^-41
  const F alpha_inv = div_sel_<F, SAFE>(F(1.0f), alpha);
  const F beta_inv = div_sel_<F, SAFE>(F(1.0f), beta);
  const F scale = div_sel_<F, SAFE>(color * beta * 0.5f * alpha_inv, gamma_impl1<F, SAFE>(beta_inv, beta));
  const F scale3 = div_const_v<F>(scale, 3.0f, 1.0f / 3.0f);
  const F d2 = dist + off, d3 = abs_v(dist - off);
  // pow(a, beta) with a >= 0 and beta in [2, 4]: for a zero or denormal `a` the full log2 returns
  // -inf and the core returns a value <= -126; times beta both are below exp2's clamp and give
  // exactly 0, so the edge-case selects of log2 are not needed here.
  const F w1 = exp_v<F, SAFE>(-exp2_v<F, SAFE>(log2_core_v<F>(abs_v(dist * alpha_inv)) * beta));
  const F w2 = exp_v<F, SAFE>(-exp2_v<F, SAFE>(log2_core_v<F>(abs_v(d2 * alpha_inv)) * beta));
  const F w3 = exp_v<F, SAFE>(-exp2_v<F, SAFE>(log2_core_v<F>(abs_v(d3 * alpha_inv)) * beta));
  return scale3 * (w1 + w2 + w3);
}

template <class SI, class SO>
__global__ void __launch_bounds__(256, 4) k_royale_scan_v(const PassLaunch L) {
  RC_SRGB_LDS_OF(lds, L, L.in);
  RC_TILE_LOOP_BEGIN
  const float tsx = (float)L.in.w, tsy = L.params[RP1_TSY];   // TextureSize.y as the reference sets it (royale_setup.cpp)
  const float y_step = L.params[RP1_Y_STEP], uv_step_y = L.params[RP1_UV_STEP_Y], ph = L.params[RP1_PH];
  const float tix = 1.0f / tsx, tiy = 1.0f / tsy;
  const float sigma_range = maxps(0.3f, 0.02f) - 0.02f, shape_range = maxps(4.0f, 2.0f) - 2.0f;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  // get_last_scanline_uv
  const float frame_count = (float)(L.frame_count0 + z);
  const float field_offset = __builtin_floorf(y_step * 0.75f) * mod_glsl(frame_count + 0.0f, 2.0f);
  const float ctx = u * tsx, cty = v * tsy;
  const float ptx = __builtin_floorf(ctx - kUnderHalf), pty = __builtin_floorf(cty - kUnderHalf);
  const float wrong_field = mod_glsl(pty + field_offset, y_step);
  const float stx = (ptx - 0.0f) + 0.5f, sty = (pty - wrong_field) + 0.5f;
  const float su = stx * tix, sv = sty * tiy;
  const float dist = (cty - sty) / y_step;
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 s2 = SI::get(L.in, img, su, sv, &lds);
  const float4 s3 = SI::get(L.in, img, su + 0.0f, sv + uv_step_y, &lds);
  const float dist_round = __builtin_rintf(dist);
  const float off_x = mix_rt(-0.0f, 2.0f * 0.0f, dist_round);
  const float off_y = mix_rt(-uv_step_y, 2.0f * uv_step_y, dist_round);
  const float4 so = SI::get(L.in, img, su + off_x, sv + off_y, &lds);
  const float off = ph / 3.0f;
  const float conv_y[3] = {0.2f, 0.4f, 0.6f};
  // colour and distance of the nine evaluations, index = scanline * 3 + channel
  float col[9] = {s2.x, s2.y, s2.z, s3.x, s3.y, s3.z, so.x, so.y, so.z}, dd[9], kk[9];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    dd[ch] = dist - conv_y[ch];
    // additive constants re-associated as the GL's compiler does: 1-(dist-c) -> (1+c)-dist, ...
    dd[3 + ch] = __builtin_fabsf((1.0f + conv_y[ch]) - dist);
    dd[6 + ch] = mix_rt(dist + (1.0f - conv_y[ch]), (2.0f + conv_y[ch]) - dist, dist_round);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const v2f k = beam_k<v2f, SI::kUnitRange>(v2f{col[2 * j], col[2 * j + 1]}, v2f{dd[2 * j], dd[2 * j + 1]}, off, sigma_range, shape_range);
    kk[2 * j] = k.x;
    kk[2 * j + 1] = k.y;
  }
  kk[8] = beam_k<float, SI::kUnitRange>(col[8], dd[8], off, sigma_range, shape_range);
  float out[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) out[ch] = ((kk[ch] + kk[3 + ch]) + kk[6 + ch]) * 0.5f;
  SO::put(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

// One pixel's inputs to the nine beam evaluations: colours and distances, index = scanline * 3 + channel.
template <class SI>
__device__ __forceinline__ void scan_v_gather(const PassLaunch& L, const SrgbLds& lds, int x, int y, int z, float* col, float* dd) {
  const bool lo = rcd::lower_tri(x, y, L.out_w, L.out_h);
  const float tsx = (float)L.in.w, tsy = L.params[RP1_TSY];
  const float y_step = L.params[RP1_Y_STEP], uv_step_y = L.params[RP1_UV_STEP_Y];
  const float tix = 1.0f / tsx, tiy = 1.0f / tsy;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float frame_count = (float)(L.frame_count0 + z);
  const float field_offset = __builtin_floorf(y_step * 0.75f) * mod_glsl(frame_count + 0.0f, 2.0f);
  const float ctx = u * tsx, cty = v * tsy;
  const float ptx = __builtin_floorf(ctx - kUnderHalf), pty = __builtin_floorf(cty - kUnderHalf);
  const float wrong_field = mod_glsl(pty + field_offset, y_step);
  const float stx = (ptx - 0.0f) + 0.5f, sty = (pty - wrong_field) + 0.5f;
  const float su = stx * tix, sv = sty * tiy;
  const float dist = (cty - sty) / y_step;
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 s2 = SI::get(L.in, img, su, sv, &lds);
  const float4 s3 = SI::get(L.in, img, su + 0.0f, sv + uv_step_y, &lds);
  const float dist_round = __builtin_rintf(dist);
  const float off_x = mix_rt(-0.0f, 2.0f * 0.0f, dist_round);
  const float off_y = mix_rt(-uv_step_y, 2.0f * uv_step_y, dist_round);
  const float4 so = SI::get(L.in, img, su + off_x, sv + off_y, &lds);
  const float conv_y[3] = {0.2f, 0.4f, 0.6f};
  col[0] = s2.x; col[1] = s2.y; col[2] = s2.z; col[3] = s3.x; col[4] = s3.y; col[5] = s3.z; col[6] = so.x; col[7] = so.y; col[8] = so.z;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    dd[ch] = dist - conv_y[ch];
    dd[3 + ch] = __builtin_fabsf((1.0f + conv_y[ch]) - dist);
    dd[6 + ch] = mix_rt(dist + (1.0f - conv_y[ch]), (2.0f + conv_y[ch]) - dist, dist_round);
  }
}

// Two vertically adjacent pixels per thread: the nine evaluations of one pixel leave one scalar evaluation next to
// four packed pairs; pairing evaluation j of the upper pixel with evaluation j of the lower one makes all nine packed
// (rc_vecmath.h).  Tiles are 64 x 8; a wave still stores 256 contiguous bytes per row.  Same results as k_royale_scan_v.
template <class SI, class SO>
__global__ void __launch_bounds__(256, 4) k_royale_scan_v2(const PassLaunch L) {
  RC_SRGB_LDS_OF(lds, L, L.in);
  const int tiles_x = (L.out_w + 63) >> 6, tiles_y = (L.out_h + 7) >> 3;
  const int tiles_per_frame = tiles_x * tiles_y, n_tiles = tiles_per_frame * L.n_frames;
  const float sigma_range = maxps(0.3f, 0.02f) - 0.02f, shape_range = maxps(4.0f, 2.0f) - 2.0f;
  const float off = L.params[RP1_PH] / 3.0f;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int z = tile / tiles_per_frame, rem = tile - z * tiles_per_frame;
    const int ty = rem / tiles_x;
    const int x = (rem - ty * tiles_x) * 64 + (int)threadIdx.x, y0 = ty * 8 + (int)threadIdx.y * 2;
    if (x >= L.out_w || y0 >= L.out_h) continue;
    const bool two = y0 + 1 < L.out_h;
    float ca[9], da[9], cb[9], db[9];
    scan_v_gather<SI>(L, lds, x, y0, z, ca, da);
    scan_v_gather<SI>(L, lds, x, two ? y0 + 1 : y0, z, cb, db);
    float ka[9], kb[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const v2f k = beam_k<v2f, SI::kUnitRange>(v2f{ca[j], cb[j]}, v2f{da[j], db[j]}, off, sigma_range, shape_range);
      ka[j] = k.x;
      kb[j] = k.y;
    }
    SO::put(L, z, x, y0, make_float4(((ka[0] + ka[3]) + ka[6]) * 0.5f, ((ka[1] + ka[4]) + ka[7]) * 0.5f, ((ka[2] + ka[5]) + ka[8]) * 0.5f, 1.0f), &lds);
    if (two)
      SO::put(L, z, x, y0 + 1, make_float4(((kb[0] + kb[3]) + kb[6]) * 0.5f, ((kb[1] + kb[4]) + kb[7]) * 0.5f, ((kb[2] + kb[5]) + kb[8]) * 0.5f, 1.0f), &lds);
  }
}

// ------------------------------------------------------------------------ table form ------
constexpr int kLogNodes = 192;               // colours in [2^-32, 2^-8): 8 nodes per octave (top 3 mantissa bits), node = bucket midpoint
constexpr int kNodes = kLogNodes + 256;      // then one node per sRGB8 byte; node kLogNodes + 0 is the zero colour
constexpr uint32_t kLogBits0 = 0x2f800000u;  // 2^-32
constexpr float kLogMax = 0.00390625f;       // 2^-8
constexpr float kMaxDelta = 2.6e-4f;         // byte nodes: largest |colour - node| the bounds are computed for
constexpr float kMaxWeight = 1.25e-4f;       // largest off-centre bilinear weight per axis (2 * kMaxWeight * 1.0 < kMaxDelta)
constexpr float kMaxDist = 6.103515625e-05f; // 2^-14: largest |dist| the bounds are computed for
#ifndef RC_SCAN_STRIP_ROWS
#define RC_SCAN_STRIP_ROWS 32   // (8 -> 32: 12.8 -> 12.0 us per 1080p frame: a strip decodes 4 + rows source rows)
#endif
constexpr int kStripRows = RC_SCAN_STRIP_ROWS;   // target rows one thread walks (at most 32: a lane's uncertain rows are a bit mask)
static_assert(kStripRows % 4 == 0 && kStripRows <= 32, "kStripRows");
constexpr int kTabWaves = 16;                // 1024 threads: one workgroup per CU (the tables fill its LDS)
constexpr int kTabThreads = kTabWaves * 64;

// dynamic LDS layout of the table kernel, in bytes (the kernel has no static LDS: absolute offsets, see royale_strip2.h).  The
// expansion table comes first, so that a node's record is addressed by (byte or bucket) * 16 plus an immediate for the role.
constexpr uint32_t kLdsA = 0u;                                        // float4 A[9][kNodes]: T, dK/dc, d2K/dc2 / 2, W
constexpr uint32_t kLdsDec = kLdsA + 9u * kNodes * 16u;               // the sRGB decode table (256 floats)
constexpr uint32_t kLdsEnc2 = kLdsDec + 1024u;                        // second form of the sRGB8 encode table (rc_device.h)
constexpr int kWaveList = 256;                                        // a wave's uncertain pixels, collected in LDS until it flushes them
constexpr uint32_t kLdsFail = (kLdsEnc2 + kSrgb2Runs * 4u + 15u) & ~15u;   // uint32 list[kTabWaves][kWaveList]
constexpr uint32_t kLdsTotalBytes = kLdsFail + (uint32_t)kTabWaves * kWaveList * 4u;
static_assert(kLdsDec + 1020u < 65536u && 8u * kNodes * 16u + (kNodes - 1u) * 16u < 65536u, "immediate offsets of the LDS reads");
// W of a record: dK/ddist with its low 11 mantissa bits replaced by the node's bound, a 5-bit exponent and a 6-bit mantissa
// rounded up: (1 + m / 64) 2^(e - 40).  (The coefficient's truncation is part of what the bound is measured against.)
__host__ __device__ __forceinline__ float scan_w_slope(float w) { return bits2f(f2bits(w) & 0xfffff800u); }
__host__ __device__ __forceinline__ float scan_w_bound(float w) { return bits2f(((f2bits(w) & 0x7ffu) << 17) + 0x2b800000u); }
// ... and half of it: the table kernel keeps HALVED records in LDS (the pass stores (K0 + K1 + K2) / 2: halving every coefficient
// is exact in binary floating point and commutes with every rounding, so the sum of the halved evaluations is the halved sum bit
// for bit, and the two multiplications by 0.5 per channel go)
__device__ __forceinline__ float scan_w_bound_half(float w) { return bits2f(((f2bits(w) & 0x7ffu) << 17) + 0x2b000000u); }
__host__ __device__ inline uint32_t scan_bound_code(float b) {   // smallest code whose value is >= b
  const uint32_t lo = 0x2b800000u, u = f2bits(b);
  if (!(b > bits2f(lo))) return 0u;
  const uint32_t code = (u - lo + 0x1ffffu) >> 17;   // round the 23-bit mantissa up to 6 bits
  return code > 0x7ffu ? 0x7ffu : code;
}
constexpr int kMaxDists = 24;                                 // distinct row distances a geometry may have for the table form

// Pixels whose byte the table form could not certify are collected per wave in LDS and appended (one global atomic
// per flush of a wave's list) to a list in the pass's scratch buffer (PassLaunch::scratch: a counter, then entries
// (frame * H + y) * W + x); k_royale_scan_v_fix renders them with the general form.
constexpr uint32_t kFixHeader = 256;  // bytes reserved for the counter (the registry sizes the scratch: header + 4 bytes per pixel, per frame)
__device__ __forceinline__ uint32_t* fix_counter(const PassLaunch& L) { return static_cast<uint32_t*>(L.scratch); }
__device__ __forceinline__ uint32_t* fix_list(const PassLaunch& L) { return reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.scratch) + kFixHeader); }

struct ScanRow {  // (32 bytes: the table kernel reads it as two float4) per target row, the same for every pixel of the row except `dist` (one value per triangle)
  float dist_lo, dist_up;
  float wy[3];     // bilinear weight between the two source rows of scanline s2 / s3 / so
  uint32_t up;     // bit j: that pair starts one row above the scanline's own row (weight 1 - tiny);
                   // bit 8 / 9: |dist_lo| / |dist_up| is beyond kMaxDist: those pixels take the exact path
  uint32_t pad[2];
};

__device__ __forceinline__ float scan_dd(int j, int ch, float dist, float dist_round) {
  const float conv_y[3] = {0.2f, 0.4f, 0.6f};
  if (j == 0) return dist - conv_y[ch];
  if (j == 1) return __builtin_fabsf((1.0f + conv_y[ch]) - dist);
  return mix_rt(dist + (1.0f - conv_y[ch]), (2.0f + conv_y[ch]) - dist, dist_round);
}

// T of every node - the general kernel's own beam_k at the node colour and the role's distance for dist = 0 - and with it the
// record's polynomial in the colour ITSELF (not in colour - node: the table kernel then needs neither the node value nor the
// subtraction): K(c) ~ a0 + c (a1 + c a2) with a2 = K''/2, a1 = K' - 2 a2 node, a0 = T - K' node + a2 node^2, from the
// host's K', K''/2 (A[i].y, .z on entry).  Rounding these to float is part of what the bound is measured against.
__global__ void __launch_bounds__(256) k_scan_tab_nodes(float4* A, const float* __restrict__ node, float off) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * kNodes) return;
  const int jc = i / kNodes;
  const float sigma_range = maxps(0.3f, 0.02f) - 0.02f, shape_range = maxps(4.0f, 2.0f) - 2.0f;
  const float c = node[i - jc * kNodes];
  const float T = beam_k<float, true>(c, scan_dd(jc / 3, jc % 3, 0.0f, 0.0f), off, sigma_range, shape_range);
  const double n = (double)c, d1 = (double)A[i].y, d2 = (double)A[i].z;
  A[i].x = (float)((double)T - d1 * n + d2 * n * n);
  A[i].y = (float)(d1 - 2.0 * d2 * n);
}

// The bound of every node: the largest difference, over EVERY float colour the table kernel can select the node for and
// every row distance of the geometry (dists[0 .. n_dists)), between the exact K (beam_k, the general kernel's code) and the
// expansion exactly as k_royale_scan_v_tab evaluates it.  blockIdx.y = role * kNodes + node; the node's colours are split
// over blockIdx.x.  bound[] must be zero on entry; non-negative floats order like their bit patterns.
__device__ __forceinline__ void scan_node_range(int n, float c0, uint32_t* lo, uint32_t* hi) {
  if (n < kLogNodes) {   // a log bucket: 2^20 consecutive floats
    *lo = kLogBits0 + ((uint32_t)n << 20);
    *hi = *lo + (1u << 20) - 1u;
  } else {               // a byte: every float within kMaxDelta, not below 2^-8 (smaller colours select a log node), not above 1
    *lo = f2bits(fmaxf(c0 - kMaxDelta, kLogMax));
    *hi = f2bits(fminf(c0 + kMaxDelta, 1.0f));
  }
}
__global__ void __launch_bounds__(256) k_scan_tab_bounds(const float4* __restrict__ A, const float* __restrict__ node, const float* __restrict__ dists,
                                                        int n_dists, float off, float* bound, float* kmax) {
  const int e = (int)blockIdx.y, jc = e / kNodes, n = e - jc * kNodes;
  const float c0 = node[n];
  if (n == kLogNodes) return;                                   // the zero colour: bound set by the host (no expansion, K < 2.5e-8)
  if (n > kLogNodes && !(c0 + kMaxDelta >= kLogMax)) return;   // a byte below the log range: never selected
  uint32_t lo, hi;
  scan_node_range(n, c0, &lo, &hi);
  const float sigma_range = maxps(0.3f, 0.02f) - 0.02f, shape_range = maxps(4.0f, 2.0f) - 2.0f;
  const float4 a = A[e];
  float worst = 0.0f, largest = 0.0f;   // ... and the largest exact K itself, per role (kmax[9]: see kScanSkipBelow)
  // two consecutive colours per step through the packed form of beam_k (the same IEEE operations per component)
  for (uint32_t i = lo + 2u * (blockIdx.x * 256u + threadIdx.x); i <= hi; i += 2u * gridDim.x * 256u) {
    const v2f c = {bits2f(i), bits2f(i + 1u <= hi ? i + 1u : i)};
    const float px = fma_(c.x, a.z, a.y), py = fma_(c.y, a.z, a.y);
    for (int d = 0; d < n_dists; ++d) {
      const float dist = dists[d], dd = scan_dd(jc / 3, jc % 3, dist, 0.0f);
      const v2f k = beam_k<v2f, true>(c, v2f{dd, dd}, off, sigma_range, shape_range);
      // the kernel: fma(c, fma(c, a.z, a.y), fma(W, dist, a.x)), W = the distance slope with the bound's code in its low 11
      // bits - not known yet: the result is monotone in W, so the worse of the two extreme codes covers whichever is stored
      float ek = 0.0f;
#pragma unroll
      for (int end = 0; end < 2; ++end) {
        const float base = fma_(bits2f((f2bits(a.w) & 0xfffff800u) | (end ? 0x7ffu : 0u)), dist, a.x);
        const float kx = fma_(c.x, px, base), ky = fma_(c.y, py, base);
        const double ex = fabs((double)k.x - (double)kx), ey = fabs((double)k.y - (double)ky);
        ek = fmaxf(ek, __double2float_ru(ex > ey ? ex : ey));
      }
      worst = fmaxf(worst, ek);
      largest = fmaxf(largest, fmaxf(__builtin_fabsf(k.x), __builtin_fabsf(k.y)));
    }
  }
  if (worst > 0.0f) atomicMax(reinterpret_cast<uint32_t*>(&bound[e]), f2bits(worst));
  if (largest > 0.0f) atomicMax(reinterpret_cast<uint32_t*>(&kmax[jc]), f2bits(largest));
}
// the stored bound absorbs the roundings of the table kernel's own sums of the three K and of the three bounds; it goes into
// the low bits of the record's W, rounded up to its 11-bit code (bound[] keeps the value that code stands for)
__global__ void __launch_bounds__(256) k_scan_tab_bounds_finish(float4* A, float* bound, const float* __restrict__ node) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 9 * kNodes) return;
  const int n = e % kNodes;
  float b = bound[e] * 1.000001f + 1e-12f;
  if (n == kLogNodes) b = 2.5e-8f;   // colours below 2^-32 (and 0): 0 <= K <= 81 colour < 1.9e-8 (beta <= 4, 1/alpha <= 35.4, gamma_impl >= 0.88)
  uint32_t code = scan_bound_code(b);
  if (n > kLogNodes && !(node[n] + kMaxDelta >= kLogMax)) code = 0x7ffu;   // a byte below the log range: never selected
  const float w = bits2f((f2bits(A[e].w) & 0xfffff800u) | code);
  A[e].w = w;
  bound[e] = scan_w_bound(w);
}

// Where the samples of every target row / column land, evaluated with the operations of scan_v_gather and of the
// LINEAR clamp-to-edge sampler; *bad is set if the geometry is not the regular 1:1 one the table form assumes.
__global__ void __launch_bounds__(256) k_scan_geometry(const PassLaunch L, ScanRow* rows, float* cols, uint32_t* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float tsx = (float)L.in.w, tsy = L.params[RP1_TSY], uv_step_y = L.params[RP1_UV_STEP_Y];
  const float tix = 1.0f / tsx, tiy = 1.0f / tsy;
  uint32_t why = 0u;  // reasons the geometry is not the regular one (bit set), for diagnostics
  if (i < L.out_h) {
    ScanRow r = {};
    if (i >= 2 && i < L.out_h - 2) {
      float sv = 0.f;
      for (int side = 0; side < 2; ++side) {
        const float v = vary(L.plane[1], 0, i, side == 0);
        const float cty = v * tsy, pty = __builtin_floorf(cty - kUnderHalf);
        const float sty = (pty - 0.0f) + 0.5f;   // progressive: wrong_field = 0
        const float dist = (cty - sty) / 1.0f;
        if (!(pty == (float)i)) why |= 2u;
        if (!(__builtin_rintf(dist) == 0.0f)) why |= 4u;
        // a row whose dist exceeds the range of the bounds (2 of 1080 rows at 1080p, one triangle each) takes the exact path
        if (!(__builtin_fabsf(dist) <= kMaxDist)) r.up |= 256u << side;
        (side == 0 ? r.dist_lo : r.dist_up) = dist;
        sv = sty * tiy;
      }
      const float svr[3] = {sv, sv + uv_step_y, sv + mix_rt(-uv_step_y, 2.0f * uv_step_y, 0.0f)};
      const int roff[3] = {0, 1, -1};
      for (int j = 0; j < 3; ++j) {
        const float w = linear_coord_edge_pair(svr[j], L.in.h);
        const float y0 = __builtin_floorf(w);
        r.wy[j] = w - y0;
        const float want = (float)(i + roff[j]);
        // the pair's weight towards the scanline's own row is 1 up to kMaxWeight, so that |colour - node| <= kMaxDelta
        if (y0 == want - 1.0f) {
          r.up |= 1u << j;
          if (!(1.0f - r.wy[j] <= kMaxWeight)) why |= 8u;
        } else {
          if (!(y0 == want)) why |= 16u;
          if (!(r.wy[j] <= kMaxWeight)) why |= 32u;
        }
      }
    }
    rows[i] = r;
  }
  if (i < L.out_w) {
    float wx = 0.f;
    for (int side = 0; side < 2; ++side) {
      const float u = vary(L.plane[0], i, 0, side == 0);
      const float ctx = u * tsx, ptx = __builtin_floorf(ctx - kUnderHalf);
      const float su = ((ptx - 0.0f) + 0.5f) * tix;
      const float w = linear_coord_edge_pair(su, L.in.w);
      const float x0 = __builtin_floorf(w);
      wx = w - x0;
      if (!(ptx == (float)i && x0 == (float)i)) why |= 64u;
      if (!(wx >= 0.0f && wx <= kMaxWeight)) why |= 128u;
    }
    cols[i] = wx;
  }
  if (why) atomicOr(bad, why);
}

// Is the sRGB8 byte the same for every value in [s - b, s + b]?  Returns the byte; *ok tells whether it is certain.
// With the second form of the encode table (one entry per RSQRTPS run from the first float that stores a non-zero byte, the
// linear segment included): both ends are clamped into the table's range (everything below stores 0 like its first float,
// everything above 1 stores 255 like 1) and looked up; inside one run the byte is monotone, and across the boundary to
// the next run it is where bit 30 of the entry says so (the encode is not monotone across every run boundary).  So the byte
// is certain when both ends give the same byte and lie in one run, or in two neighbouring runs with a monotone boundary.
__device__ __forceinline__ uint32_t srgb8_interval(float s, float b, bool* ok) {
  const uint32_t bl = f2bits(__builtin_amdgcn_fmed3f(s - b, bits2f(kSrgb2MinBits), 1.0f)), bh = f2bits(__builtin_amdgcn_fmed3f(s + b, bits2f(kSrgb2MinBits), 1.0f));
  const uint32_t el = rcstrip2::lds_u32(((bl >> 13) << 2) + (kLdsEnc2 - (kSrgb2Run0 << 2)));
  const uint32_t tl = el + (bl & 0x1fffu);
  // both ends in one run (nearly always: a run is 8192 floats wide): the upper end's position follows from the lower one's,
  // and "same run, same byte" is one comparison - the two positions and the two bit patterns agree above bit 12
  const uint32_t th = tl + (bh - bl);
  bool certain = ((tl ^ th) | (bl ^ bh)) < 8192u;
  if (__builtin_amdgcn_ballot_w64((bl ^ bh) >= 8192u) != 0ull) {   // (wave-uniform) some lane's interval straddles two runs
    const uint32_t rl = bl >> 13, rh = bh >> 13;
    const uint32_t eh = rcstrip2::lds_u32((rh << 2) + (kLdsEnc2 - (kSrgb2Run0 << 2)));
    const uint32_t byte_l = (tl >> 13) & 255u, byte_h = ((eh + (bh & 0x1fffu)) >> 13) & 255u;
    certain = byte_l == byte_h && (rh - rl) <= ((el >> 30) & 1u);
  }
  *ok = certain;
  return (tl >> 13) & 255u;
}

// One (scanline, channel) evaluation from the table.  c: the sampled colour; byte_addr: (kLogNodes + the own texel's byte) * 16;
// ROLE = scanline * 3 + channel.  Returns the expanded K, adds the node's bound to *bsum.
template <int ROLE>
__device__ __forceinline__ float scan_tab_eval(float c, uint32_t byte_addr, float dist, float* bsum) {
  using namespace rcstrip2;
  const uint32_t t = f2bits(c) - kLogBits0;
  // colours in [2^-32, 2^-8) select a log bucket (8 per octave: index = t >> 20), everything else the own texel's byte node
  // (a colour below 2^-32 belongs to a zero byte: the zero node, whose polynomial is the constant 0)
  const bool dark = t < (0x3b800000u - kLogBits0);
  const uint32_t addr = dark ? ((t >> 16) & ~15u) : byte_addr;
  const v4f a = lds_v4f(kLdsA + (uint32_t)ROLE * kNodes * 16u + addr);
  *bsum += scan_w_bound_half(a.w);
  return fma_(c, fma_(c, a.z, a.y), fma_(a.w, dist, a.x));   // (a.w as it is: k_scan_tab_bounds measured it with the code bits in)
}

// SKIP1: the scanline below contributes less than skip_r / _g / _b whatever its colour (ScanNodeTables): taken as 0, bound added
template <class SI, class SO, bool SKIP1>
__global__ void __launch_bounds__(kTabThreads) k_royale_scan_v_tab(const PassLaunch L, const float4* __restrict__ gA, const ScanRow* __restrict__ rows,
                                                                  const float* __restrict__ cols, float skip_r, float skip_g, float skip_b) {
  using namespace rcstrip2;
  extern __shared__ uint32_t rc_dyn_lds_[];
  if ((uint32_t)(uintptr_t)(RC_AS3 uint32_t*)rc_dyn_lds_ != 0u) __builtin_trap();   // absolute LDS offsets
  const int tid = (int)threadIdx.x;
  // (a folded pass 0: the texels are SOURCE bytes, and a byte's node is the node of the byte pass 0 would have stored for it)
  const uint32_t* bmap = L.in.dec ? reinterpret_cast<const uint32_t*>(L.in.dec) + 256 : nullptr;
  for (int i = tid; i < 9 * kNodes; i += kTabThreads) {
    const int n = i % kNodes;
    float4 r = gA[bmap && n >= kLogNodes ? i - n + kLogNodes + (int)bmap[n - kLogNodes] : i];
    // halved (see scan_w_bound_half): a0, a1, a2 by a multiplication, W - the distance slope with the bound's code in its low bits -
    // by one step of its exponent (a slope with a zero exponent field is a denormal, which the arithmetic flushes to 0: left as it is)
    r.x *= 0.5f;
    r.y *= 0.5f;
    r.z *= 0.5f;
    if (f2bits(r.w) & 0x7f800000u) r.w = bits2f(f2bits(r.w) - 0x00800000u);
    reinterpret_cast<float4*>(rc_dyn_lds_)[i] = r;
  }
  for (int i = tid; i < 256; i += kTabThreads) rc_dyn_lds_[kLdsDec / 4 + i] = f2bits(L.in.dec ? L.in.dec[i] : k_srgb_decode[i]);   // (a folded pass 0: its composed table)
  for (int i = tid; i < (int)kSrgb2Runs; i += kTabThreads) rc_dyn_lds_[kLdsEnc2 / 4 + i] = L.srgb_enc[kSrgbRuns + i];
  __syncthreads();
  const int W = L.out_w, H = L.out_h;
  const int cgs = (W + 63) >> 6, rss = (H + kStripRows - 1) / kStripRows;
  const int strips_per_frame = cgs * rss, n_strips = strips_per_frame * L.n_frames;
  const int n_tiles = (n_strips + kTabWaves - 1) / kTabWaves;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // Pixels whose byte is not certain go to this wave's list in LDS - positions from a ballot, no atomics, no barriers - which
  // the wave appends to the pass's global fix list with one global atomic when it is about to overflow and at the end.
  uint32_t* my_list = rc_dyn_lds_ + kLdsFail / 4 + wave * kWaveList;
  uint32_t n_listed = 0u;   // wave-uniform, kept scalar (lanes beyond the frame's right edge skip the code that updates it)
  auto flush = [&]() __attribute__((always_inline)) {
    n_listed = __builtin_amdgcn_readfirstlane(n_listed);
    if (n_listed == 0u) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS stores, in order
    uint32_t base = 0u;
    if (lane == 0) base = atomicAdd(fix_counter(L), n_listed);
    base = __builtin_amdgcn_readfirstlane(base);
    // (called with the lanes beyond the frame's right edge switched off when the width is not a multiple of 64: the entries are
    // dealt out over the lanes that ARE active - a stride of 64 over lane numbers would leave the others' entries unwritten)
    const uint64_t act = __builtin_amdgcn_ballot_w64(true);
    const uint32_t n_act = (uint32_t)__builtin_popcountll(act), rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
    for (uint32_t i = rank; i < n_listed; i += n_act) fix_list(L)[base + i] = my_list[i];
    asm volatile("" ::: "memory");
    n_listed = __builtin_amdgcn_readfirstlane(0u);
  };
  for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
    const int strip = tile * kTabWaves + wave;
    if (strip < n_strips) {
      const int z = strip / strips_per_frame, rem = strip - z * strips_per_frame;
      const int rs = rem / cgs, x = (rem - rs * cgs) * 64 + lane, ys = rs * kStripRows;
      if (x < W) {
        // wave-uniform frame bases as buffer resources: a lane's column offset is fixed, the row offset is a scalar
        const __amdgpu_buffer_rsrc_t r_img = frame_rsrc(frame_ptr(L.in, z), W, H);
        const __amdgpu_buffer_rsrc_t r_out = frame_rsrc(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z, W, H);
        const float wx = cols[x];
        const int xo = x * 4, xro = (x + 1 < W ? x + 1 : W - 1) * 4;
        const int tri_x = (2 * x + 1) * H;
        // Window of source rows around the four target rows of a step (rows y0 - 2 .. y0 + 5, slot i = row y0 - 2 + i): per row
        // and channel the sampler's horizontal lerp `crow` exactly as the GL evaluates it, the difference to the next row's, the
        // own texel's decoded value and its byte as a table address.
        constexpr int kStep = 4, kWin = kStep + 4;
        float crow[kWin][3], drow[kWin][3];
        uint32_t baddr[kWin][3];
        auto decode_row = [&](uint32_t tc, uint32_t tr, int slot) __attribute__((always_inline)) {
          {
            const float d = lds_f32(kLdsDec + byte_shl<0, 2>(tc)), dr = lds_f32(kLdsDec + byte_shl<0, 2>(tr));
            crow[slot][0] = fma_(wx, dr - d, d);
            baddr[slot][0] = byte_shl<0, 4>(tc) + (uint32_t)kLogNodes * 16u;
          }
          {
            const float d = lds_f32(kLdsDec + byte_shl<1, 2>(tc)), dr = lds_f32(kLdsDec + byte_shl<1, 2>(tr));
            crow[slot][1] = fma_(wx, dr - d, d);
            baddr[slot][1] = byte_shl<1, 4>(tc) + (uint32_t)kLogNodes * 16u;
          }
          {
            const float d = lds_f32(kLdsDec + byte_shl<2, 2>(tc)), dr = lds_f32(kLdsDec + byte_shl<2, 2>(tr));
            crow[slot][2] = fma_(wx, dr - d, d);
            baddr[slot][2] = byte_shl<2, 4>(tc) + (uint32_t)kLogNodes * 16u;
          }
        };
        uint32_t nc[kStep], nr[kStep];   // raw texels of the four rows that enter with the next step, in flight
        auto fetch_rows = [&](int first) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < kStep; ++i) {
            const int ro = clampi(first + i, 0, H - 1) * W * 4;
            nc[i] = __builtin_amdgcn_raw_buffer_load_b32(r_img, xo, ro, 0);
            nr[i] = __builtin_amdgcn_raw_buffer_load_b32(r_img, xro, ro, 0);
          }
        };
        fetch_rows(ys - 2);
#pragma unroll
        for (int i = 0; i < kStep; ++i) decode_row(nc[i], nr[i], i);
        fetch_rows(ys + 2);
        uint32_t failed = 0u;   // bit k: row ys + k of this column is not certain
#pragma unroll 1
        for (int k0 = 0; k0 < kStripRows; k0 += kStep) {
          const int y0 = ys + k0;
          if (y0 >= H) break;
#pragma unroll
          for (int i = 0; i < kStep; ++i) decode_row(nc[i], nr[i], kStep + i);
          fetch_rows(y0 + kStep + 2);
#pragma unroll
          for (int i = 0; i + 1 < kWin; ++i)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) drow[i][ch] = crow[i + 1][ch] - crow[i][ch];
#pragma unroll
          for (int k = 0; k < kStep; ++k) {
            const int y = y0 + k;
            if (y < H) {
              const ScanRow ri = rows[y];
              const bool lo = (2 * y + 1) * W <= tri_x;
              const float dist = lo ? ri.dist_lo : ri.dist_up;
              uint32_t fail = (y < 2 || y >= H - 2) ? 1u : ((ri.up >> (lo ? 8 : 9)) & 1u);
              uint32_t px = 0xff000000u;
              // scanline j of the target row: the row pair (wi - 1, wi) or (wi, wi + 1) around window slot wi, by ri.up
              const int w0 = k + 2, w1 = k + 3, w2 = k + 1;
              const bool up0 = ri.up & 1u, up1 = ri.up & 2u, up2 = ri.up & 4u;
#define RC_SCAN_CH(ch)                                                                                                        \
              {                                                                                                               \
                const float c0 = fma_(ri.wy[0], up0 ? drow[w0 - 1][ch] : drow[w0][ch], up0 ? crow[w0 - 1][ch] : crow[w0][ch]);  \
                const float c2 = fma_(ri.wy[2], up2 ? drow[w2 - 1][ch] : drow[w2][ch], up2 ? crow[w2 - 1][ch] : crow[w2][ch]);  \
                float bs = SKIP1 ? (ch == 0 ? skip_r : (ch == 1 ? skip_g : skip_b)) : 0.0f, q1 = 0.0f;                                                               \
                const float q0 = scan_tab_eval<0 + ch>(c0, baddr[w0][ch], dist, &bs);                             \
                if (!SKIP1) {                                                                                                 \
                  const float c1 = fma_(ri.wy[1], up1 ? drow[w1 - 1][ch] : drow[w1][ch], up1 ? crow[w1 - 1][ch] : crow[w1][ch]); \
                  q1 = scan_tab_eval<3 + ch>(c1, baddr[w1][ch], dist, &bs);                                       \
                }                                                                                                             \
                const float q2 = scan_tab_eval<6 + ch>(c2, baddr[w2][ch], dist, &bs);                             \
                const float s = (q0 + q1) + q2;   /* (the records are halved: this is ((K0 + K1) + K2) * 0.5 bit for bit) */           \
                const float b = fma_(2.5e-7f, s, bs);   /* the sums' roundings: at most four of 2^-24 (2 s) each, halved */           \
                bool ok;                                                                                                      \
                const uint32_t byte = srgb8_interval(s, b, &ok);                                                              \
                fail |= ok ? 0u : 1u;                                                                                         \
                px |= byte << (8 * ch);                                                                                       \
              }
              RC_SCAN_CH(0)
              RC_SCAN_CH(1)
              RC_SCAN_CH(2)
#undef RC_SCAN_CH
              if (fail == 0u) __builtin_amdgcn_raw_buffer_store_b32(px, r_out, xo, y * W * 4, 0);
              else failed |= 1u << (k0 + k);
            }
          }
          // the last four rows of the window are the first four of the next step's
#pragma unroll
          for (int i = 0; i < kStep; ++i)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
              crow[i][ch] = crow[i + kStep][ch];
              baddr[i][ch] = baddr[i + kStep][ch];
            }
        }
        while (true) {
          const uint64_t any = __builtin_amdgcn_ballot_w64(failed != 0u);
          if (any == 0ull) break;
          const uint32_t n = (uint32_t)__builtin_popcountll(any);
          // The wave's count, taken HERE from the first active lane - lane 0, which takes part in every strip: a lane that sat out a
          // strip beyond the frame's right edge (a width that is not a multiple of 64) still holds the count from before that strip in
          // its own copy, and inside the branch below the first active lane is the first FAILING lane, which may be such a lane.
          uint32_t cur = __builtin_amdgcn_readfirstlane(n_listed);
          if (cur + n > (uint32_t)kWaveList) {
            flush();
            cur = 0u;
          }
          if (failed) {
            const int kk = __builtin_ctz(failed);
            failed &= failed - 1u;
            const uint32_t pos = cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
            my_list[pos] = (uint32_t)((z * H + ys + kk) * W + x);
          }
          n_listed = cur + n;
        }
      }
    }
  }
  flush();
}

// The listed pixels in the general form (k_royale_scan_v's arithmetic: four packed pairs and one scalar evaluation).
template <class SI, class SO>
__global__ void __launch_bounds__(256) k_royale_scan_v_fix(const PassLaunch L) {
  RC_SRGB_LDS_OF(lds, L, L.in);
  const float sigma_range = maxps(0.3f, 0.02f) - 0.02f, shape_range = maxps(4.0f, 2.0f) - 2.0f;
  const float off = L.params[RP1_PH] / 3.0f;
  const uint32_t stride = gridDim.x * 256u;
  {
    const uint32_t n = *fix_counter(L);
    const uint32_t* list = fix_list(L);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
      const uint32_t p = list[i];
      if (p >= (uint32_t)L.n_frames * (uint32_t)L.out_h * (uint32_t)L.out_w) {   // never expected: a list entry that is not a pixel of the launch
        atomicAdd(fix_counter(L) + 1, 1u);
        continue;
      }
      const uint32_t row = p / (uint32_t)L.out_w;
      const int x = (int)(p - row * (uint32_t)L.out_w), z = (int)(row / (uint32_t)L.out_h), y = (int)(row - (uint32_t)z * (uint32_t)L.out_h);
      float col[9], dd[9], kk[9];
      scan_v_gather<SI>(L, lds, x, y, z, col, dd);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const v2f k = beam_k<v2f, SI::kUnitRange>(v2f{col[2 * j], col[2 * j + 1]}, v2f{dd[2 * j], dd[2 * j + 1]}, off, sigma_range, shape_range);
        kk[2 * j] = k.x;
        kk[2 * j + 1] = k.y;
      }
      kk[8] = beam_k<float, SI::kUnitRange>(col[8], dd[8], off, sigma_range, shape_range);
      SO::put(L, z, x, y, make_float4(((kk[0] + kk[3]) + kk[6]) * 0.5f, ((kk[1] + kk[4]) + kk[7]) * 0.5f, ((kk[2] + kk[5]) + kk[8]) * 0.5f, 1.0f), &lds);
    }
  }
}

// ------------------------------------------------------------------- host side of the table form ------
// K(colour, distance) in closed form and double precision (the GLSL's formulas, libm instead of the GL's
// polynomials): used for the expansion coefficients and the remainder bounds only, never for a stored value.
struct BeamModel {
  double off;
  double K(double c, double dd) const {
    if (!(c > 0.0)) return 0.0;
    const double sigma = 0.02 + 0.28 * std::cbrt(c), alpha = std::sqrt(2.0) * sigma, beta = 2.0 + 2.0 * std::pow(c, 0.25);
    const double s = 1.0 / beta, g = 1.12906830989, c0 = 0.8109119309638332633713423362694399653724431,
                 c1 = 0.4808354605142681877121661197951496120000040;
    const double gam = std::pow((s + 0.5 + g) / 2.71828182845904523536, s + 0.5) * (c0 + c1 / (s + 1.0)) * beta;
    const double scale = c * beta * 0.5 / alpha / gam;
    const double d1 = std::fabs(dd), d2 = std::fabs(dd + off), d3 = std::fabs(std::fabs(dd - off));
    const double w = std::exp(-std::pow(d1 / alpha, beta)) + std::exp(-std::pow(d2 / alpha, beta)) + std::exp(-std::pow(d3 / alpha, beta));
    return scale / 3.0 * w;
  }
};
double modelDd(int j, int ch, double dist) {
  const double conv[3] = {(double)0.2f, (double)0.4f, (double)0.6f};
  if (j == 0) return dist - conv[ch];
  if (j == 1) return std::fabs((1.0 + conv[ch]) - dist);
  return dist + (1.0 - conv[ch]);
}
float nodeColour(int n) {
  return n < kLogNodes ? bits2f(kLogBits0 + ((uint32_t)n << 20) + (1u << 19)) : k_srgb_decode_host[n - kLogNodes];
}

// host-built part: the expansion coefficients around every node (A: 0, dK/dc, d2K/dc2 / 2, dK/ddist; the node value T is
// filled in on the device by k_scan_tab_nodes) and the node colours.  Any coefficients are valid - the bound is measured
// against what is stored (k_scan_tab_bounds) - good ones make it small.
void buildScanTablesHost(float off, std::vector<float>* A, std::vector<float>* node) {
  A->assign((size_t)9 * kNodes * 4, 0.0f);
  node->assign((size_t)kNodes, 0.0f);
  const BeamModel M{(double)off};
  for (int n = 0; n < kNodes; ++n) (*node)[(size_t)n] = nodeColour(n);
  for (int jc = 0; jc < 9; ++jc) {
    const int j = jc / 3, ch = jc % 3;
    const double D0 = modelDd(j, ch, 0.0);
    for (int n = 0; n < kNodes; ++n) {
      const size_t i = (size_t)jc * kNodes + n;
      const double c0 = (double)(*node)[(size_t)n];
      if (!(c0 > 0.0)) continue;  // the zero colour: K = 0; also takes colours below 2^-32
      const double h = 1e-4 * c0;
      const double k0 = M.K(c0, D0), kp = M.K(c0 + h, D0), km = M.K(c0 - h, D0);
      (*A)[i * 4 + 1] = (float)((kp - km) / (2.0 * h));
      (*A)[i * 4 + 2] = (float)(0.5 * (kp - 2.0 * k0 + km) / (h * h));
      (*A)[i * 4 + 3] = scan_w_slope((float)((M.K(c0, modelDd(j, ch, 1e-6)) - M.K(c0, modelDd(j, ch, -1e-6))) / 2e-6));
    }
  }
}

// Device tables for sub-pixel offset `off` and the given row distances: A (with T), measured bounds, node colours.
// The scanline BELOW the pixel's own (role 1: source row y + 1, at 1.2 .. 1.6 scanlines from the sample) contributes next to
// nothing at the shipped beam parameters: the same exhaustive sweep that measures the bounds records the largest exact K of
// every role, and when that is below kScanSkipBelow for all three channels of role 1 (measured at 1080p: 1.2e-8, 1.9e-18, 1.4e-35) the table kernel
// does not evaluate that scanline at all - it takes 0 for it and adds the channel's recorded maximum to the pixel's bound (a
// proven bound like the others, and small against them).
constexpr float kScanSkipBelow = 5e-8f;    // (a node's own bound is 1e-7 .. 1e-6)
struct ScanNodeTables {
  float4* A = nullptr;
  float* bound = nullptr;
  float* node = nullptr;
  float kmax[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // host copy: the largest exact K of each role over everything the tables cover
};
void freeScanNodeTables(ScanNodeTables* T) {
  if (T->A) (void)hipFree(T->A);
  if (T->bound) (void)hipFree(T->bound);
  if (T->node) (void)hipFree(T->node);
  *T = ScanNodeTables();
}
bool buildScanNodeTables(float off, const std::vector<float>& dists, hipStream_t s, ScanNodeTables* T) {
  if (dists.empty() || dists.size() > (size_t)kMaxDists) return false;
  std::vector<float> hA, hNode;
  buildScanTablesHost(off, &hA, &hNode);
  float* dd = nullptr;
  float* kmax = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&T->A), hA.size() * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&T->bound), (size_t)9 * kNodes * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T->node), hNode.size() * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&dd), dists.size() * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&kmax), 9 * 4) == hipSuccess;
  if (ok)
    ok = hipMemcpyAsync(T->A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(T->node, hNode.data(), hNode.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemcpyAsync(dd, dists.data(), dists.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
         hipMemsetAsync(T->bound, 0, (size_t)9 * kNodes * 4, s) == hipSuccess && hipMemsetAsync(kmax, 0, 9 * 4, s) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_scan_tab_nodes, dim3((9 * kNodes + 255) / 256), dim3(256), 0, s, T->A, T->node, off);
    hipLaunchKernelGGL(k_scan_tab_bounds, dim3(16, 9 * kNodes), dim3(256), 0, s, T->A, T->node, dd, (int)dists.size(), off, T->bound, kmax);
    hipLaunchKernelGGL(k_scan_tab_bounds_finish, dim3((9 * kNodes + 255) / 256), dim3(256), 0, s, T->A, T->bound, T->node);
    // the host vectors must outlive the asynchronous copies
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(T->kmax, kmax, 9 * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (dd) (void)hipFree(dd);
  if (kmax) (void)hipFree(kmax);
  if (!ok) freeScanNodeTables(T);
  return ok;
}

struct ScanTables {
  ScanNodeTables nodes;
  ScanRow* rows = nullptr;
  float* cols = nullptr;
  bool usable = false;
  void release() {
    freeScanNodeTables(&nodes);
    if (rows) (void)hipFree(rows);
    if (cols) (void)hipFree(cols);
    *this = ScanTables();
  }
};

// Tables for a launch's geometry: the geometry kernel and one synchronisation to learn whether the geometry is the regular one and
// which row distances it has, then the node tables with their measured bounds (a few 10^9 exact evaluations per distance: 140 ms
// at 1080p, once per geometry and process - a slow build in the sense of rcstrip::geo_tables, taken off the frame path).  Reads
// nothing of the launch but its geometry.
void buildScanTables(const PassLaunch& L, hipStream_t s, ScanTables* Tp) {
  ScanTables& T = *Tp;
  uint32_t* bad = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&T.rows), sizeof(ScanRow) * (size_t)L.out_h) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T.cols), sizeof(float) * (size_t)L.out_w) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&bad), 4) == hipSuccess;
  uint32_t hbad = 1;
  std::vector<ScanRow> hrows((size_t)L.out_h);
  if (ok) ok = hipMemsetAsync(bad, 0, 4, s) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_scan_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, T.rows, T.cols, bad);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(hrows.data(), T.rows, sizeof(ScanRow) * hrows.size(), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (bad) (void)hipFree(bad);
  size_t n_dists = 0;
  if (ok && hbad == 0) {
    // the distances of the rows that take the table form (rows 2 .. H - 3 whose distance is within kMaxDist, per triangle)
    std::vector<float> dists;
    auto add = [&](float d) {
      for (float e : dists)
        if (f2bits(e) == f2bits(d)) return;
      dists.push_back(d);
    };
    for (int y = 2; y < L.out_h - 2; ++y) {
      if (!(hrows[(size_t)y].up & 256u)) add(hrows[(size_t)y].dist_lo);
      if (!(hrows[(size_t)y].up & 512u)) add(hrows[(size_t)y].dist_up);
    }
    n_dists = dists.size();
    ok = buildScanNodeTables(L.params[RP1_PH] / 3.0f, dists, s, &T.nodes);
  }
  const bool usable = ok && hbad == 0;
  if (rc::log_enabled(rc::LogLevel::Debug) && ok) {   // how the rows' pairs sit: which of the four (scanline 0, scanline 2) patterns, weights exactly 0
    int pat[4] = {0, 0, 0, 0}, w0z = 0, w2z = 0, dz = 0;
    for (int y = 2; y < L.out_h - 2; ++y) {
      const ScanRow& r = hrows[(size_t)y];
      ++pat[(r.up & 1u) | ((r.up >> 1) & 2u)];
      w0z += r.wy[0] == 0.0f;
      w2z += r.wy[2] == 0.0f;
      dz += r.dist_lo == 0.0f;
    }
    RC_LOG_DEBUG("crt-royale scanline pass rows: up patterns (s0|s2) " + std::to_string(pat[0]) + " / " + std::to_string(pat[1]) + " / " + std::to_string(pat[2]) + " / " +
                 std::to_string(pat[3]) + ", wy0 == 0: " + std::to_string(w0z) + ", wy2 == 0: " + std::to_string(w2z) + ", dist == 0: " + std::to_string(dz));
  }
  RC_LOG_DEBUG("crt-royale scanline pass " + std::to_string(L.out_w) + "x" + std::to_string(L.out_h) + ": expansion tables " +
               (usable ? "ready (" + std::to_string(n_dists) + " row distances)"
                       : "not usable (geometry flags " + std::to_string(hbad) + ", " + std::to_string(n_dists) + " row distances), exact per-pixel form"));
  if (!usable) T.release();
  T.usable = usable;
}
std::shared_ptr<const ScanTables> scanTablesFor(const PassLaunch& L, hipStream_t s) {
  if (L.in.w != L.out_w || L.in.h != L.out_h || L.out_h < 8 || L.params[RP1_Y_STEP] != 1.0f || L.params[RP1_TSY] != (float)L.in.h) return nullptr;
  const Plane &pu = L.plane[0], &pv = L.plane[1];
  if (pu.dy_lo != 0.0f || pu.dy_up != 0.0f || pv.dx_lo != 0.0f || pv.dx_up != 0.0f) return nullptr;
  static std::mutex mu;
  static std::map<rcstrip::GeoKey, rcstrip::GeoCached<ScanTables>> cache;
  return rcstrip::geo_tables<ScanTables>(L, s, mu, cache, buildScanTables, true);
}

}  // namespace

namespace rck {
// for tests/test_royale_scan_table.py: the host-built part of the tables (A: 0, dK/dc, d2K/dc2 / 2, dK/ddist; B: 0, node colour) ...
void royale_scan_tables_host(float off, float* A, uint32_t* B) {
  std::vector<float> a, node;
  buildScanTablesHost(off, &a, &node);
  std::memcpy(A, a.data(), a.size() * sizeof(float));
  for (int jc = 0; jc < 9; ++jc)
    for (int n = 0; n < kNodes; ++n) {
      B[((size_t)jc * kNodes + n) * 2] = 0u;
      B[((size_t)jc * kNodes + n) * 2 + 1] = f2bits(node[(size_t)n]);
    }
}
// ... and the complete device tables for the given row distances: A with the node values T, and the measured bounds
hipError_t royale_scan_tables_device(float off, const float* dists, int n_dists, float* A, float* bound, hipStream_t s) {
  ScanNodeTables T;
  if (n_dists < 1 || !buildScanNodeTables(off, std::vector<float>(dists, dists + n_dists), s, &T)) return hipErrorInvalidValue;
  hipError_t e = hipMemcpy(A, T.A, (size_t)9 * kNodes * 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(bound, T.bound, (size_t)9 * kNodes * 4, hipMemcpyDeviceToHost);
  freeScanNodeTables(&T);
  return e;
}
int royale_scan_table_nodes() { return kNodes; }

#define GO(...)                                                                              \
  do {                                                                                       \
    hipLaunchKernelGGL((__VA_ARGS__), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L); \
    return hipGetLastError();                                                                \
  } while (0)
using OutS = St<FMT_SRGB8>;

hipError_t launch_royale_scan_v(const PassLaunch& L, hipStream_t s) {
  if (SrgbLinEdge::matches(L.in) && OutS::matches(L)) {
    if (L.flags & RC_FLAG_GENERAL_ONLY) GO((k_royale_scan_v<SrgbLinEdge, OutS>));
    const bool room = L.scratch && L.scratch_frame_stride >= (uint64_t)kFixHeader + (uint64_t)L.out_w * L.out_h * 4u &&
                      (uint64_t)L.n_frames * L.out_w * L.out_h < (1ull << 32);
    const std::shared_ptr<const ScanTables> T = room ? scanTablesFor(L, s) : nullptr;
    if (T) {
      const float* km = T->nodes.kmax;
      const bool skip1 = km[3] < kScanSkipBelow && km[4] < kScanSkipBelow && km[5] < kScanSkipBelow;
      const float skip_r = km[3] * 1.000001f + 1e-30f, skip_g = km[4] * 1.000001f + 1e-30f, skip_b = km[5] * 1.000001f + 1e-30f;
      auto kernel = skip1 ? k_royale_scan_v_tab<SrgbLinEdge, OutS, true> : k_royale_scan_v_tab<SrgbLinEdge, OutS, false>;
      // (set on every launch: the attribute is per device, and this needs no shared flag)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsTotalBytes) != hipSuccess)
        return hipGetLastError();
      const long strips = (long)((L.out_w + 63) / 64) * ((L.out_h + kStripRows - 1) / kStripRows) * L.n_frames;
      const long tiles = (strips + kTabWaves - 1) / kTabWaves;
      if (hipMemsetAsync(L.scratch, 0, kFixHeader, s) != hipSuccess) return hipGetLastError();
      hipLaunchKernelGGL(kernel, dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(kTabThreads), kLdsTotalBytes, s, L, T->nodes.A, T->rows, T->cols, 0.5f * skip_r, 0.5f * skip_g, 0.5f * skip_b);   // (halved like the records' bounds)
      #ifndef RC_SCAN_FIX_BLOCKS
#define RC_SCAN_FIX_BLOCKS 1024
#endif
      hipLaunchKernelGGL((k_royale_scan_v_fix<SrgbLinEdge, OutS>), dim3(RC_SCAN_FIX_BLOCKS), dim3(256), rcd::srgb_lds_bytes(L), s, L);
      return hipGetLastError();
    }
    // two rows per thread: 64 x 8 tiles
    const long tiles = (long)((L.out_w + 63) / 64) * ((L.out_h + 7) / 8) * L.n_frames;
    hipLaunchKernelGGL((k_royale_scan_v2<SrgbLinEdge, OutS>), dim3((unsigned)(tiles < 2048 ? (tiles > 0 ? tiles : 1) : 2048)), px_block(), rcd::srgb_lds_bytes(L), s, L);
    return hipGetLastError();
  }
  GO(k_royale_scan_v<SRT, StRT>);
}
}  // namespace rck
