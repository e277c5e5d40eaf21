// crt-royale passes 9 and 10 (bloom-vertical.glsl, bloom-horizontal-reconstitute.glsl): the general per-pixel
// kernels, and the strip forms that run separable geometry (royale_strip.h) - the shipped preset at any size.
#include <cstdio>
#include <cstdlib>

#include "royale_strip.h"

using namespace rcd;
using namespace rcroyale;
using namespace rcstrip;

namespace {

// ------------------------------------------------------------------------ P9 / P10 ------
// tex2Dblur17fast (bloom-vertical.glsl 7132-7176); the nine (offset, weight) pairs come from
// the host, evaluated with the run-time sigma exactly as the fragment shader would.
template <class SI>
__device__ __forceinline__ float4 blur17(const Tex& t, const uint8_t* img, float u, float v, float dx, float dy, const float* P,
                                        const SrgbLds* lds) {
  const float k[4] = {P[RPG_K78], P[RPG_K56], P[RPG_K34], P[RPG_K12]};
  const float w[4] = {P[RPG_W78], P[RPG_W56], P[RPG_W34], P[RPG_W12]};
  // source order, except that the centre term (weight 1.0: a plain addend) is added before the
  // product that precedes it, as in k_blur9: (((A+B)+C) + centre) + D, then the four right-hand taps
  float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 s = SI::get(t, img, u - k[i] * dx, v - k[i] * dy, lds);
    sx += w[i] * s.x; sy += w[i] * s.y; sz += w[i] * s.z;
  }
  {
    const float4 d = SI::get(t, img, u - k[3] * dx, v - k[3] * dy, lds);
    const float4 s = SI::get(t, img, u, v, lds);
    sx += 1.0f * s.x; sy += 1.0f * s.y; sz += 1.0f * s.z;
    sx += w[3] * d.x; sy += w[3] * d.y; sz += w[3] * d.z;
  }
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    const float4 s = SI::get(t, img, u + k[i] * dx, v + k[i] * dy, lds);
    sx += w[i] * s.x; sy += w[i] * s.y; sz += w[i] * s.z;
  }
  const float si = P[RPG_SUM_INV];
  return make_float4(sx * si, sy * si, sz * si, 1.0f);
}

template <class SI, class SO>
__global__ void __launch_bounds__(256) k_royale_bloom_v(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float4 c = blur17<SI>(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), 0.0f, L.params[RPG_DXY],
                          L.params, &lds);
  SO::put(L, z, x, y, c, &lds);
  RC_TILE_LOOP_END
}

// bloom-horizontal-reconstitute.glsl FS 11407-11439.
// extra[0] = PassPrev3 (MASKED_SCANLINES), extra[1] = PassPrev2 (BRIGHTPASS), extra[2] = PassPrev6 (HALATION_BLUR)
template <class SI, class S0, class S1, class S2, class SO>
__global__ void __launch_bounds__(256, 4) k_royale_bloom_h(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float4 blurred = blur17<SI>(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), L.params[RPG_DXY], 0.0f,
                                L.params, &lds);
  const float4 idim = S0::get(L.extra[0], frame_ptr(L.extra[0], z), vary(L.plane[2], x, y, lo), vary(L.plane[3], x, y, lo), &lds);
  const float4 bright = S1::get(L.extra[1], frame_ptr(L.extra[1], z), vary(L.plane[4], x, y, lo), vary(L.plane[5], x, y, lo), &lds);
  const float4 hal = S2::get(L.extra[2], frame_ptr(L.extra[2], z), vary(L.plane[6], x, y, lo), vary(L.plane[7], x, y, lo), &lds);
  const float mask_amplify = L.params[RPG_MASK_AMPLIFY];
  const float i3[3] = {idim.x, idim.y, idim.z}, b3[3] = {bright.x, bright.y, bright.z}, bl[3] = {blurred.x, blurred.y, blurred.z};
  const float h3[3] = {hal.x, hal.y, hal.z};
  float out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dimpass = i3[c] - b3[c];
    // lerp(phosphor_bloom, diffusion_color, diffusion_weight) with compile-time parameters: a*(1-t) + b*t,
    // the constant factors of a*(1-t) gathered into one by the GL's compiler (float goldens)
    out[c] = (dimpass + bl[c]) * ((mask_amplify * 2.0f) * (1.0f - 0.075f)) + h3[c] * 0.075f;
  }
  SO::put(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
  RC_TILE_LOOP_END
}

// ------------------------------------------------------------------ P9, strip form ------
// bloom-vertical samples its input (pass 8's target) through a NEAREST sampler: the nine taps of tex2Dblur17fast
// are nine texels of the pixel's own column, at row offsets that are the same for every row of a geometry (the
// taps sit 0.03 - 0.47 texel away from a texel boundary, float rounding of the coordinate is 1e-4): 0, -+(1 or 2),
// -+(3 or 4), -+(5 or 6), -+(7 or 8), fixed by the blur's sigma.  k_bloomv_geometry verifies that with the sampler's
// operations for every row and column of both triangles and returns the offsets; the strip kernel is instantiated
// per offset pattern, a thread walks kBvRows rows of one column with the 16 + kBvRows decoded texels in registers.
constexpr int kBvRows = 8;
struct BvTables {
  int pattern = -1;   // bit q: the q-th tap pair (k12, k34, k56, k78) sits at distance 2q + 2 instead of 2q + 1
  bool usable = false;
};
__global__ void __launch_bounds__(256) k_bloomv_geometry(const PassLaunch L, int* offs, uint32_t* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* P = L.params;
  const float k[4] = {P[RPG_K78], P[RPG_K56], P[RPG_K34], P[RPG_K12]};
  const float dy = P[RPG_DXY];
  uint32_t why = 0u;
  auto row_offsets = [&](int y, bool lo, int* o) {
    const float v = vary(L.plane[1], 0, y, lo);
    for (int q = 0; q < 4; ++q) {
      o[q] = (int)__builtin_floorf((v - k[q] * dy) * (float)L.in.h) - y;
      o[8 - q] = (int)__builtin_floorf((v + k[q] * dy) * (float)L.in.h) - y;
    }
    o[4] = (int)__builtin_floorf(v * (float)L.in.h) - y;
  };
  int ref[9];
  row_offsets(L.out_h / 2, true, ref);
  if (i == 0)
    for (int q = 0; q < 9; ++q) offs[q] = ref[q];
  if (i < L.out_h)
    for (int side = 0; side < 2; ++side) {
      int o[9];
      row_offsets(i, side == 0, o);
      for (int q = 0; q < 9; ++q)
        if (o[q] != ref[q]) why |= 1u;
    }
  if (i < L.out_w)
    for (int side = 0; side < 2; ++side) {
      const float u = vary(L.plane[0], i, 0, side == 0);
      // every tap: u -+ k * 0 = u
      if ((int)__builtin_floorf((u - k[0] * 0.0f) * (float)L.in.w) != i) why |= 2u;
    }
  if (why) atomicOr(bad, why);
}

template <class SO, int PATTERN>
__global__ void __launch_bounds__(512) k_royale_bloom_v_strip(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  constexpr int o12 = 1 + ((PATTERN >> 0) & 1), o34 = 3 + ((PATTERN >> 1) & 1), o56 = 5 + ((PATTERN >> 2) & 1), o78 = 7 + ((PATTERN >> 3) & 1);
  const float* P = L.params;
  const float w78 = P[RPG_W78], w56 = P[RPG_W56], w34 = P[RPG_W34], w12 = P[RPG_W12], si = P[RPG_SUM_INV];
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const StripGrid<kBvRows> G(L.out_w, L.out_h, L.n_frames);
  const int W = G.W, H = G.H, Hin = L.in.h;
  for (int strip = (int)blockIdx.x * 8 + wave; strip < G.total; strip += (int)gridDim.x * 8) {
    int z, xw, ys;
    G.locate(strip, &z, &xw, &ys);
    const int x = xw + lane;
    if (x >= W) continue;
    const uint32_t* img = reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z));
    float win[kBvRows + 16][3];
#pragma unroll
    for (int j = 0; j < kBvRows + 16; ++j) {
      const uint32_t t = img[clampi(ys - 8 + j, 0, Hin - 1) * L.in.w + x];
      win[j][0] = lds.dec[t & 255u];
      win[j][1] = lds.dec[(t >> 8) & 255u];
      win[j][2] = lds.dec[(t >> 16) & 255u];
    }
#pragma unroll
    for (int k = 0; k < kBvRows; ++k) {
      const int y = ys + k;
      if (y >= H) break;
      float out[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        // tex2Dblur17fast in the GL's evaluation order (blur17 above)
        float s = w78 * win[k + 8 - o78][ch];
        s += w56 * win[k + 8 - o56][ch];
        s += w34 * win[k + 8 - o34][ch];
        s += 1.0f * win[k + 8][ch];
        s += w12 * win[k + 8 - o12][ch];
        s += w12 * win[k + 8 + o12][ch];
        s += w34 * win[k + 8 + o34][ch];
        s += w56 * win[k + 8 + o56][ch];
        s += w78 * win[k + 8 + o78][ch];
        out[ch] = s * si;
      }
      SO::put(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
    }
  }
}

void buildBvTables(const PassLaunch& L, hipStream_t s, BvTables* T) {
  int* offs = nullptr;
  uint32_t* bad = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&offs), 9 * sizeof(int)) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&bad), 4) == hipSuccess;
  uint32_t hbad = 1;
  int ho[9] = {0};
  if (ok) ok = hipMemsetAsync(bad, 0, 4, s) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_bloomv_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, offs, bad);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(ho, offs, sizeof(ho), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (offs) (void)hipFree(offs);
  if (bad) (void)hipFree(bad);
  int pattern = 0;
  if (ok && hbad == 0 && ho[4] == 0) {
    for (int q = 0; q < 4; ++q) {   // ho[3 - q] / ho[5 + q]: the pair at nominal distance 2q + 1
      const int d = ho[5 + q];
      if (ho[3 - q] != -d || (d != 2 * q + 1 && d != 2 * q + 2)) ok = false;
      if (d == 2 * q + 2) pattern |= 1 << q;
    }
  } else {
    ok = false;
  }
  T->usable = ok;
  T->pattern = ok ? pattern : -1;
  if (std::getenv("RC_DEBUG_SCAN")) std::fprintf(stderr, "[rc bloom-v] %dx%d: ok %d flags %u pattern %d\n", L.out_w, L.out_h, (int)ok, hbad, T->pattern);
}

// ----------------------------------------------------------------- P10, strip form ------
// One wave renders 64 columns x kBhRows rows, TWO target rows per lane and step, so that every float operation of the
// filter runs as a packed v_pk_{fma,mul,add}_f32 over the row pair (the same IEEE operation per component).
//
// The nine blur taps of a pixel are LINEAR taps of one row (or row pair) of pass 9's sRGB8 target at horizontal offsets of
// up to 7.x texels: lerp(wx, T[j], T[j+1]) = fma(wx, T[j+1] - T[j], T[j]).  The difference D[j] = T[j+1] - T[j] does not
// depend on the tap, so it is formed ONCE per staged texel (the neighbour's texel arrives by a DPP wave shift folded into
// the subtraction) and a tap is one fma.  Per step the wave stages, for its 64 columns + 10 either side (clamped like
// the sampler clamps them), per column and channel the quad {T(lo row), T(hi row), D(lo row), D(hi row)} of the row pair's
// FIRST source rows ("top" slot) and, if either row has a vertical weight, of their SECOND source rows ("bottom" slot): one
// ds_read_b128 per tap, channel and slot delivers both rows' operands in adjacent registers.  Column quantities (tap
// offset as an LDS address, weight) sit in registers per strip, row quantities are wave-uniform scalars, global
// accesses are buffer loads / stores with the row base in an SGPR (no per-lane 64-bit address arithmetic).
// At 1:1 a target row's first source row is y or y - 1 (weight 0 / a few 1e-5 / 1 minus that, irregularly from row to row:
// k_bloomh_geometry), so the wave keeps a rolling window of four decoded source rows y - 1 .. y + 2 in registers and two new
// rows enter per step.  A strip the quad's diagonal crosses is rendered once per triangle, each pixel stored by the
// pass of its own triangle.
constexpr int kBhRows = 16;
constexpr int kBhWaves = 12;
constexpr int kBhSeg = 84;      // staged columns: 10 + 64 + 10
constexpr int kBhSegLeft = 10;
constexpr int kBhColBytes = 48;                       // three channels x {T lo, T hi, D lo, D hi}
constexpr int kBhSlotBytes = kBhSeg * kBhColBytes;    // 4032
constexpr int kBhLdsTables = (256 + (int)kSrgb2Runs + 3) & ~3;   // dwords: decode table, second-form encode table
constexpr int kBhWaveDwords = 2 * kBhSlotBytes / 4;   // per wave: top and bottom slot
enum { BH_DX = 0, BH_WX = 9, BH_IDIM_X = 18, BH_BRIGHT_X = 19, BH_HAL_X0 = 20, BH_HAL_W = 21, BH_COL_FIELDS = 22 };
enum { BH_Y0 = 0, BH_WY = 1, BH_IDIM_Y = 2, BH_BRIGHT_Y = 3, BH_HAL_Y0 = 4, BH_HAL_WY = 5, BH_ROW_FIELDS = 8 };

struct BhTables {
  uint32_t* cols = nullptr;   // [BH_COL_FIELDS][2 sides][W] (ints and float bits)
  uint32_t* rows = nullptr;   // [H][2 sides][BH_ROW_FIELDS]
  bool usable = false;
};

__global__ void __launch_bounds__(256) k_bloomh_geometry(const PassLaunch L, uint32_t* cols, uint32_t* rows, uint32_t* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int W = L.out_w, H = L.out_h;
  const float* P = L.params;
  const float k[4] = {P[RPG_K78], P[RPG_K56], P[RPG_K34], P[RPG_K12]};
  const float dx = P[RPG_DXY];
  uint32_t why = 0u;
  if (i < W) {
    for (int side = 0; side < 2; ++side) {
      const bool lo = side == 0;
      const float u = vary(L.plane[0], i, 0, lo);
      // evaluation order of blur17 (below): -k78, -k56, -k34, -k12, centre, +k12, +k34, +k56, +k78
      float us[9];
      for (int q = 0; q < 4; ++q) {
        us[q] = u - k[q] * dx;
        us[8 - q] = u + k[q] * dx;
      }
      us[4] = u;
      for (int q = 0; q < 9; ++q) {
        const LinTap t = lin_tap(us[q], L.in.w);
        const int d = t.i0 - i;
        // staged column of the tap's first texel: lane + kBhSegLeft + d in [0, kBhSeg - 2] (its partner is the next one)
        if (d < -kBhSegLeft || d + 1 > kBhSeg - kBhSegLeft - 64) why |= 1u;
        cols[((BH_DX + q) * 2 + side) * W + i] = (uint32_t)d;
        cols[((BH_WX + q) * 2 + side) * W + i] = f2bits(t.w);
      }
      cols[(BH_IDIM_X * 2 + side) * W + i] = (uint32_t)near_tap(vary(L.plane[2], i, 0, lo), L.extra[0].w);
      cols[(BH_BRIGHT_X * 2 + side) * W + i] = (uint32_t)near_tap(vary(L.plane[4], i, 0, lo), L.extra[1].w);
      const LinTap h = lin_tap(vary(L.plane[6], i, 0, lo), L.extra[2].w);
      cols[(BH_HAL_X0 * 2 + side) * W + i] = (uint32_t)h.i0;
      cols[(BH_HAL_W * 2 + side) * W + i] = f2bits(h.w);
    }
  }
  if (i < H) {
    for (int side = 0; side < 2; ++side) {
      const bool lo = side == 0;
      uint32_t* r = rows + ((size_t)i * 2 + side) * BH_ROW_FIELDS;
      const float v = vary(L.plane[1], 0, i, lo);
      const LinTap t = lin_tap(v - k[0] * 0.0f, L.in.h);   // every tap: v -+ k * 0 = v
      // the strip's window holds source rows y - 1 .. y + 2 of a target row pair (y, y + 1): the pair must start at y - 1 or y
      if (t.i0 < i - 1 || t.i0 > i) why |= 2u;
      r[BH_Y0] = (uint32_t)t.i0;
      r[BH_WY] = f2bits(t.w);
      r[BH_IDIM_Y] = (uint32_t)near_tap(vary(L.plane[3], 0, i, lo), L.extra[0].h);
      r[BH_BRIGHT_Y] = (uint32_t)near_tap(vary(L.plane[5], 0, i, lo), L.extra[1].h);
      const LinTap h = lin_tap(vary(L.plane[7], 0, i, lo), L.extra[2].h);
      r[BH_HAL_Y0] = (uint32_t)h.i0;
      r[BH_HAL_WY] = f2bits(h.w);
      r[6] = r[7] = 0u;
    }
  }
  if (why) atomicOr(bad, why);
}

typedef float v4f __attribute__((ext_vector_type(4)));

struct BhRow {
  int y0;
  float wy;
  int idim_y, bright_y, hal_y0;
  float hal_wy;
};
__device__ __forceinline__ BhRow load_bh_row(const uint32_t* __restrict__ rows, int y, int side) {
  const uint32_t* r = rows + ((size_t)y * 2 + side) * BH_ROW_FIELDS;
  return BhRow{(int)r[BH_Y0], bits2f(r[BH_WY]), (int)r[BH_IDIM_Y], (int)r[BH_BRIGHT_Y], (int)r[BH_HAL_Y0], bits2f(r[BH_HAL_WY])};
}

// One decoded source row as a lane holds it: the texel of its main staged column (lane) and of its extra one (63 + lane,
// lanes 0..20) per channel, and the difference to the next staged column's texel.
struct BhDec {
  float t0[3], d0[3], t1[3], d1[3];
};
// value of the next lane (lane 63: 0); the compiler folds the move into the subtraction that consumes it (v_sub_f32_dpp)
__device__ __forceinline__ float next_lane_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ void bh_decode(BhDec& r, uint32_t tm, uint32_t te, const float* dec) {
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    r.t0[ch] = dec[(tm >> (8 * ch)) & 255u];
    r.t1[ch] = dec[(te >> (8 * ch)) & 255u];
    // main lane 63's difference is wrong (its neighbour is the extra batch's lane 1): the extra batch starts at staged column 63
    // and overwrites that entry (bh_stage)
    r.d0[ch] = next_lane_dpp(r.t0[ch]) - r.t0[ch];
    r.d1[ch] = next_lane_dpp(r.t1[ch]) - r.t1[ch];
  }
}
// the slot entries of this lane's two staged columns for the row pair (lo, hi)
__device__ __forceinline__ void bh_stage(uint8_t* slot, int lane, const BhDec& lo, const BhDec& hi) {
  uint8_t* p0 = slot + lane * kBhColBytes;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    *reinterpret_cast<float2*>(p0 + 16 * ch) = make_float2(lo.t0[ch], hi.t0[ch]);
    *reinterpret_cast<float2*>(p0 + 16 * ch + 8) = make_float2(lo.d0[ch], hi.d0[ch]);
  }
  if (lane < kBhSeg - 63) {
    uint8_t* p1 = p0 + 63 * kBhColBytes;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      *reinterpret_cast<float2*>(p1 + 16 * ch) = make_float2(lo.t1[ch], hi.t1[ch]);
      *reinterpret_cast<float2*>(p1 + 16 * ch + 8) = make_float2(lo.d1[ch], hi.d1[ch]);
    }
  }
}

struct BhCtx {   // per-strip lane state
  const float* dec;
  const uint32_t* enc2;
  uint8_t* slots;
  const uint8_t* tap[9];   // LDS address of tap q's entry in the top slot
  float wx[9];
  float w78, w56, w34, w12, si, c_main;
};

// The filter and the reconstitute of one row pair.  TWO: at least one of the rows has a vertical weight.
template <bool TWO>
__device__ __forceinline__ void bh_compute(const BhCtx& c, v2f wy2, uint32_t ia, uint32_t ib, uint32_t ja, uint32_t jb, const v2f* hal2, uint32_t* pa,
                                           uint32_t* pb) {
  auto tap = [&](int q, v2f* h) {
    const v2f wx2 = {c.wx[q], c.wx[q]};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const v4f e = *reinterpret_cast<const v4f*>(c.tap[q] + 16 * ch);
      h[ch] = __builtin_elementwise_fma(wx2, v2f{e.z, e.w}, v2f{e.x, e.y});
      if (TWO) {
        const v4f f = *reinterpret_cast<const v4f*>(c.tap[q] + kBhSlotBytes + 16 * ch);
        const v2f hb = __builtin_elementwise_fma(wx2, v2f{f.z, f.w}, v2f{f.x, f.y});
        h[ch] = __builtin_elementwise_fma(wy2, hb - h[ch], h[ch]);
      }
    }
  };
  // tex2Dblur17fast in the GL's evaluation order (blur17 above): taps 0 1 2, the centre (weight 1: a plain addend) before tap 3, 5 .. 8
  v2f s[3], h[3];
  tap(0, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] = c.w78 * h[ch];
  tap(1, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += c.w56 * h[ch];
  tap(2, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += c.w34 * h[ch];
  tap(4, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += h[ch];
  tap(3, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += c.w12 * h[ch];
  tap(5, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += c.w12 * h[ch];
  tap(6, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += c.w34 * h[ch];
  tap(7, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += c.w56 * h[ch];
  tap(8, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += c.w78 * h[ch];
  uint32_t oa = 0xff000000u, ob = 0xff000000u;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const v2f bl = s[ch] * c.si;
    const v2f idim = {c.dec[(ia >> (8 * ch)) & 255u], c.dec[(ib >> (8 * ch)) & 255u]};
    const v2f bright = {c.dec[(ja >> (8 * ch)) & 255u], c.dec[(jb >> (8 * ch)) & 255u]};
    const v2f dimpass = idim - bright;
    const v2f o = (dimpass + bl) * c.c_main + hal2[ch] * 0.075f;   // as k_royale_bloom_h
    oa |= srgb8_t2(o.x, c.enc2) << (8 * ch);
    ob |= srgb8_t2(o.y, c.enc2) << (8 * ch);
  }
  *pa = oa;
  *pb = ob;
}

// One strip as seen from one triangle (SIDE 0 lower, 1 upper).  `mixed`: the diagonal crosses the strip and only the
// pixels of this triangle are stored.
template <int SIDE>
__device__ __forceinline__ void bloomh_strip_side(const PassLaunch& L, const float* dec, const uint32_t* enc2, uint8_t* slots,
                                                  const uint32_t* __restrict__ cols, const uint32_t* __restrict__ rows, int z, int xw, int ys, int lane,
                                                  bool mixed) {
  const int W = L.out_w, H = L.out_h, Win = L.in.w, Hin = L.in.h;
  const int x = xw + lane;
  const bool live = x < W;
  const int xc = live ? x : W - 1;
  const float* P = L.params;
  BhCtx c;
  c.dec = dec;
  c.enc2 = enc2;
  c.slots = slots;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    c.tap[q] = slots + (uint32_t)((int)cols[((BH_DX + q) * 2 + SIDE) * W + xc] + (xc - xw) + kBhSegLeft) * (uint32_t)kBhColBytes;
    c.wx[q] = bits2f(cols[((BH_WX + q) * 2 + SIDE) * W + xc]);
  }
  c.w78 = P[RPG_W78]; c.w56 = P[RPG_W56]; c.w34 = P[RPG_W34]; c.w12 = P[RPG_W12]; c.si = P[RPG_SUM_INV];
  c.c_main = (P[RPG_MASK_AMPLIFY] * 2.0f) * (1.0f - 0.075f);
  const int idim_x = (int)cols[(BH_IDIM_X * 2 + SIDE) * W + xc], bright_x = (int)cols[(BH_BRIGHT_X * 2 + SIDE) * W + xc];
  const int hal_x0 = (int)cols[(BH_HAL_X0 * 2 + SIDE) * W + xc];
  const float hal_w = bits2f(cols[(BH_HAL_W * 2 + SIDE) * W + xc]);
  // buffer resources: frame bases are wave-uniform, a lane's column offset is fixed for the strip, the row base is a scalar
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frame_ptr(L.in, z)), 0, Win * Hin * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_i0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frame_ptr(L.extra[0], z)), 0, L.extra[0].w * L.extra[0].h * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_i1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frame_ptr(L.extra[1], z)), 0, L.extra[1].w * L.extra[1].h * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_hal =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frame_ptr(L.extra[2], z)), 0, L.extra[2].w * L.extra[2].h * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out =
      __builtin_amdgcn_make_buffer_rsrc(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z, 0, W * H * 4, 0x00020000);
  // staged columns: lane l holds column xw - 10 + l and, for l < 21, column xw + 53 + l (staged column 63 + l)
  const int sx0 = clampi(xw - kBhSegLeft + lane, 0, Win - 1) * 4, sx1 = clampi(xw - kBhSegLeft + 63 + lane, 0, Win - 1) * 4;
  auto fetch = [&](int r, uint32_t* tm, uint32_t* te) {
    const int ro = clampi(r, 0, Hin - 1) * Win * 4;
    *tm = __builtin_amdgcn_raw_buffer_load_b32(r_in, sx0, ro, 0);
    *te = __builtin_amdgcn_raw_buffer_load_b32(r_in, sx1, ro, 0);
  };
  const int hw = L.extra[2].w, hh = L.extra[2].h;
  const int hxa = clampi(hal_x0, 0, hw - 1) * 4, hxb = clampi(hal_x0 + 1, 0, hw - 1) * 4;
  auto hal_hrow = [&](int r, float* h) {   // the sampler's horizontal lerp of halation row r (clamped), three channels
    const int ro = clampi(r, 0, hh - 1) * hw * 4;
    const uint32_t ta = __builtin_amdgcn_raw_buffer_load_b32(r_hal, hxa, ro, 0), tb = __builtin_amdgcn_raw_buffer_load_b32(r_hal, hxb, ro, 0);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float a = dec[(ta >> (8 * ch)) & 255u], b = dec[(tb >> (8 * ch)) & 255u];
      h[ch] = fma_(hal_w, b - a, a);
    }
  };
  uint8_t* top = slots;
  uint8_t* bot = slots + kBhSlotBytes;
  // window of decoded source rows y - 1, y, y + 1, y + 2 of the current target row pair; two rows enter per step
  BhDec rm1, r0, r1, r2;
  uint32_t n1m, n1e, n2m, n2e;   // raw texels of the two rows that enter next, in flight
  {
    uint32_t am, ae, bm, be;
    fetch(ys - 1, &am, &ae);
    fetch(ys, &bm, &be);
    fetch(ys + 1, &n1m, &n1e);
    fetch(ys + 2, &n2m, &n2e);
    bh_decode(r1, am, ae, dec);   // shifted into place at the top of the first step
    bh_decode(r2, bm, be, dec);
  }
  float hl[3][3];   // horizontally filtered halation rows hbase, hbase + 1, hbase + 2
  int hbase = -1000, hvalid = 0;
#pragma unroll 2
  for (int k = 0; k < kBhRows; k += 2) {
    const int y = ys + k;
    if (y >= H) break;
    const int yb = y + 1 < H ? y + 1 : y;
    const BhRow ra = load_bh_row(rows, y, SIDE), rb = load_bh_row(rows, yb, SIDE);
    // the NEAREST taps of the two rows (consumed after the filter)
    const uint32_t ia = __builtin_amdgcn_raw_buffer_load_b32(r_i0, idim_x * 4, ra.idim_y * L.extra[0].w * 4, 0);
    const uint32_t ib = __builtin_amdgcn_raw_buffer_load_b32(r_i0, idim_x * 4, rb.idim_y * L.extra[0].w * 4, 0);
    const uint32_t ja = __builtin_amdgcn_raw_buffer_load_b32(r_i1, bright_x * 4, ra.bright_y * L.extra[1].w * 4, 0);
    const uint32_t jb = __builtin_amdgcn_raw_buffer_load_b32(r_i1, bright_x * 4, rb.bright_y * L.extra[1].w * 4, 0);
    // window: rows y + 1, y + 2 enter
    rm1 = r1;
    r0 = r2;
    bh_decode(r1, n1m, n1e, dec);
    bh_decode(r2, n2m, n2e, dec);
    fetch(y + 3, &n1m, &n1e);
    fetch(y + 4, &n2m, &n2e);
    // stage: the other lanes' reads follow in program order (the LDS executes a wave's operations in order); the compiler
    // must not move LDS accesses across the staging
    const bool up_a = ra.y0 < y, up_b = rb.y0 < y + 1;   // the row's pair starts one row above it
    const bool two = ra.wy != 0.0f || rb.wy != 0.0f;
    asm volatile("" ::: "memory");
    if (up_a) {
      if (up_b) bh_stage(top, lane, rm1, r0); else bh_stage(top, lane, rm1, r1);
    } else {
      if (up_b) bh_stage(top, lane, r0, r0); else bh_stage(top, lane, r0, r1);
    }
    if (two) {
      if (up_a) {
        if (up_b) bh_stage(bot, lane, r0, r1); else bh_stage(bot, lane, r0, r2);
      } else {
        if (up_b) bh_stage(bot, lane, r1, r1); else bh_stage(bot, lane, r1, r2);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // halation: rows ra.hal_y0, +1 for the first row, rb.hal_y0, +1 for the second (the same pair or the next one)
    {
      const int a = ra.hal_y0;
      if (a != hbase) {
        if (a == hbase + 1 && hvalid >= 2) {
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            hl[0][ch] = hl[1][ch];
            hl[1][ch] = hl[2][ch];
          }
          hvalid -= 1;
        } else {
          hvalid = 0;
        }
        hbase = a;
      }
      if (hvalid < 1) hal_hrow(a, hl[0]);
      if (hvalid < 2) hal_hrow(a + 1, hl[1]);
      if (hvalid < 2) hvalid = 2;
      if (rb.hal_y0 != a && hvalid < 3) {
        hal_hrow(a + 2, hl[2]);
        hvalid = 3;
      }
    }
    v2f hal2[3];
    if (rb.hal_y0 == ra.hal_y0) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float d = hl[1][ch] - hl[0][ch];
        hal2[ch] = __builtin_elementwise_fma(v2f{ra.hal_wy, rb.hal_wy}, v2f{d, d}, v2f{hl[0][ch], hl[0][ch]});
      }
    } else if (rb.hal_y0 == ra.hal_y0 + 1) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        hal2[ch] = __builtin_elementwise_fma(v2f{ra.hal_wy, rb.hal_wy}, v2f{hl[1][ch], hl[2][ch]} - v2f{hl[0][ch], hl[1][ch]}, v2f{hl[0][ch], hl[1][ch]});
    } else {   // never at magnification >= 1: the second row's pair on its own
      float b0[3], b1[3];
      hal_hrow(rb.hal_y0, b0);
      hal_hrow(rb.hal_y0 + 1, b1);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        hal2[ch] = v2f{fma_(ra.hal_wy, hl[1][ch] - hl[0][ch], hl[0][ch]), fma_(rb.hal_wy, b1[ch] - b0[ch], b0[ch])};
    }
    uint32_t pa, pb;
    if (two) bh_compute<true>(c, v2f{ra.wy, rb.wy}, ia, ib, ja, jb, hal2, &pa, &pb);
    else bh_compute<false>(c, v2f{0.f, 0.f}, ia, ib, ja, jb, hal2, &pa, &pb);
    asm volatile("" ::: "memory");
    const bool sa = live && (!mixed || rcd::lower_tri(x, y, W, H) == (SIDE == 0));
    const bool sb = live && y + 1 < H && (!mixed || rcd::lower_tri(x, y + 1, W, H) == (SIDE == 0));
    if (sa) __builtin_amdgcn_raw_buffer_store_b32(pa, r_out, x * 4, y * W * 4, 0);
    if (sb) __builtin_amdgcn_raw_buffer_store_b32(pb, r_out, x * 4, (y + 1) * W * 4, 0);
  }
}

__global__ void __launch_bounds__(kBhWaves * 64, 1) k_royale_bloom_h_strip(const PassLaunch L, const uint32_t* __restrict__ cols,
                                                                          const uint32_t* __restrict__ rows) {
  extern __shared__ uint32_t rc_dyn_lds_[];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* dec = reinterpret_cast<float*>(rc_dyn_lds_);
  uint32_t* enc2 = rc_dyn_lds_ + 256;
  for (int i = tid; i < 256; i += kBhWaves * 64) dec[i] = k_srgb_decode[i];
  for (int i = tid; i < (int)kSrgb2Runs; i += kBhWaves * 64) enc2[i] = L.srgb_enc[kSrgbRuns + i];
  __syncthreads();
  uint8_t* slots = reinterpret_cast<uint8_t*>(rc_dyn_lds_ + kBhLdsTables + wave * kBhWaveDwords);
  const StripGrid<kBhRows> G(L.out_w, L.out_h, L.n_frames);
  const int W = G.W, H = G.H;
  for (int strip = (int)blockIdx.x * kBhWaves + wave; strip < G.total; strip += (int)gridDim.x * kBhWaves) {
    int z, xw, ys;
    G.locate(strip, &z, &xw, &ys);
    // the lower triangle holds the pixels with (2y+1) W <= (2x+1) H: the strip is all lower if its (min x, max y)
    // pixel is, all upper if its (max x, min y) pixel is not
    const int xmax = min(xw + 63, W - 1), ymax = min(ys + kBhRows - 1, H - 1);
    const bool all_lo = rcd::lower_tri(xw, ymax, W, H), all_up = !rcd::lower_tri(xmax, ys, W, H);
    if (!all_up) bloomh_strip_side<0>(L, dec, enc2, slots, cols, rows, z, xw, ys, lane, !all_lo);
    if (!all_lo) bloomh_strip_side<1>(L, dec, enc2, slots, cols, rows, z, xw, ys, lane, !all_up);
  }
}

void buildBhTables(const PassLaunch& L, hipStream_t s, BhTables* T) {
  uint32_t* bad = nullptr;
  const size_t colWords = (size_t)BH_COL_FIELDS * 2 * L.out_w, rowWords = (size_t)L.out_h * 2 * BH_ROW_FIELDS;
  bool ok = hipMalloc(reinterpret_cast<void**>(&T->cols), colWords * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T->rows), rowWords * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&bad), 4) == hipSuccess;
  uint32_t hbad = 1;
  if (ok) ok = hipMemsetAsync(bad, 0, 4, s) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_bloomh_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, T->cols, T->rows, bad);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (bad) (void)hipFree(bad);
  T->usable = ok && hbad == 0;
  if (!T->usable) {
    if (T->cols) (void)hipFree(T->cols);
    if (T->rows) (void)hipFree(T->rows);
    *T = BhTables();
  }
}

}  // namespace

namespace rck {
#define GO(...)                                                                              \
  do {                                                                                       \
    hipLaunchKernelGGL((__VA_ARGS__), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L); \
    return hipGetLastError();                                                                \
  } while (0)
using OutS = St<FMT_SRGB8>;

template <int PATTERN>
hipError_t launch_bloom_v_strip(const PassLaunch& L, hipStream_t s) {
  const long strips = (long)((L.out_w + 63) / 64) * ((L.out_h + kBvRows - 1) / kBvRows) * L.n_frames;
  const long blocks = (strips + 7) / 8;
  hipLaunchKernelGGL((k_royale_bloom_v_strip<OutS, PATTERN>), dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(512), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_royale_bloom_v(const PassLaunch& L, hipStream_t s) {
  if (SrgbNearEdge::matches(L.in) && OutS::matches(L)) {
    if (!(L.flags & RC_FLAG_GENERAL_ONLY) && separable(L, 0, 1) && L.in.w == L.out_w) {
      static std::mutex mu;
      static std::map<GeoKey, BvTables> cache;
      if (const BvTables* T = geo_tables<BvTables>(L, s, mu, cache, buildBvTables)) {
        switch (T->pattern) {
#define RC_BV(p) case p: return launch_bloom_v_strip<p>(L, s);
          RC_BV(0) RC_BV(1) RC_BV(2) RC_BV(3) RC_BV(4) RC_BV(5) RC_BV(6) RC_BV(7)
          RC_BV(8) RC_BV(9) RC_BV(10) RC_BV(11) RC_BV(12) RC_BV(13) RC_BV(14) RC_BV(15)
#undef RC_BV
          default: break;
        }
      }
    }
    GO(k_royale_bloom_v<SrgbNearEdge, OutS>);
  }
  GO(k_royale_bloom_v<SRT, StRT>);
}
hipError_t launch_royale_bloom_h(const PassLaunch& L, hipStream_t s) {
  if (SrgbLinEdge::matches(L.in) && SrgbNearEdge::matches(L.extra[0]) && SrgbNearEdge::matches(L.extra[1]) &&
      SrgbLinEdge::matches(L.extra[2]) && OutS::matches(L)) {
    if (!(L.flags & RC_FLAG_GENERAL_ONLY) && separable(L, 0, 4) && L.in.frame_stride && L.extra[0].frame_stride && L.extra[1].frame_stride) {
      static std::mutex mu;
      static std::map<GeoKey, BhTables> cache;
      if (const BhTables* T = geo_tables<BhTables>(L, s, mu, cache, buildBhTables)) {
        const long strips = (long)((L.out_w + 63) / 64) * ((L.out_h + kBhRows - 1) / kBhRows) * L.n_frames;
        const long blocks = (strips + kBhWaves - 1) / kBhWaves;
        const unsigned lds = (unsigned)(kBhLdsTables + kBhWaves * kBhWaveDwords) * 4u;
        auto kernel = k_royale_bloom_h_strip;
        static bool attr = false;
        if (!attr) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipGetLastError();
          attr = true;
        }
        hipLaunchKernelGGL(kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(kBhWaves * 64), lds, s, L, T->cols, T->rows);
        return hipGetLastError();
      }
    }
    GO(k_royale_bloom_h<SrgbLinEdge, SrgbNearEdge, SrgbNearEdge, SrgbLinEdge, OutS>);
  }
  GO(k_royale_bloom_h<SRT, SRT, SRT, SRT, StRT>);
}
}  // namespace rck
