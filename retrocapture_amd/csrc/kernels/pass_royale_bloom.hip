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
// One wave renders 64 columns x kBhRows rows.  The nine blur taps of a pixel read a row of the input texture
// (pass 9's sRGB8 target) through a LINEAR sampler at horizontal offsets of up to 7.x texels: the wave decodes each
// source row ONCE into a segment of floats in LDS (its 64 columns + 10 either side, indices clamped like the
// sampler clamps them) and the taps read decoded neighbours from there.  Per tap a thread holds (first texel, weight)
// of its column - for both triangles - in registers (k_bloomh_geometry).  A row whose vertical weight is exactly 0
// (the usual case at 1:1) filters one source row, any other row the pair the sampler would fetch.
constexpr int kBhRows = 16;
constexpr int kBhWaves = 12;
constexpr int kBhSeg = 84;      // staged columns: 10 + 64 + 10
constexpr int kBhSegLeft = 10;
// A staged row holds, per column j, the PAIR of texels (j, j + 1) every horizontal lerp needs: red and green
// interleaved {R[j], R[j+1], G[j], G[j+1]} (one ds_read_b128, 4 LDS cycles) and blue {B[j], B[j+1]} (one ds_read_b64,
// 2 cycles) - a float4 per texel would be read as two ds_read_b96 at 8 cycles each.  The staging lane of column j gets
// column j + 1's decoded texel from the next lane (ds_bpermute) and writes the whole pair entry with 16- and 8-byte stores.
// Twelve waves share one workgroup (one copy of the 35 KB sRGB tables + 12 rings = 108 KB of LDS, one workgroup per CU).
constexpr int kBhRowRG = kBhSeg * 16;             // bytes: the {R, R', G, G'} plane of a staged row
constexpr int kBhRowBytes = kBhSeg * 24;          // ... followed by the {B, B'} plane
constexpr int kBhRingDwords = 3 * kBhRowBytes / 4;   // per wave: three staged rows
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
      // the strip keeps three consecutive source rows staged: the pair must lie within one row of the target row
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

struct BhCol {   // one triangle's column quantities of a thread
  uint32_t off[9];   // byte offset of tap q's first texel inside a staged row
  float wx[9];
  int idim_x, bright_x, hal_x0;
  float hal_w;
};
struct BhRow {
  int y0;
  float wy;
  int idim_y, bright_y, hal_y0;
  float hal_wy;
};

__device__ __forceinline__ float3 f3(float4 v) { return make_float3(v.x, v.y, v.z); }
__device__ __forceinline__ float3 lerp3(float w, float3 a, float3 b) {
  return make_float3(fma_(w, b.x - a.x, a.x), fma_(w, b.y - a.y, a.y), fma_(w, b.z - a.z, a.z));
}
__device__ __forceinline__ float3 dec3(uint32_t t, const SrgbLds& l) {
  return make_float3(l.dec[t & 255u], l.dec[(t >> 8) & 255u], l.dec[(t >> 16) & 255u]);
}

// the pair (texel j, texel j + 1) of staged column j of a ring row
__device__ __forceinline__ void ring_store(uint8_t* row, int j, float r, float g, float b, float r1, float g1, float b1) {
  *reinterpret_cast<float4*>(row + j * 16) = make_float4(r, r1, g, g1);
  *reinterpret_cast<float2*>(row + kBhRowRG + j * 8) = make_float2(b, b1);
}
// value of the next lane (lane 63 gets lane 0's): a cross-lane read through the LDS crossbar, no memory access
__device__ __forceinline__ float next_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __builtin_bit_cast(int, v)));
}
// one target pixel; ring = this wave's three staged rows (row r in slot r % 3), `TWO`: the vertical weight is not 0
// HALC: the two horizontally filtered halation rows of the pixel's row pair are handed over (hal_rows[0..2], [3..5])
// PRE: the texels of the two NEAREST taps were fetched ahead (pre_idim, pre_bright)
template <class SO, bool TWO, bool HALC, bool PRE = false>
__device__ __forceinline__ void bloomh_pixel(const PassLaunch& L, const SrgbLds& lds, const uint8_t* ring, const BhCol& c, const BhRow& r,
                                             int x, int y, int z, const float* hal_rows, uint32_t pre_idim = 0u, uint32_t pre_bright = 0u) {
  const float* P = L.params;
  const int Hin = L.in.h;
  const uint8_t* rowA = ring + (uint32_t)(clampi(r.y0, 0, Hin - 1) % 3) * (uint32_t)kBhRowBytes;
  const uint8_t* rowB = ring + (uint32_t)(clampi(r.y0 + 1, 0, Hin - 1) % 3) * (uint32_t)kBhRowBytes;
  auto pair_lerp = [&](const uint8_t* row, int q) -> float3 {   // c.off[q]: byte offset of the pair in the blue plane (8 B per column)
    const float4 rg = *reinterpret_cast<const float4*>(row + 2u * c.off[q]);
    const float2 bb = *reinterpret_cast<const float2*>(row + kBhRowRG + c.off[q]);
    return lerp3(c.wx[q], make_float3(rg.x, rg.z, bb.x), make_float3(rg.y, rg.w, bb.y));
  };
  auto tap = [&](int q) -> float3 {
    float3 top = pair_lerp(rowA, q);
    if (TWO) top = lerp3(r.wy, top, pair_lerp(rowB, q));
    return top;
  };
  // tex2Dblur17fast in the GL's evaluation order (see blur17 above)
  const float w[4] = {P[RPG_W78], P[RPG_W56], P[RPG_W34], P[RPG_W12]};
  float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float3 s = tap(q);
    sx += w[q] * s.x; sy += w[q] * s.y; sz += w[q] * s.z;
  }
  {
    const float3 d = tap(3), s = tap(4);
    sx += 1.0f * s.x; sy += 1.0f * s.y; sz += 1.0f * s.z;
    sx += w[3] * d.x; sy += w[3] * d.y; sz += w[3] * d.z;
  }
#pragma unroll
  for (int q = 3; q >= 0; --q) {
    const float3 s = tap(8 - q);
    sx += w[q] * s.x; sy += w[q] * s.y; sz += w[q] * s.z;
  }
  const float si = P[RPG_SUM_INV];
  const float bl[3] = {sx * si, sy * si, sz * si};
  // the three single taps: MASKED_SCANLINES and BRIGHTPASS (NEAREST), HALATION_BLUR (LINEAR, 320x240)
  const uint32_t* i0 = reinterpret_cast<const uint32_t*>(frame_ptr(L.extra[0], z));
  const uint32_t* i1 = reinterpret_cast<const uint32_t*>(frame_ptr(L.extra[1], z));
  const uint32_t* i2 = reinterpret_cast<const uint32_t*>(frame_ptr(L.extra[2], z));
  const float3 idim = dec3(PRE ? pre_idim : i0[r.idim_y * L.extra[0].w + c.idim_x], lds);
  const float3 bright = dec3(PRE ? pre_bright : i1[r.bright_y * L.extra[1].w + c.bright_x], lds);
  float3 hal;
  if (HALC) {
    hal = lerp3(r.hal_wy, make_float3(hal_rows[0], hal_rows[1], hal_rows[2]), make_float3(hal_rows[3], hal_rows[4], hal_rows[5]));
  } else {
    const int hw = L.extra[2].w, hh = L.extra[2].h;
    const int hx0 = clampi(c.hal_x0, 0, hw - 1), hx1 = clampi(c.hal_x0 + 1, 0, hw - 1);
    const int hy0 = clampi(r.hal_y0, 0, hh - 1), hy1 = clampi(r.hal_y0 + 1, 0, hh - 1);
    const float3 h00 = dec3(i2[hy0 * hw + hx0], lds), h10 = dec3(i2[hy0 * hw + hx1], lds);
    const float3 h01 = dec3(i2[hy1 * hw + hx0], lds), h11 = dec3(i2[hy1 * hw + hx1], lds);
    hal = lerp3(r.hal_wy, lerp3(c.hal_w, h00, h10), lerp3(c.hal_w, h01, h11));
  }
  const float mask_amplify = P[RPG_MASK_AMPLIFY];
  const float i3[3] = {idim.x, idim.y, idim.z}, b3[3] = {bright.x, bright.y, bright.z}, h3[3] = {hal.x, hal.y, hal.z};
  float out[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float dimpass = i3[ch] - b3[ch];
    out[ch] = (dimpass + bl[ch]) * ((mask_amplify * 2.0f) * (1.0f - 0.075f)) + h3[ch] * 0.075f;   // as k_royale_bloom_h
  }
  SO::put(L, z, x, y, make_float4(out[0], out[1], out[2], 1.0f), &lds);
}

__device__ __forceinline__ BhRow load_bh_row(const uint32_t* rows, int y, int side) {
  const uint32_t* r = rows + ((size_t)y * 2 + side) * BH_ROW_FIELDS;
  return BhRow{(int)r[BH_Y0], bits2f(r[BH_WY]), (int)r[BH_IDIM_Y], (int)r[BH_BRIGHT_Y], (int)r[BH_HAL_Y0], bits2f(r[BH_HAL_WY])};
}

__device__ __forceinline__ BhCol load_bh_col(const uint32_t* cols, int W, int xc, int xw, int side) {
  BhCol c;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    c.off[q] = (uint32_t)((int)cols[((BH_DX + q) * 2 + side) * W + xc] + (xc - xw) + kBhSegLeft) * 8u;
    c.wx[q] = bits2f(cols[((BH_WX + q) * 2 + side) * W + xc]);
  }
  c.idim_x = (int)cols[(BH_IDIM_X * 2 + side) * W + xc];
  c.bright_x = (int)cols[(BH_BRIGHT_X * 2 + side) * W + xc];
  c.hal_x0 = (int)cols[(BH_HAL_X0 * 2 + side) * W + xc];
  c.hal_w = bits2f(cols[(BH_HAL_W * 2 + side) * W + xc]);
  return c;
}

// One strip whose pixels all lie in one triangle (SIDE 0 lower, 1 upper: all but the few strips the quad's diagonal
// crosses): one set of column quantities in registers, row quantities wave-uniform.  The loop is software-pipelined:
// the row quantities, the texels of the next source row to stage and the texels of the two NEAREST taps of the
// next target row are fetched one iteration ahead, so that their latency hides behind the current row's arithmetic.
template <class SO, int SIDE>
__device__ __forceinline__ void bloomh_strip_side(const PassLaunch& L, const SrgbLds& lds, uint8_t* ring, const uint32_t* __restrict__ cols,
                                                  const uint32_t* __restrict__ rows, int z, int xw, int ys, int lane) {
  const int W = L.out_w, H = L.out_h, Win = L.in.w, Hin = L.in.h;
  const int x = xw + lane;
  const bool live = x < W;
  const int xc = live ? x : W - 1;
  const BhCol c0 = load_bh_col(cols, W, xc, xw, SIDE);
  const uint32_t* img = reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z));
  const uint32_t* i0 = reinterpret_cast<const uint32_t*>(frame_ptr(L.extra[0], z));
  const uint32_t* i1 = reinterpret_cast<const uint32_t*>(frame_ptr(L.extra[1], z));
  const uint32_t* himg = reinterpret_cast<const uint32_t*>(frame_ptr(L.extra[2], z));
  // the staged segment: columns xw - 10 .. xw + 73 (clamped like the sampler clamps them); lane l fetches column
  // l and, for l < 20, column 64 + l
  const int sx0 = clampi(xw - kBhSegLeft + lane, 0, Win - 1), sx1 = clampi(xw - kBhSegLeft + 64 + lane, 0, Win - 1);
  const bool second = lane < kBhSeg - 64;
  auto fetch_row = [&](int r, uint32_t* t0, uint32_t* t1) {
    *t0 = img[r * Win + sx0];
    *t1 = second ? img[r * Win + sx1] : 0u;
  };
  auto store_row = [&](int r, uint32_t t0, uint32_t t1) {   // row r lives in ring slot r % 3
    uint8_t* slot = ring + (r % 3) * kBhRowBytes;
    const float a0 = lds.dec[t0 & 255u], a1 = lds.dec[(t0 >> 8) & 255u], a2 = lds.dec[(t0 >> 16) & 255u];
    const float b0 = lds.dec[t1 & 255u], b1 = lds.dec[(t1 >> 8) & 255u], b2 = lds.dec[(t1 >> 16) & 255u];   // lanes >= 20: unused
    // the right-hand partner of column `lane` is the next lane's texel, of column 63 it is column 64 (lane 0's second texel)
    float n0 = next_lane(a0, lane), n1 = next_lane(a1, lane), n2 = next_lane(a2, lane);
    const float m0 = next_lane(b0, lane), m1 = next_lane(b1, lane), m2 = next_lane(b2, lane);
    if (lane == 63) {
      n0 = m0;   // lane 63's next_lane(b.) is lane 0's b.: the texel of column 64
      n1 = m1;
      n2 = m2;
    }
    ring_store(slot, lane, a0, a1, a2, n0, n1, n2);
    if (second) ring_store(slot, 64 + lane, b0, b1, b2, m0, m1, m2);   // (column 83's partner is never read)
  };
  auto hal_hrow = [&](int r, float* h) {
    const int hw = L.extra[2].w;
    const uint32_t* p = himg + clampi(r, 0, L.extra[2].h - 1) * hw;
    const uint32_t ta = p[clampi(c0.hal_x0, 0, hw - 1)], tb = p[clampi(c0.hal_x0 + 1, 0, hw - 1)];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float a = lds.dec[(ta >> (8 * ch)) & 255u], b = lds.dec[(tb >> (8 * ch)) & 255u];
      h[ch] = fma_(c0.hal_w, b - a, a);
    }
  };
  int staged = -1;        // highest source row in the ring
  int pre_row = -1;       // source row whose texels are in pre_t0 / pre_t1 (fetched ahead), or -1
  uint32_t pre_t0 = 0u, pre_t1 = 0u;
  float hal_rows[6];      // the two horizontally filtered halation rows of the current pair (changes every few rows)
  int hal_have = -1000;
  BhRow cur = load_bh_row(rows, ys, SIDE);
  uint32_t cur_idim = i0[cur.idim_y * L.extra[0].w + c0.idim_x], cur_bright = i1[cur.bright_y * L.extra[1].w + c0.bright_x];
#pragma unroll 1
  for (int k = 0; k < kBhRows; ++k) {
    const int y = ys + k;
    if (y >= H) break;
    const int need_lo = clampi(cur.y0, 0, Hin - 1), need_hi = clampi(cur.y0 + 1, 0, Hin - 1);   // k_bloomh_geometry: y - 1 <= y0 <= y
    // ---- stage what this row needs (normally: nothing, or the one row fetched during the previous iteration).
    // Other lanes read what a lane writes here: the compiler must not move LDS accesses across the staging (the LDS
    // itself executes a wave's operations in order); a compiler-level memory barrier does that without draining the
    // loads that were issued ahead.
    if (staged < need_lo - 1) staged = need_lo - 1;
    asm volatile("" ::: "memory");
    while (staged < need_hi) {
      ++staged;
      if (staged == pre_row) {
        store_row(staged, pre_t0, pre_t1);
      } else {
        uint32_t t0, t1;
        fetch_row(staged, &t0, &t1);
        store_row(staged, t0, t1);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- fetch ahead for the next target row: its row quantities, the texels of its two NEAREST taps, and the source
    // row it will add to the ring (at most one: y0 grows by at most 1 per row)
    const int yn = y + 1 < H ? y + 1 : y;
    const BhRow nxt = load_bh_row(rows, yn, SIDE);
    const uint32_t nxt_idim = i0[nxt.idim_y * L.extra[0].w + c0.idim_x], nxt_bright = i1[nxt.bright_y * L.extra[1].w + c0.bright_x];
    const int next_hi = clampi(nxt.y0 + 1, 0, Hin - 1);
    pre_row = -1;
    if (next_hi > staged) {
      pre_row = staged + 1;
      fetch_row(pre_row, &pre_t0, &pre_t1);
    }
    if (cur.hal_y0 != hal_have) {
      if (cur.hal_y0 == hal_have + 1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) hal_rows[ch] = hal_rows[3 + ch];
      } else {
        hal_hrow(cur.hal_y0, hal_rows);
      }
      hal_hrow(cur.hal_y0 + 1, hal_rows + 3);
      hal_have = cur.hal_y0;
    }
    if (live) {
      if (cur.wy != 0.0f) bloomh_pixel<SO, true, true, true>(L, lds, ring, c0, cur, x, y, z, hal_rows, cur_idim, cur_bright);
      else bloomh_pixel<SO, false, true, true>(L, lds, ring, c0, cur, x, y, z, hal_rows, cur_idim, cur_bright);
    }
    cur = nxt;
    cur_idim = nxt_idim;
    cur_bright = nxt_bright;
  }
}

// A strip the quad's diagonal crosses (rare): per-lane triangle, the column quantities come from memory per pixel.
template <class SO>
__device__ __forceinline__ void bloomh_strip_mixed(const PassLaunch& L, const SrgbLds& lds, uint8_t* ring, const uint32_t* __restrict__ cols,
                                                   const uint32_t* __restrict__ rows, int z, int xw, int ys, int lane) {
  const int W = L.out_w, H = L.out_h, Win = L.in.w, Hin = L.in.h;
  const int x = xw + lane;
  const bool live = x < W;
  const int xc = live ? x : W - 1;
  const uint32_t* img = reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z));
  auto stage = [&](int r) {
    uint8_t* slot = ring + (r % 3) * kBhRowBytes;
    for (int j = lane; j < kBhSeg; j += 64) {   // (rare path: both texels of the pair fetched and decoded by the lane)
      const uint32_t t = img[r * Win + clampi(xw - kBhSegLeft + j, 0, Win - 1)], u = img[r * Win + clampi(xw - kBhSegLeft + j + 1, 0, Win - 1)];
      ring_store(slot, j, lds.dec[t & 255u], lds.dec[(t >> 8) & 255u], lds.dec[(t >> 16) & 255u], lds.dec[u & 255u], lds.dec[(u >> 8) & 255u],
                 lds.dec[(u >> 16) & 255u]);
    }
  };
  int staged = -1;
#pragma unroll 1
  for (int k = 0; k < kBhRows; ++k) {
    const int y = ys + k;
    if (y >= H) break;
    const BhRow r0 = load_bh_row(rows, y, 0), r1 = load_bh_row(rows, y, 1);
    const int need_lo = clampi(min(r0.y0, r1.y0), 0, Hin - 1);
    const int need_hi = clampi(max(r0.y0, r1.y0) + 1, 0, Hin - 1);
    if (staged < need_lo - 1) staged = need_lo - 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    while (staged < need_hi) stage(++staged);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!live) continue;
    const bool lo = rcd::lower_tri(x, y, W, H);
    const BhCol c = load_bh_col(cols, W, xc, xw, lo ? 0 : 1);
    const BhRow r{lo ? r0.y0 : r1.y0, lo ? r0.wy : r1.wy, lo ? r0.idim_y : r1.idim_y, lo ? r0.bright_y : r1.bright_y,
                  lo ? r0.hal_y0 : r1.hal_y0, lo ? r0.hal_wy : r1.hal_wy};
    bloomh_pixel<SO, true, false>(L, lds, ring, c, r, x, y, z, nullptr);
  }
}

template <class SO>
__global__ void __launch_bounds__(kBhWaves * 64, 1) k_royale_bloom_h_strip(const PassLaunch L, const uint32_t* __restrict__ cols,
                                                                          const uint32_t* __restrict__ rows) {
  RC_SRGB_LDS(lds, L);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* ring = reinterpret_cast<uint8_t*>(rc_dyn_lds_ + ((256 + (int)kSrgbRuns + 3) & ~3) + wave * kBhRingDwords);
  const StripGrid<kBhRows> G(L.out_w, L.out_h, L.n_frames);
  const int W = G.W, H = G.H;
  for (int strip = (int)blockIdx.x * kBhWaves + wave; strip < G.total; strip += (int)gridDim.x * kBhWaves) {
    int z, xw, ys;
    G.locate(strip, &z, &xw, &ys);
    // the lower triangle holds the pixels with (2y+1) W <= (2x+1) H: the strip is all lower if its (min x, max y)
    // pixel is, all upper if its (max x, min y) pixel is not
    const int xmax = min(xw + 63, W - 1), ymax = min(ys + kBhRows - 1, H - 1);
    if (rcd::lower_tri(xw, ymax, W, H)) bloomh_strip_side<SO, 0>(L, lds, ring, cols, rows, z, xw, ys, lane);
    else if (!rcd::lower_tri(xmax, ys, W, H)) bloomh_strip_side<SO, 1>(L, lds, ring, cols, rows, z, xw, ys, lane);
    else bloomh_strip_mixed<SO>(L, lds, ring, cols, rows, z, xw, ys, lane);
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
  if (std::getenv("RC_DEBUG_BH") && ok) {
    std::vector<uint32_t> hc(colWords), hr(rowWords);
    (void)hipMemcpy(hc.data(), T->cols, colWords * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hr.data(), T->rows, rowWords * 4, hipMemcpyDeviceToHost);
    const int W = L.out_w, H = L.out_h;
    int cdiff = 0, rdiff = 0, wy0 = 0, wyhi = 0, y0eq = 0;
    std::map<std::vector<int>, int> pat;
    for (int i = 0; i < W; ++i) {
      bool d = false;
      std::vector<int> pv;
      for (int f = 0; f < BH_COL_FIELDS; ++f) {
        if (hc[(f * 2 + 0) * W + i] != hc[(f * 2 + 1) * W + i]) d = true;
        if (f < 9) pv.push_back((int)hc[(f * 2 + 0) * W + i]);
      }
      pv.push_back((int)hc[(BH_IDIM_X * 2) * W + i] - i);
      pv.push_back((int)hc[(BH_BRIGHT_X * 2) * W + i] - i);
      pat[pv]++;
      cdiff += d;
    }
    for (int i = 0; i < H; ++i) {
      bool d = false;
      for (int f = 0; f < 6; ++f) if (hr[((size_t)i * 2 + 0) * BH_ROW_FIELDS + f] != hr[((size_t)i * 2 + 1) * BH_ROW_FIELDS + f]) d = true;
      rdiff += d;
      const float wy = bits2f(hr[((size_t)i * 2) * BH_ROW_FIELDS + BH_WY]);
      wy0 += wy == 0.0f;
      wyhi += wy > 0.5f;
      y0eq += (int)hr[((size_t)i * 2) * BH_ROW_FIELDS + BH_Y0] == i;
    }
    std::fprintf(stderr, "[rc bloom-h dbg] cols differing between sides %d/%d, rows %d/%d; rows wy==0 %d, wy>0.5 %d, y0==y %d; patterns %zu\n", cdiff, W, rdiff, H, wy0, wyhi, y0eq, pat.size());
    for (auto& kv : pat) {
      std::fprintf(stderr, "   pattern x%d:", kv.second);
      for (int v : kv.first) std::fprintf(stderr, " %d", v);
      std::fprintf(stderr, "\n");
    }
    int hx = 0; for (int i = 1; i < W; ++i) hx += hc[(BH_HAL_X0 * 2) * W + i] != hc[(BH_HAL_X0 * 2) * W + i - 1];
    std::fprintf(stderr, "   hal x0 changes %d, hal texture %dx%d, params k %g %g %g %g dx %g\n", hx, L.extra[2].w, L.extra[2].h, L.params[RPG_K78], L.params[RPG_K56], L.params[RPG_K34], L.params[RPG_K12], L.params[RPG_DXY]);
  }
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
        const unsigned lds = (unsigned)(((256 + (int)kSrgbRuns + 3) & ~3) + kBhWaves * kBhRingDwords) * 4u;
        auto kernel = k_royale_bloom_h_strip<OutS>;
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
