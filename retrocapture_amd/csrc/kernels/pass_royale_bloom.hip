// crt-royale passes 9 and 10 (bloom-vertical.glsl, bloom-horizontal-reconstitute.glsl): the general per-pixel
// kernels, and the strip forms that run separable geometry (royale_strip.h) - the shipped preset at any size.
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstdlib>

#include "../rc_log.h"
#include "royale_bloom_h.h"

using namespace rcd;
using namespace rcroyale;
using namespace rcstrip;
using namespace rcstrip2;
using namespace rcbloomh;

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
  void release() {}
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

// ---- P9, strip form, two pixels per lane (royale_strip2.h): columns x and x + 64 of a 128-column band; the nine weighted
// taps of the pixel pair run as packed float operations.  A wave walks a run of consecutive rows of one band (equal runs of
// (frame, band, row) steps per wave) four target rows at a time with the 4 + 16 decoded source rows of the step in
// registers; the window stays where it is - five instantiations of the step address it rotated by four rows each, the four
// new rows, fetched a step ahead, are decoded over the four oldest (until round 4 sixty register moves per step shifted it):
// every source row is decoded once per run.
constexpr int kBv2Step = 4, kBv2Win = kBv2Step + 16, kBv2Waves = 12;
template <int PATTERN>
__global__ void __launch_bounds__(kBv2Waves * 64) k_royale_bloom_v_strip2(const PassLaunch L) {
  extern __shared__ uint32_t rc_dyn_lds_[];
  strip2_load_tables(rc_dyn_lds_, L, true);
  constexpr int o12 = 1 + ((PATTERN >> 0) & 1), o34 = 3 + ((PATTERN >> 1) & 1), o56 = 5 + ((PATTERN >> 2) & 1), o78 = 7 + ((PATTERN >> 3) & 1);
  const float* P = L.params;
  const float w78 = P[RPG_W78], w56 = P[RPG_W56], w34 = P[RPG_W34], w12 = P[RPG_W12], si = P[RPG_SUM_INV];
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int W = L.out_w, H = L.out_h, Hin = L.in.h, Win = L.in.w;
  const int bands = (W + 127) >> 7;
  const long total = (long)L.n_frames * bands * H;
  const long n_waves = (long)gridDim.x * kBv2Waves, me = (long)blockIdx.x * kBv2Waves + wave;
  long t = total * me / n_waves;
  const long t_end = total * (me + 1) / n_waves;
  while (t < t_end) {
    const int z = (int)(t / ((long)bands * H));
    const int rem = (int)(t - (long)z * bands * H);
    const int band = rem / H, y_first = rem - band * H;
    const int y_last = (int)min((long)H, (long)y_first + (t_end - t));   // exclusive
    t += y_last - y_first;
    const int xa = (band << 7) + lane, xb = xa + 64;
    const bool live_a = xa < W, live_b = xb < W;
    const uint32_t ca = (uint32_t)min(xa, Win - 1) * 4u, cb = (uint32_t)min(xb, Win - 1) * 4u;   // (k_bloomv_geometry: the taps' column is the pixel's own)
    const uint8_t* img = frame_ptr(L.in, z);
    const __amdgpu_buffer_rsrc_t r_out = frame_rsrc(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z, W, H);
    v2f win[kBv2Win][3];              // decoded source rows y0 - 8 .. y0 + 11 of the step at y0
    uint32_t na[kBv2Step], nb[kBv2Step];   // raw texels of the four rows that enter with the next step, in flight
    auto fetch4 = [&](int first) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < kBv2Step; ++i) {
        const uint8_t* p = img + (size_t)(clampi(first + i, 0, Hin - 1) * Win) * 4u;
        na[i] = *reinterpret_cast<const uint32_t*>(p + ca);
        nb[i] = *reinterpret_cast<const uint32_t*>(p + cb);
      }
    };
    // `black`: how many of the most recent four-row groups were black in every lane of the wave (decode(0) = 0).  With the whole
    // window black - five groups - the nine taps sum to 0 and the step stores black without the arithmetic (llvmpipe's default mode
    // hands this pass an all-black frame: pass 7's mask is zero there; otherwise letterbox bars).  The window itself is kept as
    // always (a black row decodes to zeros): the shortcut must not cost the ordinary path registers.
    int black = 0;
    auto decode4 = [&](int slot) __attribute__((always_inline)) {
      {
        uint32_t any = 0u;
#pragma unroll
        for (int i = 0; i < kBv2Step; ++i) any |= na[i] | nb[i];
        black = __builtin_amdgcn_ballot_w64((any & 0x00ffffffu) != 0u) == 0ull ? black + 1 : 0;
      }
#pragma unroll
      for (int i = 0; i < kBv2Step; ++i) {
        win[slot + i][0] = v2f{dec_byte<0>(na[i]), dec_byte<0>(nb[i])};
        win[slot + i][1] = v2f{dec_byte<1>(na[i]), dec_byte<1>(nb[i])};
        win[slot + i][2] = v2f{dec_byte<2>(na[i]), dec_byte<2>(nb[i])};
      }
    };
    // prime: rows y_first - 8 .. y_first + 7 into slots 0 .. 15
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      fetch4(y_first - 8 + 4 * g);
      decode4(4 * g);
    }
    fetch4(y_first + 8);
    // One step = four target rows.  The window does not move: step number n (mod 5) finds source row y0 - 8 + i in register slot
    // (i + 4 n) mod 20 and decodes its four new rows over the four oldest - five instantiations of the step, each with its
    // register indices as constants, instead of sixty register moves per step (a seventh of the kernel's instructions)
    auto step = [&](auto rot, int y0) __attribute__((always_inline)) {
      constexpr int R4 = 4 * decltype(rot)::value;
      decode4((kBv2Win - kBv2Step + R4) % kBv2Win);
      fetch4(y0 + kBv2Step + 8);
      if (black >= kBv2Win / kBv2Step) {   // (wave-uniform) the window holds zeros only: encode(0 * sum_inv) = 0
#pragma unroll
        for (int k = 0; k < kBv2Step; ++k) {
          const int y = y0 + k;
          if (y < y_last) {
            if (live_a) __builtin_amdgcn_raw_buffer_store_b32(0xff000000u, r_out, xa * 4, y * W * 4, 0);
            if (live_b) __builtin_amdgcn_raw_buffer_store_b32(0xff000000u, r_out, xb * 4, y * W * 4, 0);
          }
        }
        return;
      }
#pragma unroll
      for (int k = 0; k < kBv2Step; ++k) {
        const int y = y0 + k;
        if (y < y_last) {
          v2f o[3];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            // tex2Dblur17fast in the GL's evaluation order (blur17 above)
            v2f a = w78 * win[(k + 8 - o78 + R4) % kBv2Win][ch];
            a += w56 * win[(k + 8 - o56 + R4) % kBv2Win][ch];
            a += w34 * win[(k + 8 - o34 + R4) % kBv2Win][ch];
            a += win[(k + 8 + R4) % kBv2Win][ch];
            a += w12 * win[(k + 8 - o12 + R4) % kBv2Win][ch];
            a += w12 * win[(k + 8 + o12 + R4) % kBv2Win][ch];
            a += w34 * win[(k + 8 + o34 + R4) % kBv2Win][ch];
            a += w56 * win[(k + 8 + o56 + R4) % kBv2Win][ch];
            a += w78 * win[(k + 8 + o78 + R4) % kBv2Win][ch];
            o[ch] = a * si;
          }
          uint32_t pa, pb;
          srgb8_pack2(o, &pa, &pb);
          if (live_a) __builtin_amdgcn_raw_buffer_store_b32(pa, r_out, xa * 4, y * W * 4, 0);
          if (live_b) __builtin_amdgcn_raw_buffer_store_b32(pb, r_out, xb * 4, y * W * 4, 0);
        }
      }
    };
    static_assert(kBv2Win == 5 * kBv2Step, "five steps bring the window back to where it was");
    int y0 = y_first;
#pragma unroll 1
    while (true) {
      step(std::integral_constant<int, 0>(), y0);
      if ((y0 += kBv2Step) >= y_last) break;
      step(std::integral_constant<int, 1>(), y0);
      if ((y0 += kBv2Step) >= y_last) break;
      step(std::integral_constant<int, 2>(), y0);
      if ((y0 += kBv2Step) >= y_last) break;
      step(std::integral_constant<int, 3>(), y0);
      if ((y0 += kBv2Step) >= y_last) break;
      step(std::integral_constant<int, 4>(), y0);
      if ((y0 += kBv2Step) >= y_last) break;
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
  RC_LOG_DEBUG("crt-royale bloom-vertical " + std::to_string(L.out_w) + "x" + std::to_string(L.out_h) + ": flags " + std::to_string(hbad) + ", tap pattern " + std::to_string(T->pattern));
}

// ----------------------------------------------------------------- P10, strip form ------
// One wave renders a band of 128 columns, one target row per step, TWO pixels per lane: columns x and x + 64 of the row, so
// that every float operation of the filter runs as a packed v_pk_{fma,mul,add}_f32 over the pixel pair (the same IEEE
// operation per component).
//
// The nine blur taps of a pixel are LINEAR taps of one row (or row pair) of pass 9's sRGB8 target at horizontal offsets of
// up to 7.x texels: lerp(wx, T[j], T[j+1]) = fma(wx, T[j+1] - T[j], T[j]).  The difference D[j] = T[j+1] - T[j] does not
// depend on the tap, so it is formed ONCE per staged texel (the neighbour's texel arrives by a DPP wave shift folded into
// the subtraction) and a tap is one fma.  A source row is decoded and staged in LDS once: per staged column k (84 = 64 + 10
// either side, clamped like the sampler clamps them) and channel the quad {T_A[k], T_B[k], D_A[k], D_B[k]} of the band's two
// column groups A (columns xw - 10 + k) and B (64 further right), written by one conflict-free ds_write_b128 straight from
// the registers the decode leaves them in.  A tap whose first texel sits at the same offset for both of a lane's pixels -
// every tap but the centre one, whose coordinate lies on a texel centre up to rounding - is one ds_read_b128 per channel
// that delivers both pixels' operands in adjacent registers; the others read the two quads separately.  Per-column
// quantities (tap offsets as LDS addresses, weights as pairs) sit in registers for the whole band, row quantities are
// wave-uniform scalars, global accesses are buffer loads / stores with the row base in an SGPR.
// At 1:1 a target row's first source row is y or y - 1 (weight 0 / a few 1e-5 / 1 minus that, irregularly from row to row:
// k_bloomh_geometry): source rows are staged in order into a two-row ring as target rows first need them, their texels
// fetched two rows ahead.  Rows whose vertical weight is exactly 0 filter one source row.  The launch is cut into equal
// runs of consecutive (frame, band, row) steps, one per wave; rows are walked in blocks of 8, and a block the quad's
// diagonal crosses is rendered once per triangle, each pixel stored by the pass of its own triangle.
constexpr int kBhWaves = 12;
constexpr int kBhSeg = 84;        // staged columns per group: 10 + 64 + 10
constexpr int kBhSegLeft = 10;
constexpr int kBhColBytes = 48;                       // three channels x {T_A, T_B, D_A, D_B}
constexpr int kBhSlotBytes = kBhSeg * kBhColBytes;    // 4032: one staged source row
constexpr int kBhWaveDwords = 2 * kBhSlotBytes / 4;   // per wave: a ring of two staged rows (row r in slot r & 1)
struct BhTables {
  uint32_t* cols = nullptr;   // [BH_COL_FIELDS][2 sides][W] (ints and float bits)
  uint32_t* rows = nullptr;   // [H][2 sides][BH_ROW_FIELDS]
  bool usable = false;
  bool quad = false;          // the geometry also qualifies for the quad form (k_royale_bloom_h_quad), with
  int quad_taps = 0;          // bit 0 / 1: MASKED_SCANLINES / BRIGHTPASS is one texel per group of four columns (else the pixel's own column)
  uint32_t* qsteps = nullptr; // the quad form's steps of one frame pair, four words each (buildBqSteps), with their running cost
  int n_qsteps = 0;
  std::vector<uint32_t> qcost_sum;
  std::map<std::pair<int, int>, uint32_t*> qruns;
  // host: running sum of a cost estimate per (band, row) step of one frame (rows with a vertical weight filter two source rows,
  // blocks on the diagonal are rendered twice), and per (frames, waves) of a launch the step at which each wave's run begins
  std::vector<uint32_t> cost_sum;
  std::map<std::pair<int, int>, uint32_t*> runs;
  void release() {
    if (cols) (void)hipFree(cols);
    if (rows) (void)hipFree(rows);
    for (auto& r : runs)
      if (r.second) (void)hipFree(r.second);
    for (auto& r : qruns)
      if (r.second) (void)hipFree(r.second);
    if (qsteps) (void)hipFree(qsteps);
    *this = BhTables();
  }
};

__global__ void __launch_bounds__(256) k_bloomh_geometry(const PassLaunch L, uint32_t* cols, uint32_t* rows, uint32_t* bad, uint32_t* quad_bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int W = L.out_w, H = L.out_h;
  const float* P = L.params;
  const float k[4] = {P[RPG_K78], P[RPG_K56], P[RPG_K34], P[RPG_K12]};
  const float dx = P[RPG_DXY];
  uint32_t why = 0u, qwhy = 0u;
  // quad form: with the input staged clamped to the frame, tap q of column x may read the texel pair that starts at x + kBqTapOff[q]
  // (the centre tap at x - 1 or x, BH_CSEL) whenever that pair, clamped, is the sampler's own pair, clamped
  auto same_pair = [&](int i0, int j) { return clampi(i0, 0, L.in.w - 1) == clampi(j, 0, L.in.w - 1) && clampi(i0 + 1, 0, L.in.w - 1) == clampi(j + 1, 0, L.in.w - 1); };
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
        const int reg[9] = {-8, -6, -4, -2, 0, 1, 3, 5, 7};
        if (q != 4) {
          if (!same_pair(t.i0, i + reg[q])) qwhy |= 1u;
        } else {
          const bool left = same_pair(t.i0, i - 1), here = same_pair(t.i0, i);
          if (!left && !here) qwhy |= 1u;
          cols[(BH_CSEL * 2 + side) * W + i] = here ? 0u : 1u;
        }
      }
      cols[(BH_IDIM_X * 2 + side) * W + i] = (uint32_t)near_tap(vary(L.plane[2], i, 0, lo), L.extra[0].w);
      cols[(BH_BRIGHT_X * 2 + side) * W + i] = (uint32_t)near_tap(vary(L.plane[4], i, 0, lo), L.extra[1].w);
      const LinTap h = lin_tap(vary(L.plane[6], i, 0, lo), L.extra[2].w);
      cols[(BH_HAL_X0 * 2 + side) * W + i] = (uint32_t)h.i0;
      cols[(BH_HAL_W * 2 + side) * W + i] = f2bits(h.w);
      // quad form: the two single taps sit on the pixel's own column, and the four columns of an aligned group start their
      // halation pair at the group's first pair or the one after it
      // (bits 1, 2: MASKED_SCANLINES, BRIGHTPASS not on the pixel's own column; bits 3, 4: not on one texel for the whole group -
      // the reference hands the shader PassPrev3InputSize = the size pass 7 RECEIVED, 120 wide, so its tap magnifies 16 times)
      const int gi = i & ~3;
      const int n0 = near_tap(vary(L.plane[2], i, 0, lo), L.extra[0].w), n1 = near_tap(vary(L.plane[4], i, 0, lo), L.extra[1].w);
      if (n0 != i) qwhy |= 2u;
      if (n1 != i) qwhy |= 4u;
      if (n0 != near_tap(vary(L.plane[2], gi, 0, lo), L.extra[0].w)) qwhy |= 8u;
      if (n1 != near_tap(vary(L.plane[4], gi, 0, lo), L.extra[1].w)) qwhy |= 16u;
      const int hq = lin_tap(vary(L.plane[6], gi, 0, lo), L.extra[2].w).i0;
      if (h.i0 < hq || h.i0 > hq + 1) qwhy |= 32u;
    }
  }
  if (i < H) {
    for (int side = 0; side < 2; ++side) {
      const bool lo = side == 0;
      uint32_t* r = rows + ((size_t)i * 2 + side) * BH_ROW_FIELDS;
      const float v = vary(L.plane[1], 0, i, lo);
      const LinTap t = lin_tap(v - k[0] * 0.0f, L.in.h);   // every tap: v -+ k * 0 = v
      // source rows are staged in increasing order as target rows need them: the pair must start at y - 1 or y
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
  if (qwhy) atomicOr(quad_bad, qwhy);
}

// LDS accesses, the decode / encode tables and the packed helpers: royale_strip2.h
constexpr uint32_t kBhLdsUser = rcstrip2::kStrip2LdsUser;
__device__ __forceinline__ uint32_t bh_srgb8(float x) { return srgb8_lds(x); }

// One source row of this lane's staged column (its texel in group A and in group B) into its ring entry: decode, difference
// to the next staged column, one 16-byte store per channel.
__device__ __forceinline__ void bh_stage_entry(uint32_t entry, uint32_t ta, uint32_t tb) {
  {
    const float a = dec_byte<0>(ta), b = dec_byte<0>(tb);
    lds_put_v4f(entry, v4f{a, b, next_lane_dpp(a) - a, next_lane_dpp(b) - b});
  }
  {
    const float a = dec_byte<1>(ta), b = dec_byte<1>(tb);
    lds_put_v4f(entry + 16, v4f{a, b, next_lane_dpp(a) - a, next_lane_dpp(b) - b});
  }
  {
    const float a = dec_byte<2>(ta), b = dec_byte<2>(tb);
    lds_put_v4f(entry + 32, v4f{a, b, next_lane_dpp(a) - a, next_lane_dpp(b) - b});
  }
}

struct BhBand {   // per-band lane state of one triangle
  uint32_t tap_a[9];   // LDS offset of tap q's entry in ring slot 0, for the lane's pixel in group A
  uint32_t tap_d[2];   // ... and, packed 6 bits per tap, how many staged columns further right group B's pixel starts
                       // (0 where both pixels' taps start at the same offset: every tap but the centre one, away from the edges)
  uint32_t tap_b4;     // the centre tap's entry for the pixel in group B
  v2f wx[9];
};
__device__ __forceinline__ uint32_t bh_tap_b(const BhBand& c, int q) {
  if (q == 4) return c.tap_b4;
  uint32_t packed = c.tap_d[q / 5];
  asm volatile("" : "+v"(packed));   // edge bands only: unpack where it is used instead of holding eight more registers across the band
  const int d = ((int)(packed << (26 - 6 * (q % 5)))) >> 26;   // signed 6-bit field
  return c.tap_a[q] + (uint32_t)(d * kBhColBytes);
}

// the horizontal lerps of tap q from the ring slot at byte offset `slot`: both pixels, three channels.  PAIR: both pixels'
// taps start at the same staged column - one quad holds both pixels' operands.
template <bool PAIR>
__device__ __forceinline__ void bh_tap_row(const BhBand& c, int q, uint32_t slot, v2f* h) {
  const uint32_t pa = c.tap_a[q] + slot;
  if (PAIR) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const v4f e = lds_v4f(pa + 16 * ch);
      h[ch] = fma2(c.wx[q], v2f{e.z, e.w}, v2f{e.x, e.y});
    }
  } else {
    const uint32_t pb = bh_tap_b(c, q) + slot;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const v4f e = lds_v4f(pa + 16 * ch);
      const v4f f = lds_v4f(pb + 16 * ch);
      h[ch] = v2f{fma_(c.wx[q].x, e.z, e.x), fma_(c.wx[q].y, f.w, f.y)};
    }
  }
}
template <bool TWO, bool PAIR>
__device__ __forceinline__ void bh_tap(const BhBand& c, int q, uint32_t slot_a, uint32_t slot_b, float wy, v2f* h) {
  bh_tap_row<PAIR>(c, q, slot_a, h);
  if (TWO) {
    v2f g[3];
    bh_tap_row<PAIR>(c, q, slot_b, g);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) h[ch] = fma2(v2f{wy, wy}, g[ch] - h[ch], h[ch]);
  }
}

// The filter of one target row for the lane's two pixels: tex2Dblur17fast in the GL's evaluation order (blur17 above):
// taps 0 1 2, the centre (weight 1: a plain addend) before tap 3, then 5 .. 8.  TWO: the row has a vertical weight.
// EDGE: a band at the frame's edge, where clamped coordinates give the two pixels different tap offsets.
template <bool TWO, bool EDGE>
__device__ __forceinline__ void bh_filter(const BhBand& c, uint32_t slot_a, uint32_t slot_b, float wy, float w78, float w56, float w34, float w12,
                                          v2f* s) {
  v2f h[3];
  bh_tap<TWO, !EDGE>(c, 0, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] = w78 * h[ch];
  bh_tap<TWO, !EDGE>(c, 1, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += w56 * h[ch];
  bh_tap<TWO, !EDGE>(c, 2, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += w34 * h[ch];
  bh_tap<TWO, false>(c, 4, slot_a, slot_b, wy, h);   // the centre tap sits on a texel centre: its pair flips with rounding
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += h[ch];
  bh_tap<TWO, !EDGE>(c, 3, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += w12 * h[ch];
  bh_tap<TWO, !EDGE>(c, 5, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += w12 * h[ch];
  bh_tap<TWO, !EDGE>(c, 6, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += w34 * h[ch];
  bh_tap<TWO, !EDGE>(c, 7, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += w56 * h[ch];
  bh_tap<TWO, !EDGE>(c, 8, slot_a, slot_b, wy, h);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) s[ch] += w78 * h[ch];
}

__global__ void __launch_bounds__(kBhWaves * 64, 1) k_royale_bloom_h_strip(const PassLaunch L, const uint32_t* __restrict__ cols,
                                                                          const uint32_t* __restrict__ rows, const uint32_t* __restrict__ runs) {
  extern __shared__ uint32_t rc_dyn_lds_[];
  strip2_load_tables(rc_dyn_lds_, L, true);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t ring = kBhLdsUser + (uint32_t)(wave * kBhWaveDwords) * 4u;   // LDS offset of this wave's two row slots
  const int W = L.out_w, H = L.out_h, Win = L.in.w, Hin = L.in.h;
  const float* P = L.params;
  const float w78 = P[RPG_W78], w56 = P[RPG_W56], w34 = P[RPG_W34], w12 = P[RPG_W12], si = P[RPG_SUM_INV];
  const float c_main = (P[RPG_MASK_AMPLIFY] * 2.0f) * (1.0f - 0.075f);
  const int i0w = L.extra[0].w, i1w = L.extra[1].w, hw = L.extra[2].w, hh = L.extra[2].h;
  // this wave's run of consecutive (frame, band, row) steps: runs of equal estimated cost (launch_royale_bloom_h)
  const int bands = (W + 127) >> 7;
  const int me = (int)blockIdx.x * kBhWaves + wave;
  long t = runs[me];
  const long t_end = runs[me + 1];
  const uint32_t e_main = ring + (uint32_t)lane * kBhColBytes;   // this lane's ring entries (slot 0): staged column `lane` ...
  const uint32_t e_extra = e_main + 63u * kBhColBytes;           // ... and, lanes 0..20, staged column 63 + lane
  while (t < t_end) {
    const int z = (int)(t / ((long)bands * H));
    const int rem = (int)(t - (long)z * bands * H);
    const int band = rem / H, y_first = rem - band * H;
    const int y_last = (int)min((long)H, (long)y_first + (t_end - t));   // exclusive
    t += y_last - y_first;
    const int xw = band << 7;
    const int xa = xw + lane, xb = xa + 64;
    const bool live_a = xa < W, live_b = xb < W;
    const int xca = live_a ? xa : W - 1, xcb = live_b ? xb : W - 1;
    const int xmax = min(xw + 127, W - 1);
    // frame bases are wave-uniform, a lane's column offset is fixed for the band, the row base is a scalar: buffer accesses
    // for the filtered input and the target, scalar-base global loads for the single taps
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(frame_ptr(L.in, z)), 0, Win * Hin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_out =
        __builtin_amdgcn_make_buffer_rsrc(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z, 0, W * H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rows = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(rows), 0, H * 2 * BH_ROW_FIELDS * 4, 0x00020000);
    const uint8_t* i0_base = frame_ptr(L.extra[0], z);
    const uint8_t* i1_base = frame_ptr(L.extra[1], z);
    const uint8_t* hal_base = frame_ptr(L.extra[2], z);
    // staged columns: lane l holds staged column l (group A: source column xw - 10 + l, group B: 64 further right) and, for
    // l < 21, staged column 63 + l; column 63 is staged twice: its main entry's differences are wrong (the DPP shift has
    // no lane 64) and the extra batch, stored second, overwrites it
    const int sxa0 = clampi(xw - kBhSegLeft + lane, 0, Win - 1) * 4, sxb0 = clampi(xw - kBhSegLeft + 64 + lane, 0, Win - 1) * 4;
    const int sxa1 = clampi(xw - kBhSegLeft + 63 + lane, 0, Win - 1) * 4, sxb1 = clampi(xw - kBhSegLeft + 127 + lane, 0, Win - 1) * 4;
    auto fetch = [&](int r, uint32_t* q) __attribute__((always_inline)) {
      const int ro = clampi(r, 0, Hin - 1) * Win * 4;
      q[0] = __builtin_amdgcn_raw_buffer_load_b32(r_in, sxa0, ro, 0);
      q[1] = __builtin_amdgcn_raw_buffer_load_b32(r_in, sxb0, ro, 0);
      q[2] = __builtin_amdgcn_raw_buffer_load_b32(r_in, sxa1, ro, 0);
      q[3] = __builtin_amdgcn_raw_buffer_load_b32(r_in, sxb1, ro, 0);
    };
    // what of the source image is staged / in flight: rows are staged in increasing order into ring slot (row & 1)
    int st_hi = -1000;           // highest source row in the ring
    uint32_t q0[4], q1[4];       // raw texels (main A, main B, extra A, extra B) of rows st_q, st_q + 1, in flight
    int st_q = -1000;
    // the column quantities of the lane's two pixels for one triangle (`have_side`), reloaded when the triangle changes
    BhBand c;
    bool edge = false;
    uint32_t idim_xa = 0u, idim_xb = 0u, bright_xa = 0u, bright_xb = 0u;
    uint32_t hal_a0 = 0u, hal_a1 = 0u, hal_b0 = 0u, hal_b1 = 0u;   // byte offsets of the clamped halation texel pair, per pixel
    v2f hal_w = {0.f, 0.f};
    int have_side = -1;
    // halation (320 x 240, magnified): the horizontally filtered rows hbase, hbase + 1 of the current row pair stay in
    // registers; the texels of row hbase + 2 are fetched ahead
    auto hal_fetch = [&](int r, uint32_t* q) __attribute__((always_inline)) {
      const uint8_t* p = hal_base + (size_t)(clampi(r, 0, hh - 1) * hw) * 4u;
      q[0] = *reinterpret_cast<const uint32_t*>(p + hal_a0);
      q[1] = *reinterpret_cast<const uint32_t*>(p + hal_a1);
      q[2] = *reinterpret_cast<const uint32_t*>(p + hal_b0);
      q[3] = *reinterpret_cast<const uint32_t*>(p + hal_b1);
    };
    auto hal_filter = [&](const uint32_t* q, v2f* h) __attribute__((always_inline)) {   // the sampler's horizontal lerp, both pixels
      const v2f l0 = {dec_byte<0>(q[0]), dec_byte<0>(q[2])}, r0 = {dec_byte<0>(q[1]), dec_byte<0>(q[3])};
      const v2f l1 = {dec_byte<1>(q[0]), dec_byte<1>(q[2])}, r1 = {dec_byte<1>(q[1]), dec_byte<1>(q[3])};
      const v2f l2 = {dec_byte<2>(q[0]), dec_byte<2>(q[2])}, r2 = {dec_byte<2>(q[1]), dec_byte<2>(q[3])};
      h[0] = fma2(hal_w, r0 - l0, l0);
      h[1] = fma2(hal_w, r1 - l1, l1);
      h[2] = fma2(hal_w, r2 - l2, l2);
    };
    v2f hl0[3], hl1[3], hld[3];   // rows hbase, hbase + 1 and their difference
    uint32_t hq[4];
    int hbase = -1000, hq_row = -1000;
    // walk the rows in blocks that do not straddle a multiple of kBhBlockRows
    for (int yb = y_first; yb < y_last;) {
      const int ye = min(y_last, (yb / kBhBlockRows + 1) * kBhBlockRows);
      // the lower triangle holds the pixels with (2y+1) W <= (2x+1) H: the block is all lower if its (min x, max y)
      // pixel is, all upper if its (max x, min y) pixel is not
      const bool all_lo = rcd::lower_tri(xw, ye - 1, W, H), all_up = !rcd::lower_tri(xmax, yb, W, H);
      for (int side = all_up ? 1 : 0; side <= (all_lo ? 0 : 1); ++side) {
        const bool mixed = !all_lo && !all_up;
        if (side != have_side) {
          have_side = side;
          hbase = hq_row = -1000;   // the halation rows in registers were filtered with the other triangle's columns
          c.tap_d[0] = c.tap_d[1] = 0u;
          edge = false;
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const int da = (int)cols[((BH_DX + q) * 2 + side) * W + xca] + (xca - xa), db = (int)cols[((BH_DX + q) * 2 + side) * W + xcb] + (xcb - xb);
            c.tap_a[q] = ring + (uint32_t)(lane + kBhSegLeft + da) * (uint32_t)kBhColBytes;
            c.tap_d[q / 5] |= ((uint32_t)(db - da) & 63u) << (6 * (q % 5));   // |db - da| <= 19 (k_bloomh_geometry)
            c.wx[q] = v2f{bits2f(cols[((BH_WX + q) * 2 + side) * W + xca]), bits2f(cols[((BH_WX + q) * 2 + side) * W + xcb])};
            if (q == 4) c.tap_b4 = ring + (uint32_t)(lane + kBhSegLeft + db) * (uint32_t)kBhColBytes;
            if (q != 4 && __builtin_amdgcn_ballot_w64(da != db) != 0ull) edge = true;
          }
          idim_xa = cols[(BH_IDIM_X * 2 + side) * W + xca] * 4u;
          idim_xb = cols[(BH_IDIM_X * 2 + side) * W + xcb] * 4u;
          bright_xa = cols[(BH_BRIGHT_X * 2 + side) * W + xca] * 4u;
          bright_xb = cols[(BH_BRIGHT_X * 2 + side) * W + xcb] * 4u;
          const int ha = (int)cols[(BH_HAL_X0 * 2 + side) * W + xca], hb = (int)cols[(BH_HAL_X0 * 2 + side) * W + xcb];
          hal_a0 = (uint32_t)clampi(ha, 0, hw - 1) * 4u;
          hal_a1 = (uint32_t)clampi(ha + 1, 0, hw - 1) * 4u;
          hal_b0 = (uint32_t)clampi(hb, 0, hw - 1) * 4u;
          hal_b1 = (uint32_t)clampi(hb + 1, 0, hw - 1) * 4u;
          hal_w = v2f{bits2f(cols[(BH_HAL_W * 2 + side) * W + xca]), bits2f(cols[(BH_HAL_W * 2 + side) * W + xcb])};
        }
        if (mixed && side == 1) st_hi = -1000;   // the block's rows once more: source rows from the top of the block again
        BhRowRaw nxt_raw = fetch_bh_row(r_rows, yb, side);
        BhRow nxt = use_bh_row(nxt_raw);
        // the two NEAREST taps of a row (MASKED_SCANLINES, BRIGHTPASS) are fetched one step ahead, like the row quantities
        uint32_t nia, nib, nja, njb;
        auto near_fetch = [&](const BhRow& r) __attribute__((always_inline)) {
          const uint8_t* i0_row = i0_base + (size_t)(r.idim_y * i0w) * 4u;
          const uint8_t* i1_row = i1_base + (size_t)(r.bright_y * i1w) * 4u;
          nia = *reinterpret_cast<const uint32_t*>(i0_row + idim_xa);
          nib = *reinterpret_cast<const uint32_t*>(i0_row + idim_xb);
          nja = *reinterpret_cast<const uint32_t*>(i1_row + bright_xa);
          njb = *reinterpret_cast<const uint32_t*>(i1_row + bright_xb);
        };
        near_fetch(nxt);
        nxt_raw = fetch_bh_row(r_rows, min(yb + 1, H - 1), side);
        // mixed blocks: a pixel is stored by the pass of its own triangle, (2y+1) W <= (2x+1) H tells which
        const int tri_a = (2 * xa + 1) * H, tri_b = (2 * xb + 1) * H;
#pragma unroll 1
        for (int y = yb; y < ye; ++y) {
          const BhRow ra = nxt;
          nxt = use_bh_row(nxt_raw);                                      // row y + 1, fetched during the previous step
          nxt_raw = fetch_bh_row(r_rows, min(y + 2, H - 1), side);        // row y + 2
          const uint32_t ia = nia, ib = nib, ja = nja, jb = njb;
          const bool two = ra.wy != 0.0f;
          const int row_a = clampi(ra.y0, 0, Hin - 1), row_b = clampi(ra.y0 + 1, 0, Hin - 1);
          const int need = two ? row_b : row_a;
          // ---- stage the source rows this target row needs and the ring does not hold yet (normally one).  Other lanes read
          // what a lane writes here: the LDS executes a wave's operations in order, the compiler must not reorder them
          if (st_hi < row_a - 1 || st_hi > need + 1) st_hi = row_a - 1;
          asm volatile("" ::: "memory");
          while (st_hi < need) {
            const int r = st_hi + 1;
            if (st_q != r) {   // queue stale (start of a run, a jump): refill
              fetch(r, q0);
              fetch(r + 1, q1);
              st_q = r;
            }
            const uint32_t slot = (r & 1) ? (uint32_t)kBhSlotBytes : 0u;
            bh_stage_entry(e_main + slot, q0[0], q0[1]);
            if (lane < kBhSeg - 63) bh_stage_entry(e_extra + slot, q0[2], q0[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) q0[i] = q1[i];
            fetch(r + 2, q1);
            st_q = r + 1;
            st_hi = r;
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          near_fetch(nxt);   // (issued behind the staging, so that its waits do not cover these loads)
          // ---- halation rows ra.hal_y0, + 1
          if (ra.hal_y0 != hbase) {
            if (ra.hal_y0 == hbase + 1) {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) hl0[ch] = hl1[ch];
            } else {
              uint32_t q[4];
              hal_fetch(ra.hal_y0, q);
              hal_filter(q, hl0);
            }
            if (hq_row != ra.hal_y0 + 1) hal_fetch(ra.hal_y0 + 1, hq);
            hal_filter(hq, hl1);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) hld[ch] = hl1[ch] - hl0[ch];
            hbase = ra.hal_y0;
            hq_row = hbase + 2;
            hal_fetch(hq_row, hq);
          }
          // ---- filter
          const uint32_t slot_a = (row_a & 1) ? (uint32_t)kBhSlotBytes : 0u, slot_b = (row_b & 1) ? (uint32_t)kBhSlotBytes : 0u;
          v2f s[3];
          if (!edge) {
            if (two) bh_filter<true, false>(c, slot_a, slot_b, ra.wy, w78, w56, w34, w12, s);
            else bh_filter<false, false>(c, slot_a, slot_a, 0.0f, w78, w56, w34, w12, s);
          } else {
            if (two) bh_filter<true, true>(c, slot_a, slot_b, ra.wy, w78, w56, w34, w12, s);
            else bh_filter<false, true>(c, slot_a, slot_a, 0.0f, w78, w56, w34, w12, s);
          }
          // ---- reconstitute (as k_royale_bloom_h) and store
          uint32_t oa = 0xff000000u, ob = 0xff000000u;
          {
            const v2f bl = s[0] * si;
            const v2f dimpass = v2f{dec_byte<0>(ia), dec_byte<0>(ib)} - v2f{dec_byte<0>(ja), dec_byte<0>(jb)};
            const v2f hal = fma2(v2f{ra.hal_wy, ra.hal_wy}, hld[0], hl0[0]);
            const v2f o = (dimpass + bl) * c_main + hal * 0.075f;
            oa |= bh_srgb8(o.x);
            ob |= bh_srgb8(o.y);
          }
          {
            const v2f bl = s[1] * si;
            const v2f dimpass = v2f{dec_byte<1>(ia), dec_byte<1>(ib)} - v2f{dec_byte<1>(ja), dec_byte<1>(jb)};
            const v2f hal = fma2(v2f{ra.hal_wy, ra.hal_wy}, hld[1], hl0[1]);
            const v2f o = (dimpass + bl) * c_main + hal * 0.075f;
            oa |= bh_srgb8(o.x) << 8;
            ob |= bh_srgb8(o.y) << 8;
          }
          {
            const v2f bl = s[2] * si;
            const v2f dimpass = v2f{dec_byte<2>(ia), dec_byte<2>(ib)} - v2f{dec_byte<2>(ja), dec_byte<2>(jb)};
            const v2f hal = fma2(v2f{ra.hal_wy, ra.hal_wy}, hld[2], hl0[2]);
            const v2f o = (dimpass + bl) * c_main + hal * 0.075f;
            oa |= bh_srgb8(o.x) << 16;
            ob |= bh_srgb8(o.y) << 16;
          }
          asm volatile("" ::: "memory");
          bool sa = live_a, sb = live_b;
          if (mixed) {
            const int tri_y = (2 * y + 1) * W;
            sa = sa && (tri_y <= tri_a) == (side == 0);
            sb = sb && (tri_y <= tri_b) == (side == 0);
          }
          // Everything fetched ahead for the next step is waited for HERE, before this step's stores are issued: vmcnt counts
          // loads and stores in one in-order queue, so a wait placed behind the stores (where the compiler puts the loop's
          // register copies) would also wait for stores issued a few cycles earlier - a full memory round trip per step.
          asm volatile("" : "+v"(q1[0]), "+v"(q1[1]), "+v"(q1[2]), "+v"(q1[3]), "+v"(nia), "+v"(nib), "+v"(nja), "+v"(njb));
          asm volatile("" : "+v"(nxt_raw.a), "+v"(nxt_raw.b), "+v"(hq[0]), "+v"(hq[1]), "+v"(hq[2]), "+v"(hq[3]));
          if (sa) __builtin_amdgcn_raw_buffer_store_b32(oa, r_out, xa * 4, y * W * 4, 0);
          if (sb) __builtin_amdgcn_raw_buffer_store_b32(ob, r_out, xb * 4, y * W * 4, 0);
        }
      }
      yb = ye;
    }
  }
}

bool buildBqSteps(const PassLaunch& L, const std::vector<uint32_t>& hr, BhTables* T);
void buildBhTables(const PassLaunch& L, hipStream_t s, BhTables* T) {
  uint32_t* bad = nullptr;   // two words: the strip forms' flags, the quad form's
  const size_t colWords = (size_t)BH_COL_FIELDS * 2 * L.out_w, rowWords = (size_t)L.out_h * 2 * BH_ROW_FIELDS;
  bool ok = hipMalloc(reinterpret_cast<void**>(&T->cols), colWords * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T->rows), rowWords * 4) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&bad), 8) == hipSuccess;
  uint32_t hbad[2] = {1, 1};
  if (ok) ok = hipMemsetAsync(bad, 0, 8, s) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_bloomh_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, T->cols, T->rows, bad, bad + 1);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(hbad, bad, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (bad) (void)hipFree(bad);
  T->usable = ok && hbad[0] == 0;
  // the quad form additionally wants a 1:1 pass over whole groups of four columns, both frames of a pair inside 32-bit offsets,
  // and each single tap either on the pixel's own column or on one texel per group
  const uint32_t q = hbad[1];
  T->quad = T->usable && (q & (1u | 32u)) == 0 && (!(q & 2u) || !(q & 8u)) && (!(q & 4u) || !(q & 16u)) && bqGeometryOk(L, !(q & 2u), !(q & 4u));
  T->quad_taps = ((q & 2u) ? 1 : 0) | ((q & 4u) ? 2 : 0);
  if (T->usable) {
    // cost estimate per step, as the kernel will walk it: blocks of kBhBlockRows rows, per triangle the block touches;
    // a row with a vertical weight filters two source rows (measured: about 1.45 x the work)
    std::vector<uint32_t> hr(rowWords);
    T->usable = hipMemcpy(hr.data(), T->rows, rowWords * 4, hipMemcpyDeviceToHost) == hipSuccess;
    const int W = L.out_w, H = L.out_h, bands = (W + 127) / 128;
    auto lower = [&](int x, int y) { return (long)(2 * y + 1) * W <= (long)(2 * x + 1) * H; };
    T->cost_sum.assign((size_t)bands * H + 1, 0u);
    for (int b = 0; b < bands; ++b)
      for (int y = 0; y < H; ++y) {
        const int yb = y / kBhBlockRows * kBhBlockRows, ye = std::min(H, yb + kBhBlockRows);
        const int xw = b * 128, xmax = std::min(xw + 127, W - 1);
        const bool all_lo = lower(xw, ye - 1), all_up = !lower(xmax, yb);
        uint32_t cost = 0;
        for (int side = all_up ? 1 : 0; side <= (all_lo ? 0 : 1); ++side)
          cost += bits2f(hr[((size_t)y * 2 + side) * BH_ROW_FIELDS + BH_WY]) != 0.0f ? 29u : 20u;
        T->cost_sum[(size_t)b * H + y + 1] = T->cost_sum[(size_t)b * H + y] + cost;
      }
    if (T->usable && T->quad && !buildBqSteps(L, hr, T)) {
      if (T->qsteps) (void)hipFree(T->qsteps);
      T->qsteps = nullptr;
      T->quad = false;
    }
  }
  RC_LOG_DEBUG("crt-royale bloom-horizontal " + std::to_string(L.out_w) + "x" + std::to_string(L.out_h) + ": strip flags " + std::to_string(hbad[0]) + ", quad flags " +
               std::to_string(hbad[1]) + (T->quad ? ": quad form" : (T->usable ? ": strip form" : ": general form")));
  if (!T->usable) {
    if (T->cols) (void)hipFree(T->cols);
    if (T->rows) (void)hipFree(T->rows);
    *T = BhTables();
  }
}

// Where each wave's run of steps begins for a launch of n_frames frames (frame pairs, for the quad form) on n_waves waves: runs of
// equal estimated cost (n_waves + 1 entries in device memory, built once per (n_frames, n_waves) and kept with the tables).
const uint32_t* costRuns(const std::vector<uint32_t>& cost_sum, std::map<std::pair<int, int>, uint32_t*>& cache, int n_frames, int n_waves) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find({n_frames, n_waves});
  if (it != cache.end()) return it->second;
  const size_t per_frame = cost_sum.size() - 1;
  const uint64_t frame_cost = cost_sum.back(), total = frame_cost * (uint64_t)n_frames;
  std::vector<uint32_t> h((size_t)n_waves + 1);
  for (int w = 0; w <= n_waves; ++w) {
    const uint64_t target = total * (uint64_t)w / (uint64_t)n_waves;
    const uint64_t z = std::min<uint64_t>(target / frame_cost, (uint64_t)n_frames), rem = target - z * frame_cost;
    // first step of frame z whose running cost reaches `rem`
    const size_t i = (size_t)(std::lower_bound(cost_sum.begin(), cost_sum.end(), (uint32_t)rem) - cost_sum.begin());
    h[(size_t)w] = (uint32_t)(z * per_frame + std::min(i, per_frame));
  }
  h[(size_t)n_waves] = (uint32_t)(per_frame * (size_t)n_frames);
  uint32_t* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), h.size() * 4) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return nullptr;
  }
  cache[{n_frames, n_waves}] = d;
  return d;
}
const uint32_t* bhRuns(BhTables* T, int n_frames, int n_waves) { return costRuns(T->cost_sum, T->runs, n_frames, n_waves); }

#ifndef RC_BQ_COST_TWO
#define RC_BQ_COST_TWO 26
#endif
// The quad form's step list of one frame pair (pass_royale_bloom_quad.hip): bands in order, rows in blocks of kBhBlockRows, a block
// the diagonal crosses once per triangle; per step its row quantities, whether a (band, triangle) segment begins, and the
// running cost (a row with a vertical weight filters two source rows).  `hr`: the rows table on the host.
bool buildBqSteps(const PassLaunch& L, const std::vector<uint32_t>& hr, BhTables* T) {
  const int W = L.out_w, H = L.out_h, bands = (W + 127) / 128;
  if (H > 4095 || L.extra[2].h > 1021 || bands > 64 || L.extra[0].h > 65535 || L.extra[1].h > 65535) return false;
  auto lower = [&](int x, int y) { return (long)(2 * y + 1) * W <= (long)(2 * x + 1) * H; };
  std::vector<uint32_t> st;
  T->qcost_sum.assign(1, 0u);
  for (int b = 0; b < bands; ++b) {
    int last_side = -1;
    for (int yb = 0; yb < H; yb += kBhBlockRows) {
      const int ye = std::min(H, yb + kBhBlockRows), xw = b * 128, xmax = std::min(xw + 127, W - 1);
      const bool all_lo = lower(xw, ye - 1), all_up = !lower(xmax, yb), mixed = !all_lo && !all_up;
      for (int side = all_up ? 1 : 0; side <= (all_lo ? 0 : 1); ++side)
        for (int y = yb; y < ye; ++y) {
          const uint32_t* r = &hr[((size_t)y * 2 + side) * BH_ROW_FIELDS];
          const int y0 = (int)r[BH_Y0], hal_y0 = (int)r[BH_HAL_Y0];
          if (y0 < y - 1 || y0 > y || hal_y0 < -1) return false;
          // a segment begins with the band, when the triangle changes, and at the second rendering of a block (its source rows once more)
          const bool newseg = side != last_side || (mixed && y == yb);
          last_side = side;
          st.push_back((uint32_t)y | ((uint32_t)(hal_y0 + 1) << 12) | ((uint32_t)side << 22) | ((mixed ? 1u : 0u) << 23) | ((newseg ? 1u : 0u) << 24) |
                       ((y0 == y - 1 ? 1u : 0u) << 25) | ((uint32_t)b << 26));
          st.push_back(r[BH_IDIM_Y] | (r[BH_BRIGHT_Y] << 16));
          st.push_back(r[BH_WY]);
          st.push_back(r[BH_HAL_WY]);
          T->qcost_sum.push_back(T->qcost_sum.back() + (bits2f(r[BH_WY]) != 0.0f ? (uint32_t)RC_BQ_COST_TWO : 20u));
        }
    }
  }
  T->n_qsteps = (int)(st.size() / 4);
  return hipMalloc(reinterpret_cast<void**>(&T->qsteps), st.size() * 4) == hipSuccess &&
         hipMemcpy(T->qsteps, st.data(), st.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
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
  // one run of rows per wave, one workgroup of twelve waves per CU (the window's 154 registers allow three waves per SIMD):
  // 8 frames of 1080p make runs of 42 rows, 16 more being the window's lead-in
  const long steps = (long)((L.out_w + 127) / 128) * L.out_h * L.n_frames;
  const long blocks = std::max<long>(1, std::min<long>(256, steps / (kBv2Waves * 16)));
  hipLaunchKernelGGL((k_royale_bloom_v_strip2<PATTERN>), dim3((unsigned)blocks), dim3(kBv2Waves * 64), rcstrip2::kStrip2LdsUser, s, L);
  return hipGetLastError();
}
hipError_t launch_royale_bloom_v(const PassLaunch& L, hipStream_t s) {
  if (SrgbNearEdge::matches(L.in) && OutS::matches(L)) {
    if (!(L.flags & RC_FLAG_GENERAL_ONLY) && separable(L, 0, 1) && L.in.w == L.out_w) {
      static std::mutex mu;
      static std::map<GeoKey, GeoCached<BvTables>> cache;
      if (const auto T = geo_tables<BvTables>(L, s, mu, cache, buildBvTables)) {
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
      static std::map<GeoKey, GeoCached<BhTables>> cache;
      if (const auto T = geo_tables<BhTables>(L, s, mu, cache, buildBhTables)) {
        // one workgroup per CU; every wave gets a run of (frame, band, row) steps of equal estimated cost
        const long steps = (long)((L.out_w + 127) / 128) * L.out_h * L.n_frames;
        const long blocks = std::min<long>((steps + kBhWaves * 8 - 1) / (kBhWaves * 8), 256);
        const uint64_t max_stride = std::max(std::max(L.in.frame_stride, L.out_frame_stride), std::max(std::max(L.extra[0].frame_stride, L.extra[1].frame_stride), L.extra[2].frame_stride));
        if (T->quad && max_stride * (uint64_t)L.n_frames < (1ull << 31)) {
          // two frames per wave: runs of (frame pair, band, triangle, row) steps
          const int n_pairs = (L.n_frames + 1) / 2;
          const uint64_t qsteps = (uint64_t)T->n_qsteps * (uint64_t)n_pairs;
          const long qblocks = (long)std::min<uint64_t>((qsteps + (uint64_t)bq_waves() * 8 - 1) / ((uint64_t)bq_waves() * 8), 256);
          BhTables* Tm = const_cast<BhTables*>(T.get());
          const uint32_t* qruns = qsteps < (1ull << 32) ? costRuns(Tm->qcost_sum, Tm->qruns, n_pairs, (int)qblocks * bq_waves()) : nullptr;
          if (qruns) return launch_bloom_h_quad(L, s, T->cols, T->qsteps, T->n_qsteps, qruns, (unsigned)qblocks, T->quad_taps);
        }
        const uint32_t* runs = (uint64_t)steps < (1ull << 32) ? bhRuns(const_cast<BhTables*>(T.get()), L.n_frames, (int)blocks * kBhWaves) : nullptr;
        if (!runs) GO(k_royale_bloom_h<SrgbLinEdge, SrgbNearEdge, SrgbNearEdge, SrgbLinEdge, OutS>);
        const unsigned lds = kBhLdsUser + (unsigned)(kBhWaves * kBhWaveDwords) * 4u;
        auto kernel = k_royale_bloom_h_strip;
        // (set on every launch: the attribute is per device, and this needs no shared flag)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return hipGetLastError();
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBhWaves * 64), lds, s, L, T->cols, T->rows, runs);
        return hipGetLastError();
      }
    }
    GO(k_royale_bloom_h<SrgbLinEdge, SrgbNearEdge, SrgbNearEdge, SrgbLinEdge, OutS>);
  }
  GO(k_royale_bloom_h<SRT, SRT, SRT, SRT, StRT>);
}
}  // namespace rck
