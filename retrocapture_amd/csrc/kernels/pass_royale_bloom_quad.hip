// crt-royale pass 10 (bloom-horizontal-reconstitute.glsl), quad form: four adjacent columns per lane, the filter window in
// registers.  Compiled with -fno-slp-vectorize (Makefile / CMakeLists.txt): the SLP vectoriser pairs the scalar float
// operations of neighbouring pixels into v_pk_* instructions over register pairs it then has to assemble with moves, and the
// kernel spills (a packed operation costs what two scalar ones with VGPR operands do).
#include "royale_bloom_h.h"

using namespace rcd;
using namespace rcroyale;
using namespace rcstrip;
using namespace rcstrip2;
using namespace rcbloomh;

namespace {
__device__ __forceinline__ uint32_t bh_srgb8(float x) { return srgb8_lds(x); }

// ------------------------------------------------------------------ P10, quad form ------
// The strip form above makes every tap of the horizontal filter a cross-lane read: a lane owns one column (pair), its nine
// taps' texels belong to other lanes and come through LDS, 27 quad reads per pixel pair and source row.  Here a lane owns FOUR
// ADJACENT COLUMNS of one row: the 4 + 16 texels its taps touch are read from the staged row once (15 ds_read_b128 for
// twenty texels of three channels), the differences T[j+1] - T[j] are formed in registers and every tap is one fma on
// registers.  At 1:1 the taps of column x start at texels x - 8, x - 6, x - 4, x - 2, (x - 1 or x), x + 1, x + 3, x + 5, x + 7
// for ANY blur sigma (the pair offsets 1 + w2 / w12 ... lie in (1, 1.5), (3, 3.5), ...), the centre tap - a coordinate on a
// texel centre up to rounding - on either side per column (BH_CSEL); at the frame's edge the sampler's clamped pair and
// the pair at the regular offset of a row staged with clamped columns are the same two texels (k_bloomh_geometry checks
// both, per column and triangle).  The per-column weights are per-lane registers for a whole band; row quantities are
// wave-uniform as in the strip form, from which the walk is taken over: runs of consecutive (frame pair, band, row) steps of
// equal estimated cost per wave, source rows staged in order into a two-row ring in LDS as target rows first need them,
// blocks of 8 rows rendered per triangle.  A wave spans 128 columns (32 lanes) of TWO frames - same band, same row, so
// every wave-uniform quantity is shared and only the frame offset is per lane - which keeps the band as narrow as the strip
// form's (the share of blocks the diagonal crosses grows with the band) with all 64 lanes busy.
// (Replicating the sRGB decode table - lane l reading copy l & 15 or l & 31 of an entry-major table, which makes the gathers
// of 64 random bytes conflict-free - was built and measured: 13.5 us per frame against 12.7 with the one table; dropped.)
#ifndef RC_BQ_WAVES
#define RC_BQ_WAVES 12
#endif
constexpr int kBqWaves = RC_BQ_WAVES;
constexpr int kBqBand = 128;                          // columns per band
constexpr int kBqHalo = 8;
constexpr int kBqSeg = kBqBand + 2 * kBqHalo;         // staged columns per frame slot: 36 groups of four
constexpr int kBqSlotBytes = kBqSeg * 12;             // three decoded channels per staged texel: 1728
constexpr int kBqRowBytes = 2 * kBqSlotBytes;         // both frame slots of one source row
constexpr int kBqWaveBytes = 2 * kBqRowBytes;         // ring of two source rows (row r in slot r & 1)
constexpr uint32_t kBqLdsRing = rcstrip2::kStrip2LdsUser;
constexpr uint32_t kBqLdsBytes = kBqLdsRing + (uint32_t)(kBqWaves * kBqWaveBytes);
static_assert(kBqLdsBytes <= 160u * 1024u, "quad form LDS");

// byte N of texel t, decoded (the table at LDS offset 0)
template <int N>
__device__ __forceinline__ float bq_dec(uint32_t t, uint32_t) { return dec_byte<N>(t); }

struct BqCols {        // per-band lane state of one triangle: the lane's four columns
  float w[4][9];       // horizontal weight of tap q
  uint32_t left;       // bit i: the centre tap's pair of column i starts at x - 1
  float hw[4];         // halation: horizontal weight, and (bit i of hright) whether column i's pair starts one texel right of the
  uint32_t hright;     // group's first pair
};

// texel t (0 .. 19: column x0 - 8 + t), channel ch of a window read as 15 quads
#define RC_BQ_T(Q, t, ch) (Q[(3 * (t) + (ch)) >> 2][(3 * (t) + (ch)) & 3])

// The filter of one target row for the lane's four pixels: tex2Dblur17fast in the GL's evaluation order (blur17 above): taps
// 0 1 2, the centre (weight 1: a plain addend) before tap 3, then 5 .. 8.  TWO: the row has a vertical weight - each tap is
// filtered on both source rows and lerped, as the sampler does.  The window is walked texel by texel, left to right: texel t
// and its difference to texel t + 1 serve tap q of output i where t = i + {0, 2, 4, 6, -, 9, 11, 13, 15}[q]; the centre tap of
// output i (texel i + 7 or i + 8) is taken at t = i + 8, and tap 3's product waits for it.  Only the sums, the texels around t
// and the quads read ahead are alive at any time.
struct BqWin {   // the walk's state: quads of both source rows' windows, the sums, pending products
  v4f A[15], B[15];
  float s[4][3], p3[4][3], hl[4][3], hlb[4][3];
};
#ifndef RC_BQ_READ_AHEAD
#define RC_BQ_READ_AHEAD 1
#endif
constexpr int bq_quads_before(int t) { return t < 0 ? 0 : ((3 * t + 5) / 4 + RC_BQ_READ_AHEAD < 14 ? (3 * t + 5) / 4 + RC_BQ_READ_AHEAD : 14) + 1; }   // quads read once step t has begun
constexpr int bq_tap_of(int o) { return o == 0 ? 0 : o == 2 ? 1 : o == 4 ? 2 : o == 6 ? 3 : o == 9 ? 5 : o == 11 ? 6 : o == 13 ? 7 : o == 15 ? 8 : -1; }

template <bool TWO, int K, int K_END>
__device__ __forceinline__ void bq_read(BqWin& S, uint32_t win_a, uint32_t win_b) {
  if constexpr (K < K_END) {
    S.A[K] = lds_v4f(win_a + 16u * (uint32_t)K);
    if constexpr (TWO) S.B[K] = lds_v4f(win_b + 16u * (uint32_t)K);
    bq_read<TWO, K + 1, K_END>(S, win_a, win_b);
  }
}
template <bool TWO, int T, int I>
__device__ __forceinline__ void bq_out(BqWin& S, const BqCols& c, const float (&ta)[3], const float (&da)[3], const float (&tb)[3], const float (&db)[3],
                                       float wy, float w78, float w56, float w34, float w12) {
  constexpr int o = T - I, q = bq_tap_of(o);
  if constexpr (q >= 0) {
    const float w = c.w[I][q];
    const float wt = (q == 0 || q == 8) ? w78 : (q == 1 || q == 7) ? w56 : (q == 2 || q == 6) ? w34 : w12;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float h = fma_(w, da[ch], ta[ch]);
      if constexpr (TWO) {
        const float hb = fma_(w, db[ch], tb[ch]);
        h = fma_(wy, hb - h, h);
      }
      if constexpr (q == 0) S.s[I][ch] = wt * h;
      else if constexpr (q == 3) S.p3[I][ch] = wt * h;
      else S.s[I][ch] += wt * h;
    }
  }
  // the centre tap's pair starts at texel I + 7 or I + 8: the horizontal lerp from I + 7 is kept for one step, the one from I + 8
  // decides
  if constexpr (o == 7) {
    const float w = c.w[I][4];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      S.hl[I][ch] = fma_(w, da[ch], ta[ch]);
      if constexpr (TWO) S.hlb[I][ch] = fma_(w, db[ch], tb[ch]);
    }
  }
  if constexpr (o == 8) {
    const bool left = (c.left >> I) & 1u;
    const float w = c.w[I][4];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float hr = fma_(w, da[ch], ta[ch]);
      float h = left ? S.hl[I][ch] : hr;
      if constexpr (TWO) {
        const float hrb = fma_(w, db[ch], tb[ch]);
        const float hb = left ? S.hlb[I][ch] : hrb;
        h = fma_(wy, hb - h, h);
      }
      S.s[I][ch] += h;
      S.s[I][ch] += S.p3[I][ch];
    }
  }
}
template <bool TWO, int T>
__device__ __forceinline__ void bq_step(BqWin& S, const BqCols& c, uint32_t win_a, uint32_t win_b, float wy, float w78, float w56, float w34, float w12) {
  if constexpr (T <= 18) {
    bq_read<TWO, bq_quads_before(T - 1), bq_quads_before(T)>(S, win_a, win_b);
    float ta[3], da[3], tb[3] = {0.f, 0.f, 0.f}, db[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      ta[ch] = RC_BQ_T(S.A, T, ch);
      da[ch] = RC_BQ_T(S.A, T + 1, ch) - ta[ch];
      if constexpr (TWO) {
        tb[ch] = RC_BQ_T(S.B, T, ch);
        db[ch] = RC_BQ_T(S.B, T + 1, ch) - tb[ch];
      }
    }
    bq_out<TWO, T, 0>(S, c, ta, da, tb, db, wy, w78, w56, w34, w12);
    bq_out<TWO, T, 1>(S, c, ta, da, tb, db, wy, w78, w56, w34, w12);
    bq_out<TWO, T, 2>(S, c, ta, da, tb, db, wy, w78, w56, w34, w12);
    bq_out<TWO, T, 3>(S, c, ta, da, tb, db, wy, w78, w56, w34, w12);
#ifndef RC_BQ_NO_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
    bq_step<TWO, T + 1>(S, c, win_a, win_b, wy, w78, w56, w34, w12);
  }
}
template <bool TWO>
__device__ __forceinline__ void bq_filter(const BqCols& c, uint32_t win_a, uint32_t win_b, float wy, float w78, float w56, float w34, float w12,
                                          float (&s)[4][3]) {
  BqWin S;
  bq_step<TWO, 0>(S, c, win_a, win_b, wy, w78, w56, w34, w12);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) s[i][ch] = S.s[i][ch];
}

// a wave-uniform float that the arithmetic should see in a vector register (a scalar operand makes a VALU instruction issue at
// the slow rate)
__device__ __forceinline__ float in_vgpr(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

// One step = one target row of one band and triangle for a pair of frames; the steps of a frame pair are listed once per geometry
// (pass_royale_bloom.hip buildBqSteps: bands in order, rows in blocks of kBhBlockRows, a block the diagonal crosses once per
// triangle), four words each (bq_step_word0/1, the vertical weight, the halation's vertical weight); a wave walks a run of
// consecutive steps of the launch's list (pair p's steps follow pair p - 1's).
__global__ void __launch_bounds__(kBqWaves * 64, 1) k_royale_bloom_h_quad(const PassLaunch L, const uint32_t* __restrict__ cols, const uint32_t* __restrict__ steps,
                                                                         int n_steps, const uint32_t* __restrict__ runs, int group_taps) {
  extern __shared__ uint32_t rc_dyn_lds_[];
  strip2_load_tables(rc_dyn_lds_, L, true);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fslot = lane >> 5, grp = lane & 31;   // frame slot, group of four columns
  const uint32_t lane_dec = 0u;
  const uint32_t ring = kBqLdsRing + (uint32_t)(wave * kBqWaveBytes);
  // the lane's staged group in ring slot 0 - and, the staged columns starting 8 left of the band, the start of its window
  const uint32_t e_main = ring + (uint32_t)(fslot * kBqSlotBytes + grp * 48);
  const uint32_t e_extra = e_main + 32u * 48u;   // groups 32 .. 35, staged by the slot's first four lanes
  const bool has_extra = grp < 4;
  const int W = L.out_w, H = L.out_h;
  const float* P = L.params;
  const float w78 = in_vgpr(P[RPG_W78]), w56 = in_vgpr(P[RPG_W56]), w34 = in_vgpr(P[RPG_W34]), w12 = in_vgpr(P[RPG_W12]), si = in_vgpr(P[RPG_SUM_INV]);
  const float c_main = in_vgpr((P[RPG_MASK_AMPLIFY] * 2.0f) * (1.0f - 0.075f));
  const int hw = L.extra[2].w, hh = L.extra[2].h;
  const int me = (int)blockIdx.x * kBqWaves + wave;
  uint32_t k = runs[me];
  const uint32_t k_end = runs[me + 1];
  if (k >= k_end) return;
  int zp = (int)(k / (uint32_t)n_steps), j = (int)(k - (uint32_t)zp * (uint32_t)n_steps);
  // whole-launch buffers: a lane adds its frame's offset to its column offset, the row base is a scalar
  auto all_frames = [&](const void* base, uint64_t stride, int tw, int th) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)((uint32_t)stride * (uint32_t)(L.n_frames - 1) + (uint32_t)(tw * th * 4)), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t r_in = all_frames(L.in.base, L.in.frame_stride, W, H);
  const __amdgpu_buffer_rsrc_t r_i0 = all_frames(L.extra[0].base, L.extra[0].frame_stride, L.extra[0].w, L.extra[0].h);
  const __amdgpu_buffer_rsrc_t r_i1 = all_frames(L.extra[1].base, L.extra[1].frame_stride, L.extra[1].w, L.extra[1].h);
  const __amdgpu_buffer_rsrc_t r_hal = all_frames(L.extra[2].base, L.extra[2].frame_stride, hw, hh);
  const __amdgpu_buffer_rsrc_t r_out = all_frames(L.out, L.out_frame_stride, W, H);
  const __amdgpu_buffer_rsrc_t r_steps = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(steps), 0, n_steps * 16, 0x00020000);
  // per (frame pair, band, triangle) segment
  BqCols c;
  int px_out = 0, idim_off = 0, bright_off = 0;   // byte offsets of the lane's pixels in the target / of the two single taps' first (or only) texel
  int sx_main = 0, sx_extra = 0, md = 0;          // staged groups: byte offset of the fetch; bits 0-1 / 2-3: the main / extra group lies left (1) or right (2) of the frame
  int hx0 = 0, fo_hal = 0;
  int x0 = 0;
  bool live = false, edge_band = false;
  float hl0[4][3], hld[4][3];
  int hbase = -1000;
  int st_hi = -1000;   // highest source row in the ring
  v4u qm = {0u, 0u, 0u, 0u}, qe = {0u, 0u, 0u, 0u};   // raw texels of row st_q, in flight
  int st_q = -1000;
  v4u nia = {0u, 0u, 0u, 0u}, nja = {0u, 0u, 0u, 0u};   // the single taps of this step, in flight
  bool fresh = true;
  // a step's four words are fetched a step ahead through the vector path (a scalar load in flight would turn every LDS wait of the
  // step into a full drain) and taken over at the END of the step before, ahead of its stores: loads and stores share one in-order
  // counter, so a wait for them placed behind a store would wait for that store's round trip, every step
  v4u dcur = __builtin_amdgcn_raw_buffer_load_b128(r_steps, 0, j * 16, 0);
  uint32_t w0 = __builtin_amdgcn_readfirstlane(dcur.x), w1 = __builtin_amdgcn_readfirstlane(dcur.y);
  float wy = bits2f(dcur.z), hal_wy = bits2f(dcur.w);
  for (; k < k_end; ++k) {
    int jn = j + 1;
    if (jn == n_steps) jn = 0;
    const int y = (int)(w0 & 4095u), hal_y0 = (int)((w0 >> 12) & 1023u) - 1, band = (int)(w0 >> 26);
    const bool side1 = (w0 >> 22) & 1u, mixed = (w0 >> 23) & 1u, newseg = ((w0 >> 24) & 1u) || fresh || j == 0, y0_above = (w0 >> 25) & 1u;
    const int idim_ro = (int)(w1 & 0xffffu) * L.extra[0].w * 4, bright_ro = (int)(w1 >> 16) * L.extra[1].w * 4;
    if (newseg) {
      fresh = false;
      const int z0 = 2 * zp;
      const bool pair = z0 + 1 < L.n_frames;   // the second frame slot holds a frame; an idle one repeats the first frame's work without storing
      const int z = z0 + (pair ? fslot : 0);
      const int xw = band * kBqBand;
      x0 = xw + 4 * grp;
      live = x0 < W && (fslot == 0 || pair);   // (W is a multiple of four: a group is inside the frame or outside)
      const int px_off = min(x0, W - 4) * 4;
      px_out = px_off + z * (int)L.out_frame_stride;
      // staged groups: the lane's own (columns xw - 8 + 4 grp ..) and, for the slot's first four lanes, group 32 + grp.  A group
      // lies inside the frame or outside it: outside, every texel is the frame's first / last one of the row
      const int cg_main = xw - kBqHalo + 4 * grp, cg_extra = cg_main + kBqBand, fo_in = z * (int)L.in.frame_stride;
      sx_main = clampi(cg_main, 0, W - 4) * 4 + fo_in;
      sx_extra = clampi(cg_extra, 0, W - 4) * 4 + fo_in;
      md = (cg_main < 0 ? 1 : (cg_main > W - 4 ? 2 : 0)) | (cg_extra < 0 ? 4 : (cg_extra > W - 4 ? 8 : 0));
      edge_band = xw == 0 || xw + kBqBand + kBqHalo > W;   // wave-uniform
      const __amdgpu_buffer_rsrc_t r_cols = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(cols), 0, BH_COL_FIELDS * 2 * W * 4, 0x00020000);
      const int side = side1 ? 1 : 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const v4u wq = __builtin_amdgcn_raw_buffer_load_b128(r_cols, px_off, ((BH_WX + q) * 2 + side) * W * 4, 0);
        c.w[0][q] = bits2f(wq.x); c.w[1][q] = bits2f(wq.y); c.w[2][q] = bits2f(wq.z); c.w[3][q] = bits2f(wq.w);
      }
      const v4u cs = __builtin_amdgcn_raw_buffer_load_b128(r_cols, px_off, (BH_CSEL * 2 + side) * W * 4, 0);
      c.left = (cs.x & 1u) | ((cs.y & 1u) << 1) | ((cs.z & 1u) << 2) | ((cs.w & 1u) << 3);
      idim_off = (int)__builtin_amdgcn_raw_buffer_load_b32(r_cols, px_off, (BH_IDIM_X * 2 + side) * W * 4, 0) * 4 + z * (int)L.extra[0].frame_stride;
      bright_off = (int)__builtin_amdgcn_raw_buffer_load_b32(r_cols, px_off, (BH_BRIGHT_X * 2 + side) * W * 4, 0) * 4 + z * (int)L.extra[1].frame_stride;
      const v4u h0 = __builtin_amdgcn_raw_buffer_load_b128(r_cols, px_off, (BH_HAL_X0 * 2 + side) * W * 4, 0);
      const v4u hwq = __builtin_amdgcn_raw_buffer_load_b128(r_cols, px_off, (BH_HAL_W * 2 + side) * W * 4, 0);
      c.hw[0] = bits2f(hwq.x); c.hw[1] = bits2f(hwq.y); c.hw[2] = bits2f(hwq.z); c.hw[3] = bits2f(hwq.w);
      hx0 = (int)h0.x;
      c.hright = ((int)h0.y != hx0 ? 2u : 0u) | ((int)h0.z != hx0 ? 4u : 0u) | ((int)h0.w != hx0 ? 8u : 0u);
      fo_hal = z * (int)L.extra[2].frame_stride;
      hbase = st_hi = st_q = -1000;
      // this step's single taps (later steps' are fetched a step ahead)
      if (group_taps & 1) nia.x = __builtin_amdgcn_raw_buffer_load_b32(r_i0, idim_off, idim_ro, 0);
      else nia = __builtin_amdgcn_raw_buffer_load_b128(r_i0, idim_off, idim_ro, 0);
      if (group_taps & 2) nja.x = __builtin_amdgcn_raw_buffer_load_b32(r_i1, bright_off, bright_ro, 0);
      else nja = __builtin_amdgcn_raw_buffer_load_b128(r_i1, bright_off, bright_ro, 0);
    }
    const bool two = __builtin_amdgcn_readfirstlane(f2bits(wy)) != 0u;
    const int row_a = max(y - (y0_above ? 1 : 0), 0), row_b = min(row_a + 1, H - 1);   // (the pair's first row is y - 1 or y, k_bloomh_geometry; both clamped)
    const int row_b2 = y0_above && y == 0 ? 0 : row_b;                                  // (y0 = -1: both rows are row 0)
    const int need = two ? row_b2 : row_a;
    // ---- stage the source rows this target row needs and the ring does not hold yet (normally one).  Other lanes read
    // what a lane writes here: the LDS executes a wave's operations in order, the compiler must not reorder them
    if (st_hi < row_a - 1 || st_hi > need + 1) st_hi = row_a - 1;
    asm volatile("" ::: "memory");
    auto stage_group = [&](uint32_t entry, v4u q, int m) __attribute__((always_inline)) {
      if (edge_band) {
        if (m & 1) q = v4u{q.x, q.x, q.x, q.x};
        if (m & 2) q = v4u{q.w, q.w, q.w, q.w};
      }
      lds_put_v4f(entry, v4f{bq_dec<0>(q.x, lane_dec), bq_dec<1>(q.x, lane_dec), bq_dec<2>(q.x, lane_dec), bq_dec<0>(q.y, lane_dec)});
      lds_put_v4f(entry + 16, v4f{bq_dec<1>(q.y, lane_dec), bq_dec<2>(q.y, lane_dec), bq_dec<0>(q.z, lane_dec), bq_dec<1>(q.z, lane_dec)});
      lds_put_v4f(entry + 32, v4f{bq_dec<2>(q.z, lane_dec), bq_dec<0>(q.w, lane_dec), bq_dec<1>(q.w, lane_dec), bq_dec<2>(q.w, lane_dec)});
    };
    auto stage_row = [&](int r) __attribute__((always_inline)) {   // row r from the texels in flight; then row r + 1's are requested
      const uint32_t slot = (r & 1) ? (uint32_t)kBqRowBytes : 0u;
#ifndef RC_BQ_ABL_NOSTAGE
      stage_group(e_main + slot, qm, md);
      if (has_extra) stage_group(e_extra + slot, qe, md >> 2);
#else
      lds_put_v4f(e_main + slot, v4f{bits2f(qm.x), bits2f(qm.y), bits2f(qm.z), bits2f(qm.w)});
      if (has_extra) lds_put_v4f(e_extra + slot, v4f{bits2f(qe.x), bits2f(qe.y), bits2f(qe.z), bits2f(qe.w)});
#endif
      const int ro = min(r + 1, H - 1) * W * 4;
      qm = __builtin_amdgcn_raw_buffer_load_b128(r_in, sx_main, ro, 0);
      if (has_extra) qe = __builtin_amdgcn_raw_buffer_load_b128(r_in, sx_extra, ro, 0);
      st_q = r + 1;
      st_hi = r;
    };
    // (no loop, and every wait for texels requested in this step inside its rare branch: the common path - the row's texels
    // requested a step ago and waited for before the last step's stores - must not inherit a wait that covers those stores)
    if (st_hi < need) {
      if (st_q != st_hi + 1) {   // nothing in flight for this row (start of a segment, a jump)
        const int ro = (st_hi + 1) * W * 4;
        qm = __builtin_amdgcn_raw_buffer_load_b128(r_in, sx_main, ro, 0);
        if (has_extra) qe = __builtin_amdgcn_raw_buffer_load_b128(r_in, sx_extra, ro, 0);
        asm volatile("" : "+v"(qm), "+v"(qe));
      }
      stage_row(st_hi + 1);
      if (st_hi < need) {   // a second row at once: the first row of a segment with a vertical weight
        asm volatile("" : "+v"(qm), "+v"(qe));
        stage_row(st_hi + 1);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    v4u draw = __builtin_amdgcn_raw_buffer_load_b128(r_steps, 0, jn * 16, 0);   // the next step's words (the last step's look-ahead wraps to a valid entry)
    // ---- halation (320 x 240, magnified six times): the four pixels' pairs start at texel column hx0 or hx0 + 1, so three texels per
    // halation row serve the lane; the horizontally filtered values of the current row pair stay in registers
    if (hal_y0 != hbase) {
      hbase = hal_y0;
      const int ra = clampi(hal_y0, 0, hh - 1) * hw * 4, rb = clampi(hal_y0 + 1, 0, hh - 1) * hw * 4;
      uint32_t hq[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int hx = clampi(hx0 + i, 0, hw - 1) * 4 + fo_hal;
        hq[i] = __builtin_amdgcn_raw_buffer_load_b32(r_hal, hx, ra, 0);
        hq[3 + i] = __builtin_amdgcn_raw_buffer_load_b32(r_hal, hx, rb, 0);
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float fa[3], fb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          fa[i] = ch == 0 ? bq_dec<0>(hq[i], lane_dec) : (ch == 1 ? bq_dec<1>(hq[i], lane_dec) : bq_dec<2>(hq[i], lane_dec));
          fb[i] = ch == 0 ? bq_dec<0>(hq[3 + i], lane_dec) : (ch == 1 ? bq_dec<1>(hq[3 + i], lane_dec) : bq_dec<2>(hq[3 + i], lane_dec));
        }
        const float da0 = fa[1] - fa[0], da1 = fa[2] - fa[1], db0 = fb[1] - fb[0], db1 = fb[2] - fb[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // the sampler's horizontal lerps of both rows
          const bool right = (c.hright >> i) & 1u;
          const float la = fma_(c.hw[i], right ? da1 : da0, right ? fa[1] : fa[0]);
          const float lb = fma_(c.hw[i], right ? db1 : db0, right ? fb[1] : fb[0]);
          hl0[i][ch] = la;
          hld[i][ch] = lb - la;
        }
      }
    }
    // ---- filter
    const uint32_t win_a = e_main + ((row_a & 1) ? (uint32_t)kBqRowBytes : 0u), win_b = e_main + ((row_b2 & 1) ? (uint32_t)kBqRowBytes : 0u);
    float s[4][3];
#if defined(RC_BQ_ABL_NOFILTER)   // (development ablations: wrong bytes, for timing only)
    {
      const v4f qd = lds_v4f(win_a + 96u), qe2 = lds_v4f(win_b + 112u);
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[i][0] = qd[i]; s[i][1] = qe2[i]; s[i][2] = qd[3 - i]; }
    }
#elif defined(RC_BQ_ABL_ONEONLY)
    bq_filter<false>(c, win_a, win_a, 0.0f, w78, w56, w34, w12, s);
#else
    if (two) bq_filter<true>(c, win_a, win_b, wy, w78, w56, w34, w12, s);
    else bq_filter<false>(c, win_a, win_a, 0.0f, w78, w56, w34, w12, s);
#endif
    // ---- reconstitute (as k_royale_bloom_h) and store
    float di[4][3], dj[4][3];   // MASKED_SCANLINES, BRIGHTPASS
    if (group_taps & 1) {
      const float d0 = bq_dec<0>(nia.x, lane_dec), d1 = bq_dec<1>(nia.x, lane_dec), d2 = bq_dec<2>(nia.x, lane_dec);
#pragma unroll
      for (int i = 0; i < 4; ++i) { di[i][0] = d0; di[i][1] = d1; di[i][2] = d2; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { di[i][0] = bq_dec<0>(nia[i], lane_dec); di[i][1] = bq_dec<1>(nia[i], lane_dec); di[i][2] = bq_dec<2>(nia[i], lane_dec); }
    }
    if (group_taps & 2) {
      const float d0 = bq_dec<0>(nja.x, lane_dec), d1 = bq_dec<1>(nja.x, lane_dec), d2 = bq_dec<2>(nja.x, lane_dec);
#pragma unroll
      for (int i = 0; i < 4; ++i) { dj[i][0] = d0; dj[i][1] = d1; dj[i][2] = d2; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { dj[i][0] = bq_dec<0>(nja[i], lane_dec); dj[i][1] = bq_dec<1>(nja[i], lane_dec); dj[i][2] = bq_dec<2>(nja[i], lane_dec); }
    }
    // the next step's words, and its single taps (a new segment fetches its own)
    const uint32_t nw0 = __builtin_amdgcn_readfirstlane(draw.x), nw1 = __builtin_amdgcn_readfirstlane(draw.y);
    {
      const int ro0 = (int)(nw1 & 0xffffu) * L.extra[0].w * 4, ro1 = (int)(nw1 >> 16) * L.extra[1].w * 4;
      if (group_taps & 1) nia.x = __builtin_amdgcn_raw_buffer_load_b32(r_i0, idim_off, ro0, 0);
      else nia = __builtin_amdgcn_raw_buffer_load_b128(r_i0, idim_off, ro0, 0);
      if (group_taps & 2) nja.x = __builtin_amdgcn_raw_buffer_load_b32(r_i1, bright_off, ro1, 0);
      else nja = __builtin_amdgcn_raw_buffer_load_b128(r_i1, bright_off, ro1, 0);
    }
    uint32_t o[4];
#ifdef RC_BQ_ABL_NORECON
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f2bits(s[i][0] + s[i][1] + s[i][2] + di[i][0] + dj[i][1] + hld[i][0] + hl0[i][1]);
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t px = 0xff000000u;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float dimpass = di[i][ch] - dj[i][ch];
        const float v = (dimpass + s[i][ch] * si) * c_main + fma_(hal_wy, hld[i][ch], hl0[i][ch]) * 0.075f;
        px |= bh_srgb8(v) << (8 * ch);
      }
      o[i] = px;
    }
#endif
    // everything fetched ahead for the next step is waited for HERE, before this step's stores are issued (see above)
    asm volatile("" : "+v"(qm), "+v"(qe), "+v"(draw));
#ifdef RC_BQ_ABL_NOSTORE
    if (L.flags != 0x7fffffff) live = false;
#endif
    if (!mixed) {
      if (live) __builtin_amdgcn_raw_buffer_store_b128(v4u{o[0], o[1], o[2], o[3]}, r_out, px_out, y * W * 4, 0);
    } else {
      // a pixel is stored by the pass of its own triangle, (2y+1) W <= (2x+1) H tells which
      const int tri_y = (2 * y + 1) * W;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (live && (tri_y <= (2 * (x0 + i) + 1) * H) != side1) __builtin_amdgcn_raw_buffer_store_b32(o[i], r_out, px_out + 4 * i, y * W * 4, 0);
    }
    w0 = nw0;
    w1 = nw1;
    wy = bits2f(draw.z);
    hal_wy = bits2f(draw.w);
    j = jn;
    if (jn == 0) ++zp;
  }
}
#undef RC_BQ_T

}  // namespace

namespace rcbloomh {
// host side of the quad form's conditions: a 1:1 pass over whole groups of four columns, both frames of a pair within 32-bit
// offsets of the pair's first frame
bool bqGeometryOk(const PassLaunch& L, bool idim_own_column, bool bright_own_column) {
  const uint64_t lim = 1ull << 31;
  // (a tap on the pixels' own columns is fetched 16 bytes at a time: whole groups in the sampled texture as well)
  return (L.out_w & 3) == 0 && L.out_w >= 8 && L.in.w == L.out_w && L.in.h == L.out_h && (!idim_own_column || L.extra[0].w == L.out_w) &&
         (!bright_own_column || L.extra[1].w == L.out_w) && L.in.frame_stride < lim && L.extra[0].frame_stride < lim &&
         L.extra[1].frame_stride < lim && L.extra[2].frame_stride < lim && L.out_frame_stride < lim &&
         (uint64_t)L.out_w * L.out_h * 4 < lim;
}

int bq_waves() { return kBqWaves; }
hipError_t launch_bloom_h_quad(const PassLaunch& L, hipStream_t s, const uint32_t* cols, const uint32_t* steps, int n_steps, const uint32_t* runs, unsigned blocks,
                               int group_taps) {
  auto kq = k_royale_bloom_h_quad;
  // (set on every launch: the attribute is per device, and this needs no shared flag)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBqLdsBytes) != hipSuccess) return hipGetLastError();
  hipLaunchKernelGGL(kq, dim3(blocks), dim3(kBqWaves * 64), kBqLdsBytes, s, L, cols, steps, n_steps, runs, group_taps);
  return hipGetLastError();
}
}  // namespace rcbloomh
