// Shared by the two files of crt-royale's pass 10 (bloom-horizontal-reconstitute.glsl): pass_royale_bloom.hip (general form,
// strip form, the per-geometry tables) and pass_royale_bloom_quad.hip (quad form, compiled without the SLP vectoriser).
#pragma once
#include "royale_strip2.h"

namespace rcbloomh {
using namespace rcd;
using namespace rcstrip2;

constexpr int kBhBlockRows = 8;   // rows classified together by triangle
// per-column and per-row tables of one geometry (k_bloomh_geometry): cols[BH_COL_FIELDS][2 sides][W], rows[H][2 sides][BH_ROW_FIELDS]
enum { BH_DX = 0, BH_WX = 9, BH_IDIM_X = 18, BH_BRIGHT_X = 19, BH_HAL_X0 = 20, BH_HAL_W = 21, BH_CSEL = 22, BH_COL_FIELDS = 23 };
enum { BH_Y0 = 0, BH_WY = 1, BH_IDIM_Y = 2, BH_BRIGHT_Y = 3, BH_HAL_Y0 = 4, BH_HAL_WY = 5, BH_ROW_FIELDS = 8 };

struct BhRow {
  int y0;
  float wy;
  int idim_y, bright_y, hal_y0;
  float hal_wy;
};
// A row's quantities, fetched one step ahead.  They are wave-uniform, but a scalar load in flight would turn every LDS wait of
// the step into a full drain (scalar loads return out of order, so the compiler waits for lgkmcnt(0) while one is pending):
// the record is fetched through the vector path (every lane the same address) and moved to scalar registers when it is used.
struct BhRowRaw {
  v4u a;
  v2u32 b;
};
__device__ __forceinline__ BhRowRaw fetch_bh_row(__amdgpu_buffer_rsrc_t r_rows, int y, int side) {
  const int off = (y * 2 + side) * BH_ROW_FIELDS * 4;
  BhRowRaw r;
  r.a = __builtin_amdgcn_raw_buffer_load_b128(r_rows, 0, off, 0);
  r.b = __builtin_amdgcn_raw_buffer_load_b64(r_rows, 0, off + 16, 0);
  return r;
}
__device__ __forceinline__ BhRow use_bh_row(const BhRowRaw& r) {
  return BhRow{(int)__builtin_amdgcn_readfirstlane(r.a.x), bits2f(__builtin_amdgcn_readfirstlane(r.a.y)), (int)__builtin_amdgcn_readfirstlane(r.a.z),
               (int)__builtin_amdgcn_readfirstlane(r.a.w), (int)__builtin_amdgcn_readfirstlane(r.b.x), bits2f(__builtin_amdgcn_readfirstlane(r.b.y))};
}


// quad form (pass_royale_bloom_quad.hip): whether the launch's sizes qualify, and the launch itself.  `runs`: where each wave's
// run of steps begins (`steps`: the n_steps steps of one frame pair, pass_royale_bloom.hip buildBqSteps), blocks * bq_waves() + 1 entries
bool bqGeometryOk(const PassLaunch& L, bool idim_own_column, bool bright_own_column);
int bq_waves();
hipError_t launch_bloom_h_quad(const PassLaunch& L, hipStream_t s, const uint32_t* cols, const uint32_t* steps, int n_steps, const uint32_t* runs, unsigned blocks,
                               int group_taps);   // bit 0 / 1: MASKED_SCANLINES / BRIGHTPASS is one texel per group of four columns
}  // namespace rcbloomh
