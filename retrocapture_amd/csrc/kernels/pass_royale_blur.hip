// crt-royale passes 3 / 4 (blurs/blur9fast-vertical.glsl, blur9fast-horizontal.glsl; tex2Dblur9fast 1496-1524) in a TILE form.
//
// The general kernel (k_blur9, pass_royale.hip) takes five LINEAR samples per pixel through the generic sampler: 20 texel
// fetches and 60 table decodes per pixel, 497 VALU instructions per 64 pixels - on two 320 x 240 passes that is 2 us of a
// 44 us chain.  Here a workgroup decodes the texels its 64 x 32 pixels can reach ONCE into LDS (three planes of floats, so
// that a tap's texel pair is two adjacent dwords), and a pixel's five taps are four LDS reads and three lerps per channel
// each.  Everything a sampler derives from a coordinate - first texel and weight per tap, per column and per row, for both
// triangles of the quad - comes from per-geometry tables computed on the device with the sampler's own operations
// (k_blur9_geometry), as in the other strip forms (royale_strip.h); the reach of the taps (how far the first texel of any tap
// lies from the pixel) is measured there too and must fit the region a workgroup stages: nothing else about the taps'
// positions is assumed.
// Requires separable coordinates, an sRGB8 LINEAR clamp-to-edge input of the target's size, one of dx / dy exactly zero;
// anything else takes k_blur9.  Same bytes (tests/test_royale_fullsize.py: forms agree; the goldens run through this form).
#include "../rc_log.h"
#include "royale_strip2.h"

using namespace rcd;
using namespace rcroyale;

namespace {
constexpr int kBtW = 64, kBtH = 32, kBtWaves = 8;

struct Blur9Tables {
  uint2* cols = nullptr;   // [2 sides][W][ntx]: first texel (int) and weight (float bits) of the tap(s) in x
  uint2* rows = nullptr;   // [2 sides][H][nty]
  int vertical = 0;        // the taps run along y (dx == 0): ntx = 1, nty = 5; else ntx = 5, nty = 1
  int reach[4] = {0, 0, 0, 0};   // min (x0 - x), max (x0 + 1 - x), min (y0 - y), max (y0 + 1 - y) over every pixel, tap and triangle
  bool usable = false;
  void release() {
    if (cols) (void)hipFree(cols);
    if (rows) (void)hipFree(rows);
    *this = Blur9Tables();
  }
};

// sample_linear_f<., WRAP_EDGE> on one axis (rc_device.h)
__device__ __forceinline__ uint2 blur_tap(float s, int n) {
  const float u = linear_coord<WRAP_EDGE>(s, n), f = __builtin_floorf(u);
  return make_uint2((uint32_t)(int)f, f2bits(u - f));
}
// the five tap coordinates on one axis, as k_blur9 forms them (d = 0 on the other axis: every tap at c itself)
__device__ __forceinline__ void blur9_coords(float c, float k12, float k34, float d, float* o) {
  o[0] = c - k34 * d;
  o[1] = c - k12 * d;
  o[2] = c;
  o[3] = c + k12 * d;
  o[4] = c + k34 * d;
}

__global__ void __launch_bounds__(256) k_blur9_geometry(const PassLaunch L, int vertical, uint2* cols, uint2* rows, int* reach) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float k12 = L.params[RPB_K12], k34 = L.params[RPB_K34], dx = L.params[RPB_DX], dy = L.params[RPB_DY];
  const int ntx = vertical ? 1 : 5, nty = vertical ? 5 : 1;
  if (i < L.out_w)
    for (int side = 0; side < 2; ++side) {
      float c[5];
      blur9_coords(vary(L.plane[0], i, 0, side == 0), k12, k34, dx, c);
      for (int t = 0; t < 5; ++t) {
        const uint2 e = blur_tap(c[t], L.in.w);
        if (vertical) {
          if (t == 0) cols[(size_t)side * L.out_w + i] = e;
          else if (e.x != cols[(size_t)side * L.out_w + i].x || e.y != cols[(size_t)side * L.out_w + i].y) atomicMax(&reach[1], 1 << 20);   // (dx == 0: cannot happen)
        } else {
          cols[((size_t)side * L.out_w + i) * 5 + t] = e;
        }
        atomicMin(&reach[0], (int)e.x - i);
        atomicMax(&reach[1], (int)e.x + 1 - i);
      }
    }
  if (i < L.out_h)
    for (int side = 0; side < 2; ++side) {
      float c[5];
      blur9_coords(vary(L.plane[1], 0, i, side == 0), k12, k34, dy, c);
      for (int t = 0; t < 5; ++t) {
        const uint2 e = blur_tap(c[t], L.in.h);
        if (!vertical) {
          if (t == 0) rows[(size_t)side * L.out_h + i] = e;
          else if (e.x != rows[(size_t)side * L.out_h + i].x || e.y != rows[(size_t)side * L.out_h + i].y) atomicMax(&reach[3], 1 << 20);
        } else {
          rows[((size_t)side * L.out_h + i) * 5 + t] = e;
        }
        atomicMin(&reach[2], (int)e.x - i);
        atomicMax(&reach[3], (int)e.x + 1 - i);
      }
    }
  (void)ntx;
  (void)nty;
}

// One workgroup renders 64 x 32 tiles of the launch's frames (grid-stride over tiles x frames: few, long-lived workgroups, so
// that the encode table is copied into LDS a few hundred times per launch, not once per tile); wave w renders rows 4 w .. 4 w + 3
// of a tile, a lane one column.  The staged region has a fixed shape per direction - the taps' axis reaches kBtTapLo ..
// kBtTapHi texels from the pixel, the other axis -1 .. +1 (the measured reach must fit: launch_blur9_tile) - so that every LDS
// read of a tap is one base address plus immediates.  Layout: [row][channel][column].
constexpr int kBtTapLo = -4, kBtTapHi = 5, kBtCrossLo = -1, kBtCrossHi = 1;
template <bool VERTICAL>
__global__ void __launch_bounds__(kBtWaves * 64) k_blur9_tile(const PassLaunch L, const uint2* __restrict__ cols, const uint2* __restrict__ rows) {
  RC_SRGB_LDS(lds, L);
  float* const tile = reinterpret_cast<float*>(rc_dyn_lds_) + 256 + (srgb_enc_in_lds(L) ? (int)kSrgbRuns : 0);
  constexpr int hx0 = VERTICAL ? kBtCrossLo : kBtTapLo, hx1 = VERTICAL ? kBtCrossHi : kBtTapHi;
  constexpr int hy0 = VERTICAL ? kBtTapLo : kBtCrossLo, hy1 = VERTICAL ? kBtTapHi : kBtCrossHi;
  constexpr int P = kBtW + hx1 - hx0, R = kBtH + hy1 - hy0, RP = 3 * P;   // staged columns, rows; dwords per staged row
  constexpr int NTX = VERTICAL ? 1 : 5, NTY = VERTICAL ? 5 : 1, kRowsPerWave = kBtH / kBtWaves;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = L.out_w, H = L.out_h;
  const int tiles_x = (W + kBtW - 1) / kBtW, tiles_y = (H + kBtH - 1) / kBtH, per_frame = tiles_x * tiles_y, total = per_frame * L.n_frames;
  const float w12 = L.params[RPB_W12], w34 = L.params[RPB_W34], sum_inv = L.params[RPB_SUM_INV];
  // a thread's share of a tile's staged texels, fetched while the previous tile is rendered
  constexpr int kStage = (P * R + kBtWaves * 64 - 1) / (kBtWaves * 64);
  uint32_t raw[kStage];
  auto fetch = [&](int ti) __attribute__((always_inline)) {
    const int z = ti / per_frame, rem = ti - z * per_frame, tyi = rem / tiles_x;
    const int tx = (rem - tyi * tiles_x) * kBtW, ty = tyi * kBtH;
    const uint32_t* img = reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z));
#pragma unroll
    for (int j = 0; j < kStage; ++j) {
      const int i = tid + j * kBtWaves * 64, r = i / P, c = i - r * P;   // (i beyond the region: a clamped, unused texel)
      raw[j] = img[clampi(ty + hy0 + r, 0, L.in.h - 1) * L.in.w + clampi(tx + hx0 + c, 0, L.in.w - 1)];
    }
  };
  if ((int)blockIdx.x < total) fetch((int)blockIdx.x);
  for (int ti = (int)blockIdx.x; ti < total; ti += (int)gridDim.x) {
    const int z = ti / per_frame, rem = ti - z * per_frame, tyi = rem / tiles_x;
    const int tx = (rem - tyi * tiles_x) * kBtW, ty = tyi * kBtH;
    __syncthreads();   // the previous tile's readers
#pragma unroll
    for (int j = 0; j < kStage; ++j) {
      const int i = tid + j * kBtWaves * 64, r = i / P, c = i - r * P;
      if (i < P * R) {
        float* q = tile + r * RP + c;
        q[0] = lds.dec[raw[j] & 255u];
        q[P] = lds.dec[(raw[j] >> 8) & 255u];
        q[2 * P] = lds.dec[(raw[j] >> 16) & 255u];
      }
    }
    __syncthreads();
    if (ti + (int)gridDim.x < total) fetch(ti + (int)gridDim.x);
    const int x = tx + lane;
    if (x < W) {
      // the column's tap(s), for both triangles; a row's come through wave-uniform loads
      uint2 ecx[2][NTX];
#pragma unroll
      for (int sd = 0; sd < 2; ++sd)
#pragma unroll
        for (int t = 0; t < NTX; ++t) ecx[sd][t] = cols[((size_t)sd * W + x) * NTX + t];
#pragma unroll
      for (int k = 0; k < kRowsPerWave; ++k) {
        const int y = ty + wave * kRowsPerWave + k;   // (wave-uniform)
        if (y < H) {
          const bool up = !rcd::lower_tri(x, y, W, H);
          const uint2* ry0 = rows + (size_t)y * NTY;
          const uint2* ry1 = rows + ((size_t)H + y) * NTY;
          float s[5][3];
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            const uint2 ex0 = ecx[0][VERTICAL ? 0 : t], ex1 = ecx[1][VERTICAL ? 0 : t], ey0 = ry0[VERTICAL ? t : 0], ey1 = ry1[VERTICAL ? t : 0];
            const int ix = (int)(up ? ex1.x : ex0.x) - (tx + hx0), iy = (int)(up ? ey1.x : ey0.x) - (ty + hy0);
            const float wx = bits2f(up ? ex1.y : ex0.y), wy = bits2f(up ? ey1.y : ey0.y);
            const float* q = tile + (__mul24(iy, RP) + ix);
#pragma unroll
            for (int c = 0; c < 3; ++c)
              s[t][c] = lerp_(wy, lerp_(wx, q[c * P], q[c * P + 1]), lerp_(wx, q[RP + c * P], q[RP + c * P + 1]));
          }
          // the weighted sum in the GL's order (k_blur9)
          float o[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) o[c] = ((((w34 * s[0][c] + s[2][c]) + w12 * s[1][c]) + w12 * s[3][c]) + w34 * s[4][c]) * sum_inv;
          St<FMT_SRGB8>::put(L, z, x, y, make_float4(o[0], o[1], o[2], 1.0f), &lds);
        }
      }
    }
  }
}

void buildBlur9Tables(const PassLaunch& L, hipStream_t s, Blur9Tables* T) {
  const float dx = L.params[RPB_DX], dy = L.params[RPB_DY];
  T->vertical = dx == 0.0f ? 1 : 0;
  const int ntx = T->vertical ? 1 : 5, nty = T->vertical ? 5 : 1;
  int* reach = nullptr;
  int hreach[4] = {1 << 20, -(1 << 20), 1 << 20, -(1 << 20)};
  bool ok = (dx == 0.0f) != (dy == 0.0f) && hipMalloc(reinterpret_cast<void**>(&T->cols), (size_t)2 * L.out_w * ntx * sizeof(uint2)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T->rows), (size_t)2 * L.out_h * nty * sizeof(uint2)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&reach), sizeof(hreach)) == hipSuccess;
  if (ok) ok = hipMemcpyAsync(reach, hreach, sizeof(hreach), hipMemcpyHostToDevice, s) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_blur9_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, T->vertical, T->cols, T->rows, reach);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(hreach, reach, sizeof(hreach), hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipStreamSynchronize(s) == hipSuccess;
  }
  if (reach) (void)hipFree(reach);
  for (int i = 0; i < 4; ++i) T->reach[i] = hreach[i];
  // every tap pair inside the region the tile kernel stages around its pixels
  const int* rt = T->vertical ? hreach + 2 : hreach;   // the taps' axis
  const int* rc = T->vertical ? hreach : hreach + 2;   // the other one
  T->usable = ok && rt[0] >= kBtTapLo && rt[1] <= kBtTapHi && rc[0] >= kBtCrossLo && rc[1] <= kBtCrossHi && rt[0] <= rt[1] && rc[0] <= rc[1];
  RC_LOG_DEBUG(std::string("blur9fast-") + (T->vertical ? "vertical " : "horizontal ") + std::to_string(L.out_w) + "x" + std::to_string(L.out_h) + ": tile form " +
               (T->usable ? "in use" : "not usable") + ", reach x " + std::to_string(hreach[0]) + " .. " + std::to_string(hreach[1]) + ", y " +
               std::to_string(hreach[2]) + " .. " + std::to_string(hreach[3]));
  if (!T->usable) {
    const Blur9Tables keep = *T;
    T->release();
    T->vertical = keep.vertical;
  }
}
}  // namespace

namespace rck {
// true: the pass was launched in the tile form (*err = the launch's status); false: the caller takes k_blur9
bool launch_blur9_tile(const PassLaunch& L, hipStream_t s, hipError_t* err) {
  if ((L.flags & RC_FLAG_GENERAL_ONLY) || !SrgbLinEdge::matches(L.in) || L.out_fmt != FMT_SRGB8 || L.in.n_levels > 1 || L.in.dec != nullptr ||
      L.in.w != L.out_w || L.in.h != L.out_h || !rcstrip::separable(L, 0, 1))
    return false;
  static std::mutex mu;
  static std::map<rcstrip::GeoKey, rcstrip::GeoCached<Blur9Tables>> cache;
  const auto T = rcstrip::geo_tables<Blur9Tables>(L, s, mu, cache, buildBlur9Tables);
  if (!T) return false;
  const int P = kBtW + (T->vertical ? kBtCrossHi - kBtCrossLo : kBtTapHi - kBtTapLo), R = kBtH + (T->vertical ? kBtTapHi - kBtTapLo : kBtCrossHi - kBtCrossLo);
  const unsigned lds_bytes = rcd::srgb_lds_bytes(L) + (unsigned)(3 * P * R) * 4u;
  const long tiles = (long)((L.out_w + kBtW - 1) / kBtW) * ((L.out_h + kBtH - 1) / kBtH) * L.n_frames;
  const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);
  auto kernel = T->vertical ? k_blur9_tile<true> : k_blur9_tile<false>;
  // (set on every launch: the attribute is per device, and this needs no shared flag)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
    *err = hipGetLastError();
    return true;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBtWaves * 64), lds_bytes, s, L, T->cols, T->rows);
  *err = hipGetLastError();
  return true;
}
}  // namespace rck
