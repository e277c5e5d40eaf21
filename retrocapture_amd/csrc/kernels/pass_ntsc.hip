// Pass kernels for ntsc/ntsc-256px-svideo.glslp (arithmetic spec = the GLSL text):
//   ntsc/shaders/ntsc-pass1-svideo-3phase.glsl   VS lines 64-70, FS lines 147-166
//   ntsc/shaders/ntsc-pass2-3phase-gamma.glsl    VS lines 44-49, FS lines 215-281 (the file
//     carries "#version 130", so the unrolled macro_loopz branch is the one that is compiled)
// Operation order is the one Mesa's GLSL lowering produces: vec*mat is a dot per matrix column
// evaluated x*c0 + (y*c1 + z*c2); mod(x,3) = x - 3*floor(x/3) with a true division.
#include "pass_launch.h"

using namespace rcd;

namespace {

// sin and cos of the same argument with one shared range reduction (same results as
// sin_() / cos_(): both polynomials are evaluated once and each function selects its own).
__device__ __forceinline__ void sincos_both(float x, float& s, float& c) {
  const uint32_t xi = f2bits(x);
  const float xa = bits2f(xi & 0x7fffffffu);
  const uint32_t sign = xi & 0x80000000u;
  const float y = xa * 1.27323954473516f;
  int32_t j = (int32_t)y;
  j = (j + 1) & ~1;
  const float y2 = (float)j;
  float x3 = fma_(y2, -0.78515625f, xa);
  x3 = fma_(y2, -2.4187564849853515625e-4f, x3);
  x3 = fma_(y2, -3.77489497744594108e-8f, x3);
  const float z = x3 * x3;
  float ys = fma_(-1.9515295891E-4f, z, 8.3321608736E-3f);
  ys = fma_(ys, z, -1.6666654611E-1f);
  ys = ys * z;
  ys = fma_(ys, x3, x3);
  float yc = fma_(2.443315711809948E-005f, z, -1.388731625493765E-003f);
  yc = fma_(yc, z, 4.166664568298827E-002f);
  yc = yc * z;
  yc = yc * z;
  yc = yc - z * 0.5f;
  yc = yc + 1.0f;
  const int32_t j2 = j - 2;
  const float rs = (j & 2) == 0 ? ys : yc;
  const float rc = (j2 & 2) == 0 ? ys : yc;
  s = bits2f(f2bits(rs) ^ (sign ^ (((uint32_t)j & 4u) << 29)));
  c = bits2f(f2bits(rc) ^ (((uint32_t)~j2 & 4u) << 29));
}

// plane[0], plane[1]: TEX0;  plane[2], plane[3]: pix_no
// The four pass-1 files differ in two #defines: COMPOSITE / SVIDEO (the cross-talk matrix mix_mat) and
// TWO_PHASE / THREE_PHASE (chroma phase and CHROMA_MOD_FREQ).
template <int IN_FMT, int IN_LINEAR, int IN_WRAP, int OUT_FMT, bool GENERIC, bool COMPOSITE, bool TWO_PHASE>
__global__ void __launch_bounds__(256) k_ntsc_pass1(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float k_phase = TWO_PHASE ? 3.14159265f : 0.6667f * 3.14159265f;
  const float k_freq = TWO_PHASE ? (4.0f * 3.14159265f) / 15.0f : 3.14159265f / 3.0f;
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float pnx = vary(L.plane[2], x, y, lo), pny = vary(L.plane[3], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 col = GENERIC ? sample_rt(L.in, img, u, v, &lds) : sample<IN_FMT, IN_LINEAR, IN_WRAP>(L.in, img, u, v, &lds);
  const float yy = col.x * 0.2989f + (col.y * 0.5870f + col.z * 0.1140f);
  float ii = col.x * 0.5959f + (col.y * -0.2744f + col.z * -0.3216f);
  float qq = col.x * 0.2115f + (col.y * -0.5229f + col.z * 0.3114f);
  // mod(pix_no.y, period) = y - period*floor(y/period); /2 is exact as *0.5, /3 by div_const_
  const float m3 = TWO_PHASE ? pny - 2.0f * __builtin_floorf(pny * 0.5f)
                             : pny - 3.0f * __builtin_floorf(div_const_(pny, 3.0f, 1.0f / 3.0f));
  const float fc = (float)(L.frame_count0 + z);
  const float mod_phase = k_phase * (m3 + fc) + pnx * k_freq;
  float i_mod, q_mod;
  sincos_both(mod_phase, q_mod, i_mod);
  ii *= i_mod;
  qq *= q_mod;
  // yiq *= mix_mat (a dot per column; factors 1, 2, 0 are exact).  COMPOSITE's first column (1,1,1):
  // the plain addend yy joins the pending multiply-add's addend first, (yy + qq) + ii (float targets)
  const float my = COMPOSITE ? (yy + qq) + ii : yy;
  float mi = COMPOSITE ? yy + (ii * 2.0f + 0.0f) : ii * 2.0f;
  float mq = COMPOSITE ? yy + (0.0f + qq * 2.0f) : qq * 2.0f;
  mi *= i_mod;
  mq *= q_mod;
  const float4 o = make_float4(my, mi, mq, 1.0f);
  if (GENERIC) store_rt(L, z, x, y, o, &lds);
  else store<OUT_FMT>(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

__constant__ float k_luma3[25] = {
    -0.000012020f, -0.000022146f, -0.000013155f, -0.000012020f, -0.000049979f, -0.000113940f, -0.000122150f,
    -0.000005612f, 0.000170516f,  0.000237199f,  0.000169640f,  0.000285688f,  0.000984574f,  0.002018683f,
    0.002002275f,  -0.000909882f, -0.007049081f, -0.013222860f, -0.012606931f, 0.002460860f,  0.035868225f,
    0.084016453f,  0.135563500f,  0.175261268f,  0.190176552f};
__constant__ float k_chroma3[25] = {
    -0.000118847f, -0.000271306f, -0.000502642f, -0.000930833f, -0.001451013f, -0.002064744f, -0.002700432f,
    -0.003241276f, -0.003524948f, -0.003350284f, -0.002491729f, -0.000721149f, 0.002164659f,  0.006313635f,
    0.011789103f,  0.018545660f,  0.026414396f,  0.035100710f,  0.044196567f,  0.053207202f,  0.061590275f,
    0.068803602f,  0.074356193f,  0.077856564f,  0.079052396f};

// 2-phase tables (ntsc-pass2-2phase*.glsl :116-182, the unrolled "#version 130" branch)
__constant__ float k_luma2[33] = {
    -0.000174844f, -0.000205844f, -0.000149453f, -0.000051693f, 0.000000000f,  -0.000066171f, -0.000245058f,
    -0.000432928f, -0.000472644f, -0.000252236f, 0.000198929f,  0.000687058f,  0.000944112f,  0.000803467f,
    0.000363199f,  0.000013422f,  0.000253402f,  0.001339461f,  0.002932972f,  0.003983485f,  0.003026683f,
    -0.001102056f, -0.008373026f, -0.016897700f, -0.022914480f, -0.021642347f, -0.008863273f, 0.017271957f,
    0.054921920f,  0.098342579f,  0.139044281f,  0.168055832f,  0.178571429f};
__constant__ float k_chroma2[33] = {
    0.001384762f, 0.001678312f, 0.002021715f, 0.002420562f, 0.002880460f, 0.003406879f, 0.004004985f,
    0.004679445f, 0.005434218f, 0.006272332f, 0.007195654f, 0.008204665f, 0.009298238f, 0.010473450f,
    0.011725413f, 0.013047155f, 0.014429548f, 0.015861306f, 0.017329037f, 0.018817382f, 0.020309220f,
    0.021785952f, 0.023227857f, 0.024614500f, 0.025925203f, 0.027139546f, 0.028237893f, 0.029201910f,
    0.030015081f, 0.030663170f, 0.031134640f, 0.031420995f, 0.031517031f};

// The six pass-2 files: TAPS 24 (3-phase) or 32 (2-phase); epilogue EPI 0 plain, 1 pow(rgb, 2.5/2.0)
// (-gamma), 2 pow(rgb, 2.4) (-linear)
template <int TAPS> __device__ __forceinline__ float luma_w(int i) { return TAPS == 24 ? k_luma3[i] : k_luma2[i]; }
template <int TAPS> __device__ __forceinline__ float chroma_w(int i) { return TAPS == 24 ? k_chroma3[i] : k_chroma2[i]; }
template <int EPI>
__device__ __forceinline__ float4 ntsc_epilogue(float r, float g, float b) {
  if (EPI == 0) return make_float4(r, g, b, 1.0f);
  const float gm = EPI == 1 ? 2.5f / 2.0f : 2.4f;
  return make_float4(pow_(r, gm), pow_(g, gm), pow_(b, gm), 1.0f);
}

// plane[0], plane[1]: TEX0 = TexCoord - (0.5 / SourceSize.x, 0)
template <int IN_FMT, int IN_LINEAR, int IN_WRAP, int OUT_FMT, bool GENERIC, int TAPS, int EPI>
__global__ void __launch_bounds__(256) k_ntsc_pass2(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float one_x = 1.0f / (float)L.in.w;
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  float sy = 0.f, si = 0.f, sq = 0.f;
  constexpr int kUnroll = GENERIC ? 1 : TAPS;  // the run-time sampler switch is not worth 2*TAPS copies
#pragma unroll kUnroll
  for (int c = 1; c <= TAPS; ++c) {
    const float off = (float)(c - 1 - TAPS);
    const float4 p = GENERIC ? sample_rt(L.in, img, u + off * one_x, v, &lds)
                             : sample<IN_FMT, IN_LINEAR, IN_WRAP>(L.in, img, u + off * one_x, v, &lds);
    const float4 n = GENERIC ? sample_rt(L.in, img, u + (-off) * one_x, v, &lds)
                             : sample<IN_FMT, IN_LINEAR, IN_WRAP>(L.in, img, u + (-off) * one_x, v, &lds);
    sy = sy + (p.x + n.x) * luma_w<TAPS>(c - 1);
    si = si + (p.y + n.y) * chroma_w<TAPS>(c - 1);
    sq = sq + (p.z + n.z) * chroma_w<TAPS>(c - 1);
  }
  const float4 m = GENERIC ? sample_rt(L.in, img, u, v, &lds) : sample<IN_FMT, IN_LINEAR, IN_WRAP>(L.in, img, u, v, &lds);
  sy = sy + m.x * luma_w<TAPS>(TAPS);
  si = si + m.y * chroma_w<TAPS>(TAPS);
  sq = sq + m.z * chroma_w<TAPS>(TAPS);
  // yiq2rgb = a dot per column with first factor 1.0: the plain addend joins the pending multiply-add's
  // addend first, (x + z*c2) + y*c1 (measured on float targets)
  const float r = (sy + sq * 0.6210f) + si * 0.956f;
  const float g = (sy + sq * -0.6474f) + si * -0.2720f;
  const float b = (sy + sq * 1.7046f) + si * -1.1060f;
  const float4 o = ntsc_epilogue<EPI>(r, g, b);
  if (GENERIC) store_rt(L, z, x, y, o, &lds);
  else store<OUT_FMT>(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

// ---- pass 2, row-staged form ------------------------------------------------------------------
// The 49 taps of a pixel lie on one source row at consecutive columns, and the windows of the 64
// pixels a wave computes overlap: 49 x 64 sixteen-byte fetches for 176 distinct texels.  When the
// host has verified (kernel_registry.cpp, ntscTapsAreRegular: every tap of every target column,
// evaluated with these very float operations) that tap k of column x reads source column
// c(x) + k - 24 with c(x + 1) = c(x) + 2, each wave stages its row segment once - wrap applied while
// staging, Y / I / Q split into planes and de-interleaved by column parity so that the 64 lanes of a
// tap read 64 consecutive words - and every tap is three LDS reads at compile-time offsets.
// Accumulation order and arithmetic are those of k_ntsc_pass2.
constexpr int kNtscSegMax = 192;              // 2 * 63 + (2 * 32 + 1) columns, rounded up to even
struct NtscRow { float y[2][kNtscSegMax / 2], i[2][kNtscSegMax / 2], q[2][kNtscSegMax / 2]; };

template <int IN_FMT, int IN_WRAP, int TAPS, int EPI>
__global__ void __launch_bounds__(256, 8) k_ntsc_pass2_rows(const PassLaunch L) {
  constexpr int kNtscSeg = (2 * 63 + 2 * TAPS + 1 + 1) & ~1;
  __shared__ NtscRow rows[4];
  const int tiles_x = (L.out_w + 63) >> 6, tiles_y = (L.out_h + 3) >> 2;
  const int tiles_per_frame = tiles_x * tiles_y, n_tiles = tiles_per_frame * L.n_frames;
  const int lane = threadIdx.x, wv = threadIdx.y;
  NtscRow& row = rows[wv];
  for (int tile_i = blockIdx.x; tile_i < n_tiles; tile_i += gridDim.x) {
    const int z = tile_i / tiles_per_frame, rem = tile_i - z * tiles_per_frame;
    const int tyi = rem / tiles_x, x0 = (rem - tyi * tiles_x) * 64;
    const int x = x0 + lane, y = min(tyi * 4 + wv, L.out_h - 1);  // a row beyond the target recomputes the last one
    const bool live = (tyi * 4 + wv) < L.out_h && x < L.out_w;
    const uint8_t* img = frame_ptr(L.in, z);
    // source row of this wave and first source column of its segment (rectangle-path planes: u depends on x only)
    const float v = vary(L.plane[1], x0, y, false);
    const int sy_raw = (int)__builtin_floorf(v * (float)L.in.h);
    const int c_first = (int)__builtin_floorf(vary(L.plane[0], x0, y, false) * (float)L.in.w) - TAPS;
    __syncthreads();  // every wave is done reading the previous tile's rows
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      const int j = part * 64 + lane;  // segment column
      if (j < kNtscSeg) {
        const int sx_raw = c_first + j;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (IN_WRAP == WRAP_BORDER) {
          if (sx_raw >= 0 && sx_raw < L.in.w && sy_raw >= 0 && sy_raw < L.in.h) t = texel<IN_FMT>(L.in, img, sx_raw, sy_raw, nullptr);
        } else {
          t = texel<IN_FMT>(L.in, img, clampi(sx_raw, 0, L.in.w - 1), clampi(sy_raw, 0, L.in.h - 1), nullptr);
        }
        row.y[j & 1][j >> 1] = t.x;
        row.i[j & 1][j >> 1] = t.y;
        row.q[j & 1][j >> 1] = t.z;
      }
    }
    __syncthreads();
    // tap k (0..2*TAPS) of lane l sits at segment column 2l + k
    float sy = 0.f, si = 0.f, sq = 0.f;
#pragma unroll
    for (int c = 1; c <= TAPS; ++c) {
      const int kp = c - 1, kn = 2 * TAPS + 1 - c;  // offsets c - 1 - TAPS and TAPS + 1 - c
      const float py = row.y[kp & 1][lane + (kp >> 1)], ny = row.y[kn & 1][lane + (kn >> 1)];
      const float pi = row.i[kp & 1][lane + (kp >> 1)], ni = row.i[kn & 1][lane + (kn >> 1)];
      const float pq = row.q[kp & 1][lane + (kp >> 1)], nq = row.q[kn & 1][lane + (kn >> 1)];
      sy = sy + (py + ny) * luma_w<TAPS>(c - 1);
      si = si + (pi + ni) * chroma_w<TAPS>(c - 1);
      sq = sq + (pq + nq) * chroma_w<TAPS>(c - 1);
    }
    sy = sy + row.y[TAPS & 1][lane + (TAPS >> 1)] * luma_w<TAPS>(TAPS);
    si = si + row.i[TAPS & 1][lane + (TAPS >> 1)] * chroma_w<TAPS>(TAPS);
    sq = sq + row.q[TAPS & 1][lane + (TAPS >> 1)] * chroma_w<TAPS>(TAPS);
    const float r = (sy + sq * 0.6210f) + si * 0.956f;  // see k_ntsc_pass2
    const float g = (sy + sq * -0.6474f) + si * -0.2720f;
    const float b = (sy + sq * 1.7046f) + si * -1.1060f;
    if (live) store<FMT_RGBA8>(L, z, x, tyi * 4 + wv, ntsc_epilogue<EPI>(r, g, b), nullptr);
  }
}

}  // namespace

namespace rck {

template <bool COMPOSITE, bool TWO_PHASE>
hipError_t launch_pass1(const PassLaunch& L, hipStream_t s) {
  // shipped presets: nearest on the RGB source frame, RGBA32F target
  if (L.in.fmt == FMT_RGBX8 && !L.in.linear && L.in.wrap == WRAP_EDGE && L.out_fmt == FMT_F32)
    hipLaunchKernelGGL((k_ntsc_pass1<FMT_RGBX8, 0, WRAP_EDGE, FMT_F32, false, COMPOSITE, TWO_PHASE>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else if (L.in.fmt == FMT_RGBX8 && !L.in.linear && L.in.wrap == WRAP_EDGE && L.out_fmt == FMT_F16)   // fp16 storage option
    hipLaunchKernelGGL((k_ntsc_pass1<FMT_RGBX8, 0, WRAP_EDGE, FMT_F16, false, COMPOSITE, TWO_PHASE>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else if (L.in.fmt == FMT_RGBX8 && !L.in.linear && L.in.wrap == WRAP_BORDER && L.out_fmt == FMT_F32)
    hipLaunchKernelGGL((k_ntsc_pass1<FMT_RGBX8, 0, WRAP_BORDER, FMT_F32, false, COMPOSITE, TWO_PHASE>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else
    hipLaunchKernelGGL((k_ntsc_pass1<0, 0, 0, 0, true, COMPOSITE, TWO_PHASE>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
template <int TAPS, int EPI>
hipError_t launch_pass2(const PassLaunch& L, hipStream_t s) {
  if ((L.in.fmt == FMT_F32 || L.in.fmt == FMT_F16) && !L.in.linear && L.out_fmt == FMT_RGBA8 && (L.flags & RC_FLAG_NTSC_REGULAR) &&
      !(L.flags & RC_FLAG_GENERAL_ONLY) && (L.in.wrap == WRAP_EDGE || L.in.wrap == WRAP_BORDER)) {
    if (L.in.fmt == FMT_F32) {
      if (L.in.wrap == WRAP_EDGE) hipLaunchKernelGGL((k_ntsc_pass2_rows<FMT_F32, WRAP_EDGE, TAPS, EPI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
      else hipLaunchKernelGGL((k_ntsc_pass2_rows<FMT_F32, WRAP_BORDER, TAPS, EPI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
    } else {
      if (L.in.wrap == WRAP_EDGE) hipLaunchKernelGGL((k_ntsc_pass2_rows<FMT_F16, WRAP_EDGE, TAPS, EPI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
      else hipLaunchKernelGGL((k_ntsc_pass2_rows<FMT_F16, WRAP_BORDER, TAPS, EPI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
    }
    return hipGetLastError();
  }
  if (L.in.fmt == FMT_F32 && !L.in.linear && L.in.wrap == WRAP_EDGE && L.out_fmt == FMT_RGBA8)
    hipLaunchKernelGGL((k_ntsc_pass2<FMT_F32, 0, WRAP_EDGE, FMT_RGBA8, false, TAPS, EPI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else if (L.in.fmt == FMT_F32 && !L.in.linear && L.in.wrap == WRAP_BORDER && L.out_fmt == FMT_RGBA8)
    hipLaunchKernelGGL((k_ntsc_pass2<FMT_F32, 0, WRAP_BORDER, FMT_RGBA8, false, TAPS, EPI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  else
    hipLaunchKernelGGL((k_ntsc_pass2<0, 0, 0, 0, true, TAPS, EPI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_ntsc_pass1(const PassLaunch& L, hipStream_t s) { return launch_pass1<false, false>(L, s); }
hipError_t launch_ntsc_pass1_composite_3phase(const PassLaunch& L, hipStream_t s) { return launch_pass1<true, false>(L, s); }
hipError_t launch_ntsc_pass1_svideo_2phase(const PassLaunch& L, hipStream_t s) { return launch_pass1<false, true>(L, s); }
hipError_t launch_ntsc_pass1_composite_2phase(const PassLaunch& L, hipStream_t s) { return launch_pass1<true, true>(L, s); }
hipError_t launch_ntsc_pass2(const PassLaunch& L, hipStream_t s) { return launch_pass2<24, 1>(L, s); }
hipError_t launch_ntsc_pass2_3phase_linear(const PassLaunch& L, hipStream_t s) { return launch_pass2<24, 2>(L, s); }
hipError_t launch_ntsc_pass2_3phase_plain(const PassLaunch& L, hipStream_t s) { return launch_pass2<24, 0>(L, s); }
hipError_t launch_ntsc_pass2_2phase_gamma(const PassLaunch& L, hipStream_t s) { return launch_pass2<32, 1>(L, s); }
hipError_t launch_ntsc_pass2_2phase_linear(const PassLaunch& L, hipStream_t s) { return launch_pass2<32, 2>(L, s); }
hipError_t launch_ntsc_pass2_2phase_plain(const PassLaunch& L, hipStream_t s) { return launch_pass2<32, 0>(L, s); }

}  // namespace rck
