// Pass kernels for the crt-royale preset (reference shaders/shaders_glsl/crt/crt-royale.glslp):
// GLSL files crt/shaders/crt-royale/src/crt-royale-*.glsl and blurs/blur9fast-{vertical,
// horizontal}.glsl.  Passes 0-10 carry no #pragma parameter, so the reference compiles them
// without PARAMETER_UNIFORM and they run on the static user-settings constants (first-pass file
// lines 104-473); pass 11 takes its 44 parameters from PassLaunch::params.
//
// Everything a vertex shader derives from uniforms alone (tile sizes, blur sigma and weights,
// uv scales) is computed once per launch on the host (royale_setup.cpp) with the same float
// operations and handed over in PassLaunch::params / planes; the kernels do the per-pixel part.
// One thread per target pixel, 64x4 workgroups, blockIdx.z = frame.
#include <cmath>
#include <cstdio>
#include <vector>
#include <cstdlib>

#include "../rc_log.h"
#include "royale_strip2.h"

using namespace rcd;
using namespace rcroyale;

namespace {

// ------------------------------------------------------------------------------- P0 ------
// first-pass-linearize-crt-gamma-bob-fields.glsl FS 4850-4884.  The source texel is a byte per
// channel, so pow(texel, crt_gamma = 2.5) takes 256 values: tabulated once per workgroup.
__global__ void __launch_bounds__(256) k_royale_first(const PassLaunch L) {
    __shared__ float lin[256];
  RC_SRGB_LDS(lds, L);
  {
    const int t = threadIdx.y * 64 + threadIdx.x;
    lin[t] = pow_((float)t * (1.0f / 255.0f), 2.5f);
  }
  __syncthreads();
  RC_TILE_LOOP_BEGIN
  const float tsy = (float)L.in.h;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  const float interlaced = L.params[RP0_INTERLACED];
  // tex2D_linearize on the byte texels: decode == table lookup of the stored byte
  auto tap = [&](float tv) -> float4 {
    // the generic sampler returns byte/255 for RGBX8/RGBA8 nearest taps; recover the byte
    float4 c = sample_rt(L.in, img, u, tv, &lds);
    if (L.in.fmt == FMT_RGBX8 || L.in.fmt == FMT_RGBA8) {
      if (!L.in.linear || true) {
        // values are k/255 only for nearest or fixed-point linear filtering (both quantised)
        int r = (int)__builtin_rintf(c.x * 255.0f), g = (int)__builtin_rintf(c.y * 255.0f), b = (int)__builtin_rintf(c.z * 255.0f);
        if ((float)r * (1.0f / 255.0f) == c.x && (float)g * (1.0f / 255.0f) == c.y && (float)b * (1.0f / 255.0f) == c.z)
          return make_float4(lin[r], lin[g], lin[b], c.w);
      }
    }
    return make_float4(pow_(c.x, 2.5f), pow_(c.y, 2.5f), pow_(c.z, 2.5f), c.w);
  };
  const float4 cur = tap(v);
  float4 o = make_float4(cur.x, cur.y, cur.z, 1.0f);
  if (interlaced != 0.0f) {
    const float uv_step_y = 1.0f / tsy;
    const float4 last = tap(v - uv_step_y), next = tap(v + uv_step_y);
    const float ix = 0.5f * (last.x + next.x), iy = 0.5f * (last.y + next.y), iz = 0.5f * (last.z + next.z);
    const float modulus = interlaced + 1.0f;
    const float field_offset = mod_glsl((float)(L.frame_count0 + z) + 0.0f, modulus);
    const float line_num_last = __builtin_floorf(v * tsy - kUnderHalf);
    const float wrong_field = mod_glsl(line_num_last + field_offset, modulus);
    o = make_float4(mix_rt(cur.x, ix, wrong_field), mix_rt(cur.y, iy, wrong_field), mix_rt(cur.z, iz, wrong_field), 1.0f);
  }
  // progressive source: modulus 1 makes wrong_field exactly 0, and cur + 0*(interp - cur) == cur
  store_rt(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

// Progressive source, 8-bit NEAREST clamp-to-edge input, 8-bit target (the shipped configuration):
// the stored byte is a function of the source byte alone - encode(pow(byte/255, 2.5)) - so the
// pass is a 256-entry byte map applied to the nearest texel.  Same results as k_royale_first.
template <int OUT_FMT>
__global__ void __launch_bounds__(256) k_royale_first_bytemap(const PassLaunch L) {
    __shared__ uint32_t map[256];
  RC_SRGB_LDS(lds, L);
  {
    const int t = threadIdx.y * 64 + threadIdx.x;
    const float lin = pow_((float)t * (1.0f / 255.0f), 2.5f);
    map[t] = OUT_FMT == FMT_SRGB8 ? srgb8(lin, &lds) : unorm8(lin);
  }
  __syncthreads();
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  // sample_nearest<., WRAP_EDGE>: texel index
  const int sx = clampi((int)__builtin_floorf(u * (float)L.in.w), 0, L.in.w - 1);
  const int sy = clampi((int)__builtin_floorf(v * (float)L.in.h), 0, L.in.h - 1);
  const uint32_t p = *reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z) + ((size_t)sy * L.in.w + sx) * 4);
  // alpha: the shader writes 1.0
  const uint32_t o = map[p & 255u] | (map[(p >> 8) & 255u] << 8) | (map[(p >> 16) & 255u] << 16) | 0xff000000u;
  *(reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z) + ((size_t)y * L.out_w + x)) = o;
  RC_TILE_LOOP_END
}

// The same at 1:1 (every target pixel's NEAREST texel is the texel under it - verified per geometry by
// k_first_identity with the sampler's operations): the byte map itself is computed once per geometry by
// k_first_bytemap_table, and the pass is a streaming copy through it, four pixels per thread.
struct FirstTables {
  uint32_t* map = nullptr;   // 256 entries
  bool usable = false;
  void release() {
    if (map) (void)hipFree(map);
    *this = FirstTables();
  }
};
template <int OUT_FMT>
__global__ void __launch_bounds__(256) k_first_bytemap_table(const PassLaunch L, uint32_t* map) {
  RC_SRGB_LDS(lds, L);
  const int t = (int)threadIdx.x;
  const float lin = pow_((float)t * (1.0f / 255.0f), 2.5f);
  map[t] = OUT_FMT == FMT_SRGB8 ? srgb8(lin, &lds) : unorm8(lin);
}
__global__ void __launch_bounds__(256) k_first_identity(const PassLaunch L, uint32_t* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool ok = true;
  for (int side = 0; side < 2; ++side) {
    if (i < L.out_w) ok = ok && rcstrip::near_tap(vary(L.plane[0], i, 0, side == 0), L.in.w) == i;
    if (i < L.out_h) ok = ok && rcstrip::near_tap(vary(L.plane[1], 0, i, side == 0), L.in.h) == i;
  }
  if (!ok) atomicOr(bad, 1u);
}
__global__ void __launch_bounds__(256) k_royale_first_copy(const PassLaunch L, const uint32_t* __restrict__ gmap) {
  __shared__ uint32_t map[256];
  map[threadIdx.x] = gmap[threadIdx.x];
  __syncthreads();
  const uint32_t quads_per_frame = (uint32_t)(L.out_w * L.out_h) >> 2, total = quads_per_frame * (uint32_t)L.n_frames;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t z = i / quads_per_frame, q = i - z * quads_per_frame;
    const uint4 p = reinterpret_cast<const uint4*>(frame_ptr(L.in, (int)z))[q];
    auto m = [&](uint32_t t) { return map[t & 255u] | (map[(t >> 8) & 255u] << 8) | (map[(t >> 16) & 255u] << 16) | 0xff000000u; };
    reinterpret_cast<uint4*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z)[q] = make_uint4(m(p.x), m(p.y), m(p.z), m(p.w));
  }
}
// the decoded value of the byte this pass would store for source byte t in an sRGB8 target (royale_first_byte_map)
__global__ void __launch_bounds__(256) k_first_decode_table(const PassLaunch L, float* dec) {
  RC_SRGB_LDS(lds, L);
  const int t = (int)threadIdx.x;
  const uint32_t byte = srgb8(pow_((float)t * (1.0f / 255.0f), 2.5f), &lds);
  dec[t] = k_srgb_decode[byte];
  reinterpret_cast<uint32_t*>(dec)[256 + t] = byte;
}
void buildFirstTables(const PassLaunch& L, hipStream_t s, FirstTables* T) {
  uint32_t* bad = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&T->map), 1024) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&bad), 4) == hipSuccess;
  uint32_t hbad = 1;
  if (ok) ok = hipMemsetAsync(bad, 0, 4, s) == hipSuccess;
  if (ok) {
    if (L.out_fmt == FMT_SRGB8) hipLaunchKernelGGL(k_first_bytemap_table<FMT_SRGB8>, dim3(1), dim3(256), rcd::srgb_lds_bytes(L), s, L, T->map);
    else hipLaunchKernelGGL(k_first_bytemap_table<FMT_RGBA8>, dim3(1), dim3(256), rcd::srgb_lds_bytes(L), s, L, T->map);
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_first_identity, dim3((n + 255) / 256), dim3(256), 0, s, L, bad);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (bad) (void)hipFree(bad);
  T->usable = ok && hbad == 0;
  if (!T->usable && T->map) {
    (void)hipFree(T->map);
    T->map = nullptr;
  }
}

// ------------------------------------------------------------------------------- P2 ------
// bloom-approx.glsl FS 14053-14184: the only live statement samples extra[0] at tex_uv.
template <class S0, class SO>
__global__ void __launch_bounds__(256) k_royale_bloom_approx(const PassLaunch L) {
  RC_SRGB_LDS_OF(lds, L, L.extra[0]);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  SO::put(L, z, x, y, S0::get(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds), &lds);
  RC_TILE_LOOP_END
}

// -------------------------------------------------------------------------- P3 / P4 ------
// blurs/blur9fast-*.glsl: tex2Dblur9fast 1496-1524; weights are compile-time constants there,
// folded by the host (royale_setup.cpp) the way the GL's compiler folds them.
template <class SI, class SO>
__global__ void __launch_bounds__(256) k_blur9(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float w12 = L.params[RPB_W12], w34 = L.params[RPB_W34], k12 = L.params[RPB_K12], k34 = L.params[RPB_K34];
  const float sum_inv = L.params[RPB_SUM_INV], dx = L.params[RPB_DX], dy = L.params[RPB_DY];
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 s0 = SI::get(L.in, img, u - k34 * dx, v - k34 * dy, &lds);
  const float4 s1 = SI::get(L.in, img, u - k12 * dx, v - k12 * dy, &lds);
  const float4 s2 = SI::get(L.in, img, u, v, &lds);
  const float4 s3 = SI::get(L.in, img, u + k12 * dx, v + k12 * dy, &lds);
  const float4 s4 = SI::get(L.in, img, u + k34 * dx, v + k34 * dy, &lds);
  // w34*s0 + w12*s1 + 1.0*s2 + w12*s3 + w34*s4 as the GL evaluates it: the centre term is a plain
  // addend (weight 1.0) and joins the running sum BEFORE the product that precedes it - the GL's
  // compiler rewrites fadd(x, ffma(a,b,c)) to ffma(a,b, x + c) - measured on float render targets:
  // ((((A + s2) + B) + D) + E) is the only association of all 5-leaf trees that matches
  const float sx = (((w34 * s0.x + s2.x) + w12 * s1.x) + w12 * s3.x) + w34 * s4.x;
  const float sy = (((w34 * s0.y + s2.y) + w12 * s1.y) + w12 * s3.y) + w34 * s4.y;
  const float sz = (((w34 * s0.z + s2.z) + w12 * s1.z) + w12 * s3.z) + w34 * s4.z;
  SO::put(L, z, x, y, make_float4(sx * sum_inv, sy * sum_inv, sz * sum_inv, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// -------------------------------------------------------------------------- P5 / P6 ------
// mask-resize-{vertical,horizontal}.glsl: 24-tap Lanczos-windowed sinc (phosphor-mask-resizing
// functions 2624-2994, USE_SINGLE_STATIC_LOOP).
template <bool VERTICAL>
__device__ __forceinline__ float4 sinc_tiled(const Tex& t, const uint8_t* img, float fixed_coord, float r_coord, float r_size, float dr,
                                            float magnification, float tile_size_uv_r, const SrgbLds* lds) {
  const float pi = 3.141592653589f, pi_over_lobes = pi / 3.0f;
  const float tiles_per_tex = 1.0f / tile_size_uv_r;
  const float curr = r_coord * r_size;
  const float prev = __builtin_floorf(curr - kUnderHalf) + 0.5f;
  const float first = prev - (24.0f / 2.0f - 1.0f);
  const float uv_wrap = first * dr;
  const float first_dist = curr - first;
  const float tile_uv_wrap = uv_wrap * tiles_per_tex;
  const float first_tile_uv = fractf(tile_uv_wrap) + (tile_uv_wrap < 0.0f ? 1.0f : 0.0f);
  const float tile_dr = dr * tiles_per_tex;
  float wsum[4] = {0.f, 0.f, 0.f, 0.f};
  float cx = 0.f, cy = 0.f, cz = 0.f;
  for (int i = 0; i < 24; i += 4) {
    float w[4];
    float4 s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float true_i = (float)i + (float)k;
      const float tile_uv_r = fractf(first_tile_uv + true_i * tile_dr);
      const float tex_uv_r = tile_uv_r * tile_size_uv_r;
      s[k] = VERTICAL ? sample_rt(t, img, fixed_coord, tex_uv_r, lds) : sample_rt(t, img, tex_uv_r, fixed_coord, lds);
      const float dist = magnification * __builtin_fabsf(first_dist - true_i);
      const float pi_dist = pi * dist;
      const float pdl = pi_over_lobes * dist;
      w[k] = minps(sin_(pi_dist) * sin_(pdl) / (pi_dist * pdl), 1.0f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cx += s[k].x * w[k]; cy += s[k].y * w[k]; cz += s[k].z * w[k];
      wsum[k] += w[k];
    }
  }
  const float total = (wsum[0] + wsum[2]) + (wsum[1] + wsum[3]);
  return make_float4(cx / total, cy / total, cz / total, 1.0f);
}

__global__ void __launch_bounds__(256) k_royale_mask_v(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  const float wu = vary(L.plane[0], x, y, lo), wv = vary(L.plane[1], x, y, lo);
  uint8_t* o = static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z;
  if (wv <= 2.0f) {  // mask_resize_num_tiles
    const float4 c = sinc_tiled<true>(L.extra[0], frame_ptr(L.extra[0], z), fractf(wu), fractf(wv), 64.0f, 1.0f / 64.0f,
                                      L.params[RP5_MAG_Y], 1.0f, &lds);
    store_rt(L, z, x, y, c, &lds);
  } else {  // discard: the target keeps its clear colour (0,0,0,0)
    *reinterpret_cast<uint32_t*>(o + ((size_t)y * L.out_w + x) * texel_bytes(L.out_fmt)) = 0u;
  }
  RC_TILE_LOOP_END
}

__global__ void __launch_bounds__(256) k_royale_mask_h(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  uint8_t* o = static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z;
  if (!(L.flags & RC_FLAG_UNDEF_VARYING_ZERO)) {
    // The fragment shader's keep/discard test reads a varying that its vertex shader never
    // writes; on the GL this engine is matched against, that discards every fragment.
    *reinterpret_cast<uint32_t*>(o + ((size_t)y * L.out_w + x) * 4) = 0u;
    continue;
  }
  const float wu = vary(L.plane[0], x, y, lo), wv = vary(L.plane[1], x, y, lo);
  const float4 c = sinc_tiled<false>(L.in, frame_ptr(L.in, z), fractf(wv), fractf(wu), (float)L.in.w, L.params[RP6_SRC_DX],
                                     L.params[RP6_MAG_X], L.params[RP6_TILE_SIZE_UV_X], &lds);
  store_rt(L, z, x, y, c, &lds);
  RC_TILE_LOOP_END
}

// ------------------------------------------------------------------------------- P7 ------
// scanlines-horizontal-apply-mask.glsl FS 10877-11030; sample_single_scanline_horizontal
// 5198-5241 with the Quilez weights (beam_horiz_filter 0) and linear-RGB mixing.
template <class SS>
__device__ __forceinline__ float scanline_h_1ch(const Tex& t, const uint8_t* img, float u, float v, float tsx, float tsy, float tix,
                                                float tiy, int ch, const SrgbLds* lds) {
  const float ctx = u * tsx, cty = v * tsy;
  const float phx = __builtin_floorf(ctx - kUnderHalf) + 0.5f;
  const float puv_x = phx * tix, puv_y = cty * tiy;
  const float xd = ctx - phx;
  const float w2 = xd * xd * xd * (xd * (xd * 6.0f - 15.0f) + 10.0f);
  const float wy = 1.0f - w2, wz = w2;
  const float dot = ((0.0f * 1.0f + wy * 1.0f) + wz * 1.0f) + 0.0f * 1.0f;
  const float fx = 0.0f / dot, fy = wy / dot, fz = wz / dot, fw = 0.0f / dot;
  const float4 c1 = SS::get(t, img, puv_x, puv_y, lds);
  const float4 c2 = SS::get(t, img, puv_x + tix, puv_y + 0.0f, lds);
  const float a1 = ch == 0 ? c1.x : (ch == 1 ? c1.y : c1.z);
  const float a2 = ch == 0 ? c2.x : (ch == 1 ? c2.y : c2.z);
  const float m = ((0.0f * fx + a1 * fy) + a2 * fz) + 0.0f * fw;
  return maxps(m, 0.0f);
}

// PHOSPHOR_BLOOM_FAKE tail (scanlines-horizontal-apply-mask-fake-bloom.glsl FS 10948-11027): every
// parameter is a compile-time constant in that file; operation order as the GL's compiler leaves it
// (constant factors of a product chain gathered, x + t*(y - x) for the run-time weight; float goldens).
__device__ __forceinline__ float fake_bloom_tail(float scan, float mask, float soft, float hal) {
  const float mask_amplify = 1.0f / (46.0f / 255.0f);
  const float undim = 1.0f / 0.5f, under = 0.8f, diffusion = 0.075f, contrast = 1.05f;
  const float ped = scan * mask;
  const float pe = ped * (undim * mask_amplify);
  const float ei = scan * undim;
  const float lerped = soft * (1.0f - 0.1f) + ei * 0.1f;
  const float approx = lerped * contrast;
  const float pbu = lerped * (contrast * under);
  const float amu = lerped * ((contrast * under) * mask_amplify);
  const float rt = (amu - 1.0f) / (amu - pbu);
  const float ratio = maxps(minps(maxps(rt, 0.0f), 1.0f), 0.0f);
  const float unclipped = pe + ratio * (approx - pe);
  return unclipped * (1.0f - diffusion) + hal * diffusion;
}

// FAKE: extra = PassPrev6 (VERTICAL_SCANLINES), PassPrev5 (BLOOM_APPROX), PassPrev3 (HALATION_BLUR)
template <class SI, class S0, class SO, bool FAKE>
__device__ __forceinline__ void scan_h_pixel(const PassLaunch& L, const SrgbLds& lds_, int x, int y, int z, bool lo) {
  const SrgbLds* ldsp = &lds_;
#define lds (*ldsp)
  const float vu = vary(L.plane[0], x, y, lo), vv = vary(L.plane[1], x, y, lo);
  const float twx = vu * L.params[RP7_TPS_X], twy = vv * L.params[RP7_TPS_Y];
  const float tux = fractf(twx * 0.5f) * 2.0f, tuy = fractf(twy * 0.5f) * 2.0f;
  const float mu = L.params[RP7_START_X] + tux * L.params[RP7_UVS_X], mv = L.params[RP7_START_Y] + tuy * L.params[RP7_UVS_Y];
  const float4 mask = SI::get(L.in, frame_ptr(L.in, z), mu, mv, &lds);
  float4 o = make_float4(0.f, 0.f, 0.f, 1.0f);
  // without the fake bloom: scan * 0 is 0 (or NaN, which every target format stores as 0): skip the scanline taps
  if (FAKE || mask.x != 0.0f || mask.y != 0.0f || mask.z != 0.0f) {
    const float su = vary(L.plane[2], x, y, lo), sv = vary(L.plane[3], x, y, lo);
    const Tex& scan = L.extra[0];
    const uint8_t* simg = frame_ptr(scan, z);
    const float tsx = L.params[RP7_SCAN_TW], tsy = L.params[RP7_SCAN_TH], tix = L.params[RP7_SCAN_TIX], tiy = L.params[RP7_SCAN_TIY];
    const float conv_x[3] = {0.1f, 0.3f, 0.5f};
    float sc[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) sc[ch] = scanline_h_1ch<S0>(scan, simg, su - conv_x[ch] * tix, sv - 0.0f, tsx, tsy, tix, tiy, ch, &lds);
    o = make_float4(sc[0] * mask.x, sc[1] * mask.y, sc[2] * mask.z, 1.0f);
    if (FAKE) {
      const float4 soft = S0::get(L.extra[1], frame_ptr(L.extra[1], z), vary(L.plane[4], x, y, lo), vary(L.plane[5], x, y, lo), &lds);
      const float4 hal = S0::get(L.extra[2], frame_ptr(L.extra[2], z), vary(L.plane[6], x, y, lo), vary(L.plane[7], x, y, lo), &lds);
      o = make_float4(fake_bloom_tail(sc[0], mask.x, soft.x, hal.x), fake_bloom_tail(sc[1], mask.y, soft.y, hal.y),
                      fake_bloom_tail(sc[2], mask.z, soft.z, hal.z), 1.0f);
    }
  }
  SO::put(L, z, x, y, o, &lds);
#undef lds
}

template <class SI, class S0, class SO, bool FAKE>
__global__ void __launch_bounds__(256) k_royale_scan_h(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  scan_h_pixel<SI, S0, SO, FAKE>(L, lds, x, y, z, lo);
  RC_TILE_LOOP_END
}

// ------------------------------------------------------------------- P7, strip form ------
// Separable geometry (royale_strip.h).  The six scanline taps of a pixel (two per channel, at the channel's
// convergence offset) are LINEAR taps of pass 1's target at texel centres: the first reads texels (x-1, x), the
// second (x, x+1) - verified per geometry - with per-column weights and per-column Quilez factors; vertically they
// share one row pair.  A thread walks kShRows rows of one column and filters each source row horizontally once
// (six values); rows whose mask texels are all zero in a wave (every row on a GL that discards pass 6's fragments)
// skip the scanline work like the general kernel does per pixel.
#ifndef RC_SH_ROWS
#define RC_SH_ROWS 16   // (8 -> 16, round 4: 9.5 -> 9.2 us per frame with the mask rendered)
#endif
constexpr int kShRows = RC_SH_ROWS;
enum { SH_WXA = 0, SH_WXB = 3, SH_FY = 6, SH_FZ = 9, SH_MX = 12, SH_COL_FIELDS = 13 };
enum { SH_Y0 = 0, SH_WY = 1, SH_MY = 2, SH_ROW_FIELDS = 4 };
struct ScanHTables {
  uint32_t* cols = nullptr;   // [SH_COL_FIELDS][2][W]
  uint32_t* rows = nullptr;   // [H][2][SH_ROW_FIELDS]
  bool usable = false;
  void release() {
    if (cols) (void)hipFree(cols);
    if (rows) (void)hipFree(rows);
    *this = ScanHTables();
  }
};
__global__ void __launch_bounds__(256) k_scanh_geometry(const PassLaunch L, uint32_t* cols, uint32_t* rows, uint32_t* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* P = L.params;
  const float tsx = P[RP7_SCAN_TW], tsy = P[RP7_SCAN_TH], tix = P[RP7_SCAN_TIX], tiy = P[RP7_SCAN_TIY];
  const Tex& scan = L.extra[0];
  uint32_t why = 0u;
  if (i < L.out_w)
    for (int side = 0; side < 2; ++side) {
      const bool lo = side == 0;
      const float su = vary(L.plane[2], i, 0, lo);
      const float conv_x[3] = {0.1f, 0.3f, 0.5f};
      for (int ch = 0; ch < 3; ++ch) {   // scanline_h_1ch's horizontal part
        const float u = su - conv_x[ch] * tix;
        const float ctx = u * tsx;
        const float phx = __builtin_floorf(ctx - kUnderHalf) + 0.5f;
        const float puv_x = phx * tix;
        const float xd = ctx - phx;
        const float w2 = xd * xd * xd * (xd * (xd * 6.0f - 15.0f) + 10.0f);
        const float wy = 1.0f - w2, wz = w2;
        const float dot = ((0.0f * 1.0f + wy * 1.0f) + wz * 1.0f) + 0.0f * 1.0f;
        const float fx = 0.0f / dot, fy = wy / dot, fz = wz / dot, fw = 0.0f / dot;
        const rcstrip::LinTap ta = rcstrip::lin_tap(puv_x, scan.w), tb = rcstrip::lin_tap(puv_x + tix, scan.w);
        if (ta.i0 != i - 1 || tb.i0 != i) why |= 1u;
        if (!(fx == 0.0f && fw == 0.0f)) why |= 2u;
        cols[((SH_WXA + ch) * 2 + side) * L.out_w + i] = f2bits(ta.w);
        cols[((SH_WXB + ch) * 2 + side) * L.out_w + i] = f2bits(tb.w);
        cols[((SH_FY + ch) * 2 + side) * L.out_w + i] = f2bits(fy);
        cols[((SH_FZ + ch) * 2 + side) * L.out_w + i] = f2bits(fz);
      }
      const float vu = vary(L.plane[0], i, 0, lo);
      const float tux = fractf(vu * P[RP7_TPS_X] * 0.5f) * 2.0f;
      cols[(SH_MX * 2 + side) * L.out_w + i] = (uint32_t)rcstrip::near_tap(P[RP7_START_X] + tux * P[RP7_UVS_X], L.in.w);
    }
  if (i < L.out_h)
    for (int side = 0; side < 2; ++side) {
      const bool lo = side == 0;
      uint32_t* r = rows + ((size_t)i * 2 + side) * SH_ROW_FIELDS;
      const float sv = vary(L.plane[3], 0, i, lo) - 0.0f;
      const float puv_y = (sv * tsy) * tiy;
      const rcstrip::LinTap t = rcstrip::lin_tap(puv_y + 0.0f, scan.h);
      if (t.i0 < i - 1 || t.i0 > i || rcstrip::lin_tap(puv_y, scan.h).i0 != t.i0) why |= 4u;
      r[SH_Y0] = (uint32_t)t.i0;
      r[SH_WY] = f2bits(t.w);
      const float vv = vary(L.plane[1], 0, i, lo);
      const float tuy = fractf(vv * P[RP7_TPS_Y] * 0.5f) * 2.0f;
      r[SH_MY] = (uint32_t)rcstrip::near_tap(P[RP7_START_Y] + tuy * P[RP7_UVS_Y], L.in.h);
      r[3] = 0u;
    }
  if (scan.w != L.out_w || scan.h != L.out_h) why |= 8u;
  if (why) atomicOr(bad, why);
}

template <class SO>
__global__ void __launch_bounds__(512) k_royale_scan_h_strip(const PassLaunch L, const uint32_t* __restrict__ cols, const uint32_t* __restrict__ rows) {
  RC_SRGB_LDS(lds, L);
  using MaskS = S<FMT_RGBA8, 0, WRAP_EDGE>;
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const rcstrip::StripGrid<kShRows> G(L.out_w, L.out_h, L.n_frames);
  const int W = G.W, H = G.H;
  const Tex& scan = L.extra[0];
  for (int strip = (int)blockIdx.x * 8 + wave; strip < G.total; strip += (int)gridDim.x * 8) {
    int z, xw, ys;
    G.locate(strip, &z, &xw, &ys);
    const int x = xw + lane;
    const bool live = x < W;
    const int xc = live ? x : W - 1;
    const int xmax = min(xw + 63, W - 1), ymax = min(ys + kShRows - 1, H - 1);
    const bool all_lo = rcd::lower_tri(xw, ymax, W, H), all_up = !rcd::lower_tri(xmax, ys, W, H);
    if (!all_lo && !all_up) {   // the quad's diagonal crosses this strip: per-pixel form
      if (live)
        for (int y = ys; y <= ymax; ++y) scan_h_pixel<MaskS, SrgbLinEdge, SO, false>(L, lds, x, y, z, rcd::lower_tri(x, y, W, H));
      continue;
    }
    const int side = all_lo ? 0 : 1;
    float wxa[3], wxb[3], fy[3], fz[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      wxa[ch] = bits2f(cols[((SH_WXA + ch) * 2 + side) * W + xc]);
      wxb[ch] = bits2f(cols[((SH_WXB + ch) * 2 + side) * W + xc]);
      fy[ch] = bits2f(cols[((SH_FY + ch) * 2 + side) * W + xc]);
      fz[ch] = bits2f(cols[((SH_FZ + ch) * 2 + side) * W + xc]);
    }
    const int mx = (int)cols[(SH_MX * 2 + side) * W + xc];
    const uint32_t* mimg = reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z));
    const uint32_t* simg = reinterpret_cast<const uint32_t*>(frame_ptr(scan, z));
    const int xl = clampi(xc - 1, 0, scan.w - 1), xm = clampi(xc, 0, scan.w - 1), xr = clampi(xc + 1, 0, scan.w - 1);
    // horizontally filtered source rows: ha = tap 1 (texels x-1, x), hb = tap 2 (texels x, x+1), per channel
    auto hrow = [&](int r, float* ha, float* hb) {
      const uint32_t* p = simg + clampi(r, 0, scan.h - 1) * scan.w;
      const uint32_t tl = p[xl], tm = p[xm], tr = p[xr];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float dl = lds.dec[(tl >> (8 * ch)) & 255u], dm = lds.dec[(tm >> (8 * ch)) & 255u], dr = lds.dec[(tr >> (8 * ch)) & 255u];
        ha[ch] = fma_(wxa[ch], dm - dl, dl);
        hb[ch] = fma_(wxb[ch], dr - dm, dm);
      }
    };
    float a0[3], b0[3], a1[3], b1[3];   // the row pair in use: source rows `have` and `have + 1`
    int have = -1000;
#pragma unroll 1
    for (int k = 0; k < kShRows; ++k) {
      const int y = ys + k;
      if (y >= H) break;
      const uint32_t* rr = rows + ((size_t)y * 2 + side) * SH_ROW_FIELDS;
      const int y0 = (int)rr[SH_Y0], my = (int)rr[SH_MY];
      const float wy = bits2f(rr[SH_WY]);
      const uint32_t mt = mimg[my * L.in.w + mx];
      const bool any = (mt & 0x00ffffffu) != 0u;
      uint32_t px = 0xff000000u;   // scan * 0 (or NaN): stored as 0, alpha 1
      if (__builtin_amdgcn_ballot_w64(any) != 0ull) {   // wave-uniform
        if (y0 != have) {
          if (y0 == have + 1) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
              a0[ch] = a1[ch];
              b0[ch] = b1[ch];
            }
          } else {
            hrow(y0, a0, b0);
          }
          hrow(y0 + 1, a1, b1);
          have = y0;
        }
        if (live && any) {
          const float k255 = 1.0f / 255.0f;
          const float mask[3] = {(float)(mt & 255u) * k255, (float)((mt >> 8) & 255u) * k255, (float)((mt >> 16) & 255u) * k255};
          float o[3];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float c1 = fma_(wy, a1[ch] - a0[ch], a0[ch]), c2 = fma_(wy, b1[ch] - b0[ch], b0[ch]);
            const float m = c1 * fy[ch] + c2 * fz[ch];   // ((0*fx + a1*fy) + a2*fz) + 0*fw with fx = fw = 0
            o[ch] = maxps(m, 0.0f) * mask[ch];
          }
          SO::put(L, z, x, y, make_float4(o[0], o[1], o[2], 1.0f), &lds);
          continue;
        }
      }
      if (live) reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z)[(size_t)y * W + x] = px;
    }
  }
}

// ---- P7, strip form, two pixels per lane (royale_strip2.h): columns x and x + 64 of a 128-column band; the lerps and the
// Quilez mix of the pixel pair run as packed float operations.  Per source row a lane decodes the three texels x - 1, x,
// x + 1 of each of its pixels once and keeps the horizontally filtered pair of the current row pair (six values per channel
// pair).  A strip the quad's diagonal crosses is rendered once per triangle, each pixel stored by the pass of its own triangle.
#ifndef RC_SH_WAVES
#define RC_SH_WAVES 8   // (with the mask rendered, one lane: 5 waves 8.85 us per frame against 9.5 - but three such workgroups fill a CU's LDS and the engine's second lane finds no room: 18.5 k frames/s against 19.3 k on two lanes)
#endif
constexpr int kSh2Waves = RC_SH_WAVES;   // waves per workgroup (the 53 KB of tables are per workgroup)
__global__ void __launch_bounds__(kSh2Waves * 64) k_royale_scan_h_strip2(const PassLaunch L, const uint32_t* __restrict__ cols, const uint32_t* __restrict__ rows) {
  using namespace rcstrip2;
  extern __shared__ uint32_t rc_dyn_lds_[];
  strip2_load_tables(rc_dyn_lds_, L, true);
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int W = L.out_w, H = L.out_h;
  const int bands = (W + 127) >> 7, rss = (H + kShRows - 1) / kShRows, per_frame = bands * rss, total = per_frame * L.n_frames;
  const Tex& scan = L.extra[0];
  const int sw = scan.w, sh = scan.h;
  for (int strip = (int)blockIdx.x * kSh2Waves + wave; strip < total; strip += (int)gridDim.x * kSh2Waves) {
    const int z = strip / per_frame, rem = strip - z * per_frame, rs = rem / bands;
    const int xw = (rem - rs * bands) << 7, ys = rs * kShRows;
    const int xa = xw + lane, xb = xa + 64;
    const bool live_a = xa < W, live_b = xb < W;
    const int xca = live_a ? xa : W - 1, xcb = live_b ? xb : W - 1;
    const int xmax = min(xw + 127, W - 1), ymax = min(ys + kShRows - 1, H - 1);
    const bool all_lo = rcd::lower_tri(xw, ymax, W, H), all_up = !rcd::lower_tri(xmax, ys, W, H);
    const uint8_t* mimg = frame_ptr(L.in, z);
    const uint8_t* simg = frame_ptr(scan, z);
    const __amdgpu_buffer_rsrc_t r_out = frame_rsrc(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z, W, H);
    const int tri_a = (2 * xa + 1) * H, tri_b = (2 * xb + 1) * H;
    // texels x - 1, x, x + 1 of both pixels (k_scanh_geometry: every channel's two taps are (x - 1, x) and (x, x + 1))
    const uint32_t la = (uint32_t)clampi(xca - 1, 0, sw - 1) * 4u, ma = (uint32_t)clampi(xca, 0, sw - 1) * 4u, ra_ = (uint32_t)clampi(xca + 1, 0, sw - 1) * 4u;
    const uint32_t lb = (uint32_t)clampi(xcb - 1, 0, sw - 1) * 4u, mb = (uint32_t)clampi(xcb, 0, sw - 1) * 4u, rb_ = (uint32_t)clampi(xcb + 1, 0, sw - 1) * 4u;
    for (int side = all_up ? 1 : 0; side <= (all_lo ? 0 : 1); ++side) {
      const bool mixed = !all_lo && !all_up;
      v2f wxa[3], wxb[3], fy[3], fz[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        wxa[ch] = v2f{bits2f(cols[((SH_WXA + ch) * 2 + side) * W + xca]), bits2f(cols[((SH_WXA + ch) * 2 + side) * W + xcb])};
        wxb[ch] = v2f{bits2f(cols[((SH_WXB + ch) * 2 + side) * W + xca]), bits2f(cols[((SH_WXB + ch) * 2 + side) * W + xcb])};
        fy[ch] = v2f{bits2f(cols[((SH_FY + ch) * 2 + side) * W + xca]), bits2f(cols[((SH_FY + ch) * 2 + side) * W + xcb])};
        fz[ch] = v2f{bits2f(cols[((SH_FZ + ch) * 2 + side) * W + xca]), bits2f(cols[((SH_FZ + ch) * 2 + side) * W + xcb])};
      }
      const uint32_t mxa = cols[(SH_MX * 2 + side) * W + xca] * 4u, mxb = cols[(SH_MX * 2 + side) * W + xcb] * 4u;
      // horizontally filtered source row: ha = tap 1 (texels x - 1, x), hb = tap 2 (texels x, x + 1), per channel, both pixels;
      // the six texels of the next source row are fetched ahead of their use
      auto hfetch = [&](int r, uint32_t* q) __attribute__((always_inline)) {
        const uint8_t* p = simg + (size_t)(clampi(r, 0, sh - 1) * sw) * 4u;
        q[0] = *reinterpret_cast<const uint32_t*>(p + la);
        q[1] = *reinterpret_cast<const uint32_t*>(p + ma);
        q[2] = *reinterpret_cast<const uint32_t*>(p + ra_);
        q[3] = *reinterpret_cast<const uint32_t*>(p + lb);
        q[4] = *reinterpret_cast<const uint32_t*>(p + mb);
        q[5] = *reinterpret_cast<const uint32_t*>(p + rb_);
      };
      auto hfilter = [&](const uint32_t* q, v2f* ha, v2f* hb) __attribute__((always_inline)) {
        {
          const v2f dl = {dec_byte<0>(q[0]), dec_byte<0>(q[3])}, dm = {dec_byte<0>(q[1]), dec_byte<0>(q[4])}, dr = {dec_byte<0>(q[2]), dec_byte<0>(q[5])};
          ha[0] = fma2(wxa[0], dm - dl, dl);
          hb[0] = fma2(wxb[0], dr - dm, dm);
        }
        {
          const v2f dl = {dec_byte<1>(q[0]), dec_byte<1>(q[3])}, dm = {dec_byte<1>(q[1]), dec_byte<1>(q[4])}, dr = {dec_byte<1>(q[2]), dec_byte<1>(q[5])};
          ha[1] = fma2(wxa[1], dm - dl, dl);
          hb[1] = fma2(wxb[1], dr - dm, dm);
        }
        {
          const v2f dl = {dec_byte<2>(q[0]), dec_byte<2>(q[3])}, dm = {dec_byte<2>(q[1]), dec_byte<2>(q[4])}, dr = {dec_byte<2>(q[2]), dec_byte<2>(q[5])};
          ha[2] = fma2(wxa[2], dm - dl, dl);
          hb[2] = fma2(wxb[2], dr - dm, dm);
        }
      };
      uint32_t nq[6];
      int nq_row = -1000;
      v2f a0[3], b0[3], a1[3], b1[3];   // the row pair in use: source rows `have` and `have + 1`
      int have = -1000;
      // the mask texel of a row is fetched one step ahead
      uint32_t nma, nmb;
      {
        const uint32_t* r0p = rows + ((size_t)ys * 2 + side) * SH_ROW_FIELDS;
        const uint8_t* p = mimg + (size_t)(r0p[SH_MY] * (uint32_t)L.in.w) * 4u;
        nma = *reinterpret_cast<const uint32_t*>(p + mxa);
        nmb = *reinterpret_cast<const uint32_t*>(p + mxb);
      }
#pragma unroll 1
      for (int y = ys; y <= ymax; ++y) {
        const uint32_t* rr = rows + ((size_t)y * 2 + side) * SH_ROW_FIELDS;
        const int y0 = (int)rr[SH_Y0];
        const float wy = bits2f(rr[SH_WY]);
        const uint32_t mta = nma, mtb = nmb;
        {
          const uint32_t* rn = rows + ((size_t)min(y + 1, H - 1) * 2 + side) * SH_ROW_FIELDS;
          const uint8_t* p = mimg + (size_t)(rn[SH_MY] * (uint32_t)L.in.w) * 4u;
          nma = *reinterpret_cast<const uint32_t*>(p + mxa);
          nmb = *reinterpret_cast<const uint32_t*>(p + mxb);
        }
        uint32_t pa = 0xff000000u, pb = 0xff000000u;   // scan * 0 (or NaN): stored as 0, alpha 1
        // rows whose mask texels are all zero across the wave skip the scanline work, as the per-pixel form does per pixel
        if (__builtin_amdgcn_ballot_w64(((mta | mtb) & 0x00ffffffu) != 0u) != 0ull) {
          if (y0 != have) {
            if (y0 == have + 1) {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) {
                a0[ch] = a1[ch];
                b0[ch] = b1[ch];
              }
            } else {
              uint32_t q[6];
              hfetch(y0, q);
              hfilter(q, a0, b0);
            }
            if (nq_row != y0 + 1) hfetch(y0 + 1, nq);
            hfilter(nq, a1, b1);
            have = y0;
            nq_row = y0 + 2;
            hfetch(nq_row, nq);
          }
          const float k255 = 1.0f / 255.0f;
          v2f o[3];
          {
            const v2f c1 = fma2(splat2(wy), a1[0] - a0[0], a0[0]), c2 = fma2(splat2(wy), b1[0] - b0[0], b0[0]);
            const v2f m = c1 * fy[0] + c2 * fz[0];   // ((0*fx + c1*fy) + c2*fz) + 0*fw with fx = fw = 0
            const v2f mask = v2f{(float)(mta & 255u), (float)(mtb & 255u)} * k255;
            o[0] = v2f{maxps(m.x, 0.0f), maxps(m.y, 0.0f)} * mask;
          }
          {
            const v2f c1 = fma2(splat2(wy), a1[1] - a0[1], a0[1]), c2 = fma2(splat2(wy), b1[1] - b0[1], b0[1]);
            const v2f m = c1 * fy[1] + c2 * fz[1];
            const v2f mask = v2f{(float)((mta >> 8) & 255u), (float)((mtb >> 8) & 255u)} * k255;
            o[1] = v2f{maxps(m.x, 0.0f), maxps(m.y, 0.0f)} * mask;
          }
          {
            const v2f c1 = fma2(splat2(wy), a1[2] - a0[2], a0[2]), c2 = fma2(splat2(wy), b1[2] - b0[2], b0[2]);
            const v2f m = c1 * fy[2] + c2 * fz[2];
            const v2f mask = v2f{(float)((mta >> 16) & 255u), (float)((mtb >> 16) & 255u)} * k255;
            o[2] = v2f{maxps(m.x, 0.0f), maxps(m.y, 0.0f)} * mask;
          }
          srgb8_pack2(o, &pa, &pb);
        }
        bool sa = live_a, sb = live_b;
        if (mixed) {
          const int tri_y = (2 * y + 1) * W;
          sa = sa && (tri_y <= tri_a) == (side == 0);
          sb = sb && (tri_y <= tri_b) == (side == 0);
        }
        if (sa) __builtin_amdgcn_raw_buffer_store_b32(pa, r_out, xa * 4, y * W * 4, 0);
        if (sb) __builtin_amdgcn_raw_buffer_store_b32(pb, r_out, xb * 4, y * W * 4, 0);
      }
    }
  }
}

void buildScanHTables(const PassLaunch& L, hipStream_t s, ScanHTables* T) {
  uint32_t* bad = nullptr;
  bool ok = hipMalloc(reinterpret_cast<void**>(&T->cols), (size_t)SH_COL_FIELDS * 2 * L.out_w * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T->rows), (size_t)L.out_h * 2 * SH_ROW_FIELDS * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&bad), 4) == hipSuccess;
  uint32_t hbad = 1;
  if (ok) ok = hipMemsetAsync(bad, 0, 4, s) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_scanh_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, T->cols, T->rows, bad);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (bad) (void)hipFree(bad);
  T->usable = ok && hbad == 0;
  RC_LOG_DEBUG("crt-royale scanlines-horizontal " + std::to_string(L.out_w) + "x" + std::to_string(L.out_h) + ": strip form " + (ok && hbad == 0 ? "in use" : "not usable (flags " + std::to_string(hbad) + ")"));
  if (!T->usable) {
    if (T->cols) (void)hipFree(T->cols);
    if (T->rows) (void)hipFree(T->rows);
    *T = ScanHTables();
  }
}

// ------------------------------------------------------------------------------- P8 ------
// brightpass.glsl FS 14610-14663; extra[0] = PassPrev4Texture.
template <class SI, class S0, class SO>
__device__ __forceinline__ void brightpass_pixel(const PassLaunch& L, const SrgbLds& lds_, int x, int y, int z, bool lo) {
  const SrgbLds* ldsp = &lds_;
#define lds (*ldsp)
  const float4 idim = SI::get(L.in, frame_ptr(L.in, z), vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo), &lds);
  float4 o = make_float4(0.f, 0.f, 0.f, 1.0f);
  // brightpass = intensity_dim * ratio: a zero (or NaN-producing) input stores 0 whatever the ratio
  if (idim.x != 0.0f || idim.y != 0.0f || idim.z != 0.0f) {
    const float4 blur = S0::get(L.extra[0], frame_ptr(L.extra[0], z), vary(L.plane[2], x, y, lo), vary(L.plane[3], x, y, lo), &lds);
    const float cw = L.params[RP8_CENTER_WEIGHT], mask_amplify = L.params[RP8_MASK_AMPLIFY];
    const float in3[3] = {idim.x, idim.y, idim.z}, bl3[3] = {blur.x, blur.y, blur.z};
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float intensity = in3[c] * 2.0f * mask_amplify * 1.0f;
      const float pba = 1.0f * bl3[c];
      const float max_area = maxps(pba - cw * intensity, 0.0f);
      const float area_under = 0.8f * max_area;
      // the GL's compiler gathers the constant factors of in*undim*mask_amplify*contrast*underestimate
      // into one (all compile-time constants in this file): in * ((2*mask_amplify)*0.8)
      const float int_under = in3[c] * ((2.0f * mask_amplify) * 0.8f);
      const float ratio_temp = ((1.0f - area_under) / int_under - 1.0f) / (cw - 1.0f);
      out[c] = in3[c] * clampf(ratio_temp, 0.0f, 1.0f);
    }
    o = make_float4(out[0], out[1], out[2], 1.0f);
  }
  SO::put(L, z, x, y, o, &lds);
#undef lds
}

template <class SI, class S0, class SO>
__global__ void __launch_bounds__(256) k_royale_brightpass(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  brightpass_pixel<SI, S0, SO>(L, lds, x, y, z, lo);
  RC_TILE_LOOP_END
}

// the per-channel arithmetic of brightpass.glsl, as in brightpass_pixel
__device__ __forceinline__ float brightpass_channel(float in, float bl, float cw, float mask_amplify) {
  const float intensity = in * 2.0f * mask_amplify * 1.0f;
  const float pba = 1.0f * bl;
  const float max_area = maxps(pba - cw * intensity, 0.0f);
  const float area_under = 0.8f * max_area;
  const float int_under = in * ((2.0f * mask_amplify) * 0.8f);
  const float ratio_temp = ((1.0f - area_under) / int_under - 1.0f) / (cw - 1.0f);
  return in * clampf(ratio_temp, 0.0f, 1.0f);
}

// ------------------------------------------------------------------- P8, strip form ------
// Separable geometry (royale_strip.h): the NEAREST tap of pass 7's target is the texel (ix(x), iy(y)); the LINEAR tap
// of the 320x240 halation blur is magnified, so a thread walking a column keeps the two horizontally filtered blur
// rows of the current row pair and refilters only when the pair moves on (every fourth target row or so).
#ifndef RC_BP_ROWS
#define RC_BP_ROWS 32   // (8 -> 32, round 4: 6.8 -> 6.4 us per frame with the mask rendered)
#endif
constexpr int kBpRows = RC_BP_ROWS;
enum { BP_IX = 0, BP_BX0 = 1, BP_BWX = 2, BP_COL_FIELDS = 3 };
enum { BP_IY = 0, BP_BY0 = 1, BP_BWY = 2, BP_ROW_FIELDS = 4 };
struct BpTables {
  uint32_t* cols = nullptr;
  uint32_t* rows = nullptr;
  bool usable = false;
  void release() {
    if (cols) (void)hipFree(cols);
    if (rows) (void)hipFree(rows);
    *this = BpTables();
  }
};
__global__ void __launch_bounds__(256) k_brightpass_geometry(const PassLaunch L, uint32_t* cols, uint32_t* rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < L.out_w)
    for (int side = 0; side < 2; ++side) {
      const rcstrip::LinTap t = rcstrip::lin_tap(vary(L.plane[2], i, 0, side == 0), L.extra[0].w);
      cols[(BP_IX * 2 + side) * L.out_w + i] = (uint32_t)rcstrip::near_tap(vary(L.plane[0], i, 0, side == 0), L.in.w);
      cols[(BP_BX0 * 2 + side) * L.out_w + i] = (uint32_t)t.i0;
      cols[(BP_BWX * 2 + side) * L.out_w + i] = f2bits(t.w);
    }
  if (i < L.out_h)
    for (int side = 0; side < 2; ++side) {
      uint32_t* r = rows + ((size_t)i * 2 + side) * BP_ROW_FIELDS;
      const rcstrip::LinTap t = rcstrip::lin_tap(vary(L.plane[3], 0, i, side == 0), L.extra[0].h);
      r[BP_IY] = (uint32_t)rcstrip::near_tap(vary(L.plane[1], 0, i, side == 0), L.in.h);
      r[BP_BY0] = (uint32_t)t.i0;
      r[BP_BWY] = f2bits(t.w);
      r[3] = 0u;
    }
}

template <class SO>
__global__ void __launch_bounds__(512) k_royale_brightpass_strip(const PassLaunch L, const uint32_t* __restrict__ cols, const uint32_t* __restrict__ rows) {
  RC_SRGB_LDS(lds, L);
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const rcstrip::StripGrid<kBpRows> G(L.out_w, L.out_h, L.n_frames);
  const int W = G.W, H = G.H;
  const Tex& blur = L.extra[0];
  const float cw = L.params[RP8_CENTER_WEIGHT], mask_amplify = L.params[RP8_MASK_AMPLIFY];
  for (int strip = (int)blockIdx.x * 8 + wave; strip < G.total; strip += (int)gridDim.x * 8) {
    int z, xw, ys;
    G.locate(strip, &z, &xw, &ys);
    const int x = xw + lane;
    const bool live = x < W;
    const int xc = live ? x : W - 1;
    const int xmax = min(xw + 63, W - 1), ymax = min(ys + kBpRows - 1, H - 1);
    const bool all_lo = rcd::lower_tri(xw, ymax, W, H), all_up = !rcd::lower_tri(xmax, ys, W, H);
    if (!all_lo && !all_up) {   // the quad's diagonal crosses this strip: per-pixel form
      if (live)
        for (int y = ys; y <= ymax; ++y) brightpass_pixel<SrgbNearEdge, SrgbLinEdge, SO>(L, lds, x, y, z, rcd::lower_tri(x, y, W, H));
      continue;
    }
    const int side = all_lo ? 0 : 1;
    const int ix = (int)cols[(BP_IX * 2 + side) * W + xc], bx0 = (int)cols[(BP_BX0 * 2 + side) * W + xc];
    const float bwx = bits2f(cols[(BP_BWX * 2 + side) * W + xc]);
    const int bxa = clampi(bx0, 0, blur.w - 1), bxb = clampi(bx0 + 1, 0, blur.w - 1);
    const uint32_t* iimg = reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z));
    const uint32_t* bimg = reinterpret_cast<const uint32_t*>(frame_ptr(blur, z));
    auto hrow = [&](int r, float* h) {
      const uint32_t* p = bimg + clampi(r, 0, blur.h - 1) * blur.w;
      const uint32_t ta = p[bxa], tb = p[bxb];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float a = lds.dec[(ta >> (8 * ch)) & 255u], b = lds.dec[(tb >> (8 * ch)) & 255u];
        h[ch] = fma_(bwx, b - a, a);
      }
    };
    float h0[3], h1[3];
    int have = -1000;
#pragma unroll 1
    for (int k = 0; k < kBpRows; ++k) {
      const int y = ys + k;
      if (y >= H) break;
      const uint32_t* rr = rows + ((size_t)y * 2 + side) * BP_ROW_FIELDS;
      const int iy = (int)rr[BP_IY], by0 = (int)rr[BP_BY0];
      const float bwy = bits2f(rr[BP_BWY]);
      const uint32_t it = iimg[iy * L.in.w + ix];
      const bool any = (it & 0x00ffffffu) != 0u;   // decode(0) = 0: a zero input stores 0 whatever the ratio
      if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
        if (by0 != have) {
          if (by0 == have + 1) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) h0[ch] = h1[ch];
          } else {
            hrow(by0, h0);
          }
          hrow(by0 + 1, h1);
          have = by0;
        }
        if (live && any) {
          float o[3];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float in = lds.dec[(it >> (8 * ch)) & 255u];
            const float bl = fma_(bwy, h1[ch] - h0[ch], h0[ch]);
            o[ch] = brightpass_channel(in, bl, cw, mask_amplify);
          }
          SO::put(L, z, x, y, make_float4(o[0], o[1], o[2], 1.0f), &lds);
          continue;
        }
      }
      if (live) reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z)[(size_t)y * W + x] = 0xff000000u;
    }
  }
}

// ---- P8, strip form, two pixels per lane (royale_strip2.h): columns x and x + 64 of a 128-column band, so the blur-ratio
// algebra runs as packed float operations over the pixel pair.  The two IEEE divisions of a channel take div_safe_'s
// sequence: the first divides 1 - 0.8 max_area, in [0.2, 1], by input * 8.86.. (a decoded byte: 0 or >= 3e-4; a zero input
// makes the quotient NaN or infinite either way and the channel stores 0 either way: 0 * clamp(..) with clamp(NaN) = 0 as in
// brightpass_pixel), the second divides by the launch constant center_weight - 1.  A strip the quad's diagonal crosses
// is rendered once per triangle, each pixel stored by the pass of its own triangle.
__device__ __forceinline__ v2f brightpass_pair(v2f in, v2f bl, float cw, float mask_amplify) {
  using namespace rcstrip2;
  const v2f intensity = in * (2.0f * mask_amplify);             // in * 2 * mask_amplify * 1: doubling is exact, so (in * 2) * m and in * (2 m) round the same product once
  const v2f area = bl - cw * intensity;                         // 1 * blur - center_weight * intensity
  const v2f max_area = {__builtin_fmaxf(area.x, 0.0f), __builtin_fmaxf(area.y, 0.0f)};
  const v2f area_under = 0.8f * max_area;
  const v2f int_under = in * ((2.0f * mask_amplify) * 0.8f);
  const v2f q = div_safe2(1.0f - area_under, int_under);
  const v2f ratio = div_safe2(q - 1.0f, splat2(cw - 1.0f));
  const v2f cl = {__builtin_fminf(__builtin_fmaxf(ratio.x, 0.0f), 1.0f), __builtin_fminf(__builtin_fmaxf(ratio.y, 0.0f), 1.0f)};
  return in * cl;
}

#ifndef RC_BP_WAVES
#define RC_BP_WAVES 8
#endif
constexpr int kBp2Waves = RC_BP_WAVES;
__global__ void __launch_bounds__(kBp2Waves * 64) k_royale_brightpass_strip2(const PassLaunch L, const uint32_t* __restrict__ cols, const uint32_t* __restrict__ rows) {
  using namespace rcstrip2;
  extern __shared__ uint32_t rc_dyn_lds_[];
  strip2_load_tables(rc_dyn_lds_, L, true);
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int W = L.out_w, H = L.out_h;
  const int bands = (W + 127) >> 7, rss = (H + kBpRows - 1) / kBpRows, per_frame = bands * rss, total = per_frame * L.n_frames;
  const Tex& blur = L.extra[0];
  const float cw = L.params[RP8_CENTER_WEIGHT], mask_amplify = L.params[RP8_MASK_AMPLIFY];
  for (int strip = (int)blockIdx.x * kBp2Waves + wave; strip < total; strip += (int)gridDim.x * kBp2Waves) {
    const int z = strip / per_frame, rem = strip - z * per_frame, rs = rem / bands;
    const int xw = (rem - rs * bands) << 7, ys = rs * kBpRows;
    const int xa = xw + lane, xb = xa + 64;
    const bool live_a = xa < W, live_b = xb < W;
    const int xca = live_a ? xa : W - 1, xcb = live_b ? xb : W - 1;
    const int xmax = min(xw + 127, W - 1), ymax = min(ys + kBpRows - 1, H - 1);
    const bool all_lo = rcd::lower_tri(xw, ymax, W, H), all_up = !rcd::lower_tri(xmax, ys, W, H);
    const uint8_t* iimg = frame_ptr(L.in, z);
    const uint8_t* bimg = frame_ptr(blur, z);
    const __amdgpu_buffer_rsrc_t r_out = frame_rsrc(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z, W, H);
    const int tri_a = (2 * xa + 1) * H, tri_b = (2 * xb + 1) * H;
    for (int side = all_up ? 1 : 0; side <= (all_lo ? 0 : 1); ++side) {
      const bool mixed = !all_lo && !all_up;
      const uint32_t ixa = cols[(BP_IX * 2 + side) * W + xca] * 4u, ixb = cols[(BP_IX * 2 + side) * W + xcb] * 4u;
      const int ba = (int)cols[(BP_BX0 * 2 + side) * W + xca], bb = (int)cols[(BP_BX0 * 2 + side) * W + xcb];
      const uint32_t ba0 = (uint32_t)clampi(ba, 0, blur.w - 1) * 4u, ba1 = (uint32_t)clampi(ba + 1, 0, blur.w - 1) * 4u;
      const uint32_t bb0 = (uint32_t)clampi(bb, 0, blur.w - 1) * 4u, bb1 = (uint32_t)clampi(bb + 1, 0, blur.w - 1) * 4u;
      const v2f bwx = {bits2f(cols[(BP_BWX * 2 + side) * W + xca]), bits2f(cols[(BP_BWX * 2 + side) * W + xcb])};
      auto hrow = [&](int r, v2f* h) __attribute__((always_inline)) {   // the sampler's horizontal lerp of blur row r (clamped), both pixels
        const uint8_t* p = bimg + (size_t)(clampi(r, 0, blur.h - 1) * blur.w) * 4u;
        const uint32_t a0 = *reinterpret_cast<const uint32_t*>(p + ba0), a1 = *reinterpret_cast<const uint32_t*>(p + ba1);
        const uint32_t b0 = *reinterpret_cast<const uint32_t*>(p + bb0), b1 = *reinterpret_cast<const uint32_t*>(p + bb1);
        const v2f l0 = {dec_byte<0>(a0), dec_byte<0>(b0)}, r0 = {dec_byte<0>(a1), dec_byte<0>(b1)};
        const v2f l1 = {dec_byte<1>(a0), dec_byte<1>(b0)}, r1 = {dec_byte<1>(a1), dec_byte<1>(b1)};
        const v2f l2 = {dec_byte<2>(a0), dec_byte<2>(b0)}, r2 = {dec_byte<2>(a1), dec_byte<2>(b1)};
        h[0] = fma2(bwx, r0 - l0, l0);
        h[1] = fma2(bwx, r1 - l1, l1);
        h[2] = fma2(bwx, r2 - l2, l2);
      };
      v2f h0[3], h1[3], hd[3];
      int have = -1000;
      // the NEAREST tap of a row is fetched one step ahead
      const uint32_t* r0p = rows + ((size_t)ys * 2 + side) * BP_ROW_FIELDS;
      uint32_t nia = *reinterpret_cast<const uint32_t*>(iimg + (size_t)(r0p[BP_IY] * (uint32_t)L.in.w) * 4u + ixa);
      uint32_t nib = *reinterpret_cast<const uint32_t*>(iimg + (size_t)(r0p[BP_IY] * (uint32_t)L.in.w) * 4u + ixb);
#pragma unroll 1
      for (int y = ys; y <= ymax; ++y) {
        const uint32_t* rr = rows + ((size_t)y * 2 + side) * BP_ROW_FIELDS;
        const int by0 = (int)rr[BP_BY0];
        const float bwy = bits2f(rr[BP_BWY]);
        const uint32_t ia = nia, ib = nib;
        {
          const uint32_t* rn = rows + ((size_t)min(y + 1, H - 1) * 2 + side) * BP_ROW_FIELDS;
          const uint8_t* p = iimg + (size_t)(rn[BP_IY] * (uint32_t)L.in.w) * 4u;
          nia = *reinterpret_cast<const uint32_t*>(p + ixa);
          nib = *reinterpret_cast<const uint32_t*>(p + ixb);
        }
        uint32_t pa = 0xff000000u, pb = 0xff000000u;
        // decode(0) = 0: a zero input stores 0 whatever the ratio (as the per-pixel form); skip rows that are black across the wave
        if (__builtin_amdgcn_ballot_w64(((ia | ib) & 0x00ffffffu) != 0u) != 0ull) {
          if (by0 != have) {
            if (by0 == have + 1) {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) h0[ch] = h1[ch];
            } else {
              hrow(by0, h0);
            }
            hrow(by0 + 1, h1);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) hd[ch] = h1[ch] - h0[ch];
            have = by0;
          }
          v2f o[3];
          o[0] = brightpass_pair(v2f{dec_byte<0>(ia), dec_byte<0>(ib)}, fma2(splat2(bwy), hd[0], h0[0]), cw, mask_amplify);
          o[1] = brightpass_pair(v2f{dec_byte<1>(ia), dec_byte<1>(ib)}, fma2(splat2(bwy), hd[1], h0[1]), cw, mask_amplify);
          o[2] = brightpass_pair(v2f{dec_byte<2>(ia), dec_byte<2>(ib)}, fma2(splat2(bwy), hd[2], h0[2]), cw, mask_amplify);
          srgb8_pack2(o, &pa, &pb);
        }
        bool sa = live_a, sb = live_b;
        if (mixed) {
          const int tri_y = (2 * y + 1) * W;
          sa = sa && (tri_y <= tri_a) == (side == 0);
          sb = sb && (tri_y <= tri_b) == (side == 0);
        }
        if (sa) __builtin_amdgcn_raw_buffer_store_b32(pa, r_out, xa * 4, y * W * 4, 0);
        if (sb) __builtin_amdgcn_raw_buffer_store_b32(pb, r_out, xb * 4, y * W * 4, 0);
      }
    }
  }
}

void buildBpTables(const PassLaunch& L, hipStream_t s, BpTables* T) {
  bool ok = hipMalloc(reinterpret_cast<void**>(&T->cols), (size_t)BP_COL_FIELDS * 2 * L.out_w * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T->rows), (size_t)L.out_h * 2 * BP_ROW_FIELDS * 4) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_brightpass_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, T->cols, T->rows);
    ok = hipGetLastError() == hipSuccess;
  }
  T->usable = ok;
  if (!ok) {
    if (T->cols) (void)hipFree(T->cols);
    if (T->rows) (void)hipFree(T->rows);
    *T = BpTables();
  }
}

// ------------------------------------------------------------------------------ P11 ------
// geometry-aa-last-pass.glsl FS 5451-5531 (flat geometry path), get_border_dim_factor 5250.
template <class SI, class SO, bool MIP>
__device__ __forceinline__ void last_pixel(const PassLaunch& L, const SrgbLds& lds_, int x, int y, int z, bool lo) {
  const SrgbLds* ldsp = &lds_;
#define lds (*ldsp)
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  const float* P = L.params;
  const float lcd_gamma = P[1], osx = P[37], osy = P[38], border_size = P[39], border_darkness = P[40], border_compress = P[41];
  const float vsix = 1.0f / tsx, vsiy = 1.0f / tsy;
  const float geom_aspect_x = P[RP11_ASPECT_X], geom_aspect_y = P[RP11_ASPECT_Y];
  const float inv_gamma = 1.0f / lcd_gamma;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float fu = u * (tsx * vsix), fv = v * (tsy * vsiy);
  const float vu = (fu - 0.5f) / osx + 0.5f, vv = (fv - 0.5f) / osy + 0.5f;
  const float tu = vu * (tsx * vsix), tv = vv * (tsy * vsiy);
  float4 c;
  if (MIP) {
    // mipmap_input (crt-royale-fake-bloom's last pass): the sample coordinate at the quad's pixels as this pixel's
    // triangle extrapolates them; on a 1:1 pass the LOD is 0 or a hair above it (rc_device.h, mip-mapped sampling)
    const int x0 = x & ~1, y0 = y & ~1;
    const int qx[4] = {x0, x0 + 1, x, x}, qy[4] = {y, y, y0, y0 + 1};
    float qu[4], qv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      qu[k] = ((vary(L.plane[0], qx[k], qy[k], lo) * (tsx * vsix) - 0.5f) / osx + 0.5f) * (tsx * vsix);
      qv[k] = ((vary(L.plane[1], qx[k], qy[k], lo) * (tsy * vsiy) - 0.5f) / osy + 0.5f) * (tsy * vsiy);
    }
    const float lod = lod_from_quad(L.in, qu[0], qu[1], qv[0], qv[1], qu[2], qu[3], qv[2], qv[3]);
    c = lod > 0.0f ? sample_mip(L.in, z, tu, tv, lod, &lds) : SI::get(L.in, frame_ptr(L.in, z), tu, tv, &lds);
  } else {
    c = SI::get(L.in, frame_ptr(L.in, z), tu, tv, &lds);
  }
  const float ex = minps(vu, 1.0f - vu) * geom_aspect_x, ey = minps(vv, 1.0f - vv) * geom_aspect_y;
  const float bx = maxps(border_size - ex, 0.0f), by = maxps(border_size - ey, 0.0f);
  // Away from the border bx = by = 0: pen = sqrt(0)/size = 0, esc = 1, pow(1, d) = exp2(0*d) = 1 exactly
  // (log2's polynomial is y*P with y = 0 at mantissa 1), and min(1 * max(1, compress), 1) = 1.
  // (border_size = 0 would make pen 0/0: that case takes the full path)
  float f = 1.0f;
  if (bx != 0.0f || by != 0.0f || !(border_size > 0.0f)) {
    const float pen = __builtin_sqrtf(bx * bx + by * by) / border_size;
    const float esc = maxps(1.0f - pen, 0.0f);
    f = minps(pow_(esc, border_darkness) * maxps(1.0f, border_compress), 1.0f);
  }
  // the three output-gamma pows: red and green as a packed pair (rc_vecmath.h), blue alone
  const v2f rg = exp2_v<v2f>(log2_v(v2f{c.x * f, c.y * f}) * inv_gamma);
  SO::put(L, z, x, y, make_float4(rg.x, rg.y, pow_(c.z * f, inv_gamma), 1.0f), &lds);
#undef lds
}

template <class SI, class SO, bool MIP>
__global__ void __launch_bounds__(256) k_royale_last(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  RC_TILE_LOOP_BEGIN
  last_pixel<SI, SO, MIP>(L, lds, x, y, z, lo);
  RC_TILE_LOOP_END
}

// ------------------------------------------------------------------ P11, strip form ------
// Separable geometry (royale_strip.h), flat mode: the one LINEAR tap of a pixel decomposes into a per-column pair
// (first texel, weight) and a per-row pair; a thread walks kLastRows rows of one column, filters each source row
// horizontally once, and evaluates two target rows per step so that their six output-gamma pows run as three
// packed pairs (rc_vecmath.h).  The border factor is 1 away from the border (k_royale_last explains why), which
// k_last_geometry records per column / row as "border distance is zero".
#ifndef RC_LAST_ROWS
#define RC_LAST_ROWS 16   // (8 -> 16, round 4: 7.6 -> 7.4 us per frame)
#endif
constexpr int kLastRows = RC_LAST_ROWS;   // (even, at most 32: a lane's uncertain rows are a bit mask)
enum { LS_X0 = 0, LS_WX = 1, LS_BX = 2, LS_COL_FIELDS = 3 };
enum { LS_Y0 = 0, LS_WY = 1, LS_BY = 2, LS_ROW_FIELDS = 4 };
struct LastTables {
  uint32_t* cols = nullptr;   // [LS_COL_FIELDS][2][W]
  uint32_t* rows = nullptr;   // [H][2][LS_ROW_FIELDS]
  float4* gamma_tab = nullptr;   // certified expansion of the output-gamma pow (below), or null
  bool usable = false;
  void release() {
    if (cols) (void)hipFree(cols);
    if (rows) (void)hipFree(rows);
    if (gamma_tab) (void)hipFree(gamma_tab);
    *this = LastTables();
  }
};

// ---- the output gamma from a table, with a proven bound ------------------------------------------------------------
// Away from the border the pass stores unorm8(G(c)) for the sampled colour c, G(c) = exp2(log2(c) / lcd_gamma) in the GL's
// polynomials (~40 float operations, three per pixel).  G is smooth, so the strip kernel evaluates it from a table of
// log-spaced nodes: 32 per octave (the top five mantissa bits of c select the node, its colour c0 is the bucket's midpoint)
// from 2^-20 - below which G(c) * 255 < 0.47 and the byte is 0 for certain - up to 1.  Per node: the quadratic through T = G(c0)
// (the float the exact code computes) with slope and half curvature from the closed form, written in the colour itself,
// a0 + c (a1 + c a2) - since round 4 scaled by 255, so that the table delivers y ~ 255 G, the value the byte is rounded from - and a bound E
// on |fl(min(G(c), 1) * 255) - fma(c, fma(c, a2, a1), a0)|, the exact code's own pre-rounding value against the very expression the strip kernel
// evaluates, that k_last_gamma_err MEASURES over EVERY float of the bucket (2^18 floats; 168 M exact evaluations per table): a bound by
// exhaustion, not by sampling.  With r = rint(y) the exact code's byte is r whenever |y - r| + E < 0.5, and the kernel re-renders the
// few other pixels with the exact per-pixel code.
// (kLastTabBits0 / kLastTabShift / kLastTabNodes / kLastLdsTab and last_gamma_byte: royale_common.h)
__device__ __forceinline__ float last_gamma(float c, float inv_gamma) { return exp2_(log2_(c) * inv_gamma); }   // = the strips' packed form per component
__host__ __device__ __forceinline__ float last_node_colour(int n) {
  return n == kLastTabNodes - 1 ? 1.0f : bits2f(kLastTabBits0 + ((uint32_t)n << kLastTabShift) + (1u << (kLastTabShift - 1)));
}
// phase 0: T; phase 1: R = the largest error over the node's floats
__global__ void __launch_bounds__(256) k_last_gamma_err(float inv_gamma, float4* tab, int phase) {
  const int n = (int)blockIdx.y;
  const float c0 = last_node_colour(n);
  if (phase == 0) {
    // the node's quadratic written in the colour itself (the strip kernel then needs neither the node colour nor a subtraction):
    // a0 + c (a1 + c a2) with a2 = G''/2, a1 = G' - 2 a2 c0, a0 = T - G' c0 + a2 c0^2 from the host's G', G''/2 (tab[n].y, .z)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      // (... times 255: the table delivers the value the byte is rounded from, y = 255 G, in two fmas)
      const double T = (double)last_gamma(c0, inv_gamma), X = (double)c0, g1 = (double)tab[n].y, g2 = (double)tab[n].z;
      tab[n].x = (float)(255.0 * (T - g1 * X + g2 * X * X));
      tab[n].y = (float)(255.0 * (g1 - 2.0 * g2 * X));
      tab[n].z = (float)(255.0 * g2);
    }
    return;
  }
  const float4 e = tab[n];
  const uint32_t lo = kLastTabBits0 + ((uint32_t)n << kLastTabShift), hi = n == kLastTabNodes - 1 ? lo : lo + (1u << kLastTabShift) - 1u;
  uint32_t worst = 0u;
  for (uint32_t i = lo + blockIdx.x * 256u + threadIdx.x; i <= hi; i += gridDim.x * 256u) {
    const float c = bits2f(i);
    // against what the exact code rounds to its byte: fl(min(G, 1) * 255) (rc_device.h unorm8; G > 0 here)
    const float g = last_gamma(c, inv_gamma);
    const double err = fabs((double)((g > 1.0f ? 1.0f : g) * 255.0f) - (double)fma_(c, fma_(c, e.z, e.y), e.x));
    worst = max(worst, f2bits(__double2float_ru(err)));
  }
  if (worst) atomicMax(reinterpret_cast<uint32_t*>(&tab[n].w), worst);   // non-negative floats order like their bits
}
// What a record keeps is not the bound E but the threshold of last_gamma_byte's test, |y - r| < 0.5 - E, taken down so that its own
// roundings stay on the safe side (a node whose bound leaves no room keeps 0: every lookup fails over to the exact code)
__global__ void __launch_bounds__(256) k_last_gamma_finish(float4* tab) {
  const int n = (int)(blockIdx.x * 256 + threadIdx.x);
  if (n >= kLastTabNodes) return;
  const float thr = 0.5f - (tab[n].w * 1.000001f + 1e-7f);
  tab[n].w = thr > 1e-6f ? bits2f(f2bits(thr) - 2u) : 0.0f;
}

__global__ void __launch_bounds__(256) k_last_geometry(const PassLaunch L, uint32_t* cols, uint32_t* rows, uint32_t* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float tsx = (float)L.in.w, tsy = (float)L.in.h;
  const float* P = L.params;
  const float osx = P[37], osy = P[38], border_size = P[39];
  const float vsix = 1.0f / tsx, vsiy = 1.0f / tsy;
  uint32_t why = 0u;
  if (i < L.out_w)
    for (int side = 0; side < 2; ++side) {
      const float u = vary(L.plane[0], i, 0, side == 0);
      const float fu = u * (tsx * vsix);
      const float vu = (fu - 0.5f) / osx + 0.5f;
      const rcstrip::LinTap t = rcstrip::lin_tap(vu * (tsx * vsix), L.in.w);
      const float ex = minps(vu, 1.0f - vu) * P[RP11_ASPECT_X];
      cols[(LS_X0 * 2 + side) * L.out_w + i] = (uint32_t)t.i0;
      cols[(LS_WX * 2 + side) * L.out_w + i] = f2bits(t.w);
      cols[(LS_BX * 2 + side) * L.out_w + i] = f2bits(maxps(border_size - ex, 0.0f));
    }
  if (i < L.out_h)
    for (int side = 0; side < 2; ++side) {
      const float v = vary(L.plane[1], 0, i, side == 0);
      const float fv = v * (tsy * vsiy);
      const float vv = (fv - 0.5f) / osy + 0.5f;
      const rcstrip::LinTap t = rcstrip::lin_tap(vv * (tsy * vsiy), L.in.h);
      const float ey = minps(vv, 1.0f - vv) * P[RP11_ASPECT_Y];
      uint32_t* r = rows + ((size_t)i * 2 + side) * LS_ROW_FIELDS;
      if (t.i0 < i - 1 || t.i0 > i) why |= 1u;   // the strip keeps rows y-1 .. y+2 of a row pair
      r[LS_Y0] = (uint32_t)t.i0;
      r[LS_WY] = f2bits(t.w);
      r[LS_BY] = f2bits(maxps(border_size - ey, 0.0f));
      r[3] = 0u;
    }
  if (!(border_size > 0.0f)) why |= 2u;   // border_size = 0 makes the penetration 0/0 everywhere: general form
  if (why) atomicOr(bad, why);
}

#ifndef RC_LAST_WAVES
#define RC_LAST_WAVES 8
#endif
constexpr int kLastWaves = RC_LAST_WAVES;
constexpr int kLastList = 64 + 64 * kLastRows;   // entries of a wave's list of uncertain pixels: up to 63 carried over + what a strip can add
constexpr uint32_t kLastLdsList = (kLastLdsTab + (uint32_t)kLastTabNodes * 16u + 15u) & ~15u;   // behind the gamma table
template <class SO>
__global__ void __launch_bounds__(kLastWaves * 64) k_royale_last_strip(const PassLaunch L, const uint32_t* __restrict__ cols, const uint32_t* __restrict__ rows,
                                                         const float4* __restrict__ gamma_tab) {
  RC_SRGB_LDS(lds, L);
  if ((uint32_t)(uintptr_t)(RC_AS3 uint32_t*)rc_dyn_lds_ != 0u) __builtin_trap();   // the tables are addressed by absolute LDS offsets
  if (gamma_tab) {   // uniform
    for (int i = (int)threadIdx.x; i < kLastTabNodes; i += kLastWaves * 64) reinterpret_cast<float4*>(rc_dyn_lds_ + kLastLdsTab / 4)[i] = gamma_tab[i];
    __syncthreads();
  }
  const int lane = (int)threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const rcstrip::StripGrid<kLastRows> G(L.out_w, L.out_h, L.n_frames);
  const int W = G.W, H = G.H, Win = L.in.w, Hin = L.in.h;
  const float* P = L.params;
  const float inv_gamma = 1.0f / P[1], border_size = P[39], border_darkness = P[40], border_compress = P[41];
  // Pixels the gamma table cannot certify (0.3 % of them) are rendered with the exact per-pixel form - SIXTY-FOUR AT A TIME: a wave
  // collects them in its list in LDS (positions from a ballot, as pass 1 does) and renders a full wave of them whenever the list
  // holds that many, the remainder when it is done.  (Until round 4 every strip ended in a divergent loop that ran the exact form
  // for its one or two uncertain pixels with 62 lanes idle: 0.8 such loops per strip, a quarter of the kernel's instructions.)
  uint32_t* my_list = rc_dyn_lds_ + kLastLdsList / 4 + wave * kLastList;
  uint32_t n_listed = 0u;   // wave-uniform (lane 0 takes part in every strip and holds the valid copy: see k_royale_scan_v_tab)
  auto drain = [&](bool all) __attribute__((always_inline)) {   // (every lane active)
    uint32_t n = __builtin_amdgcn_readfirstlane(n_listed);
    while (n >= 64u || (all && n > 0u)) {
      const uint32_t take = n < 64u ? n : 64u, first = n - take;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS stores, in order
      if ((uint32_t)lane < take) {
        const uint32_t id = my_list[first + (uint32_t)lane];
        const uint32_t row = id / (uint32_t)W;
        const int px = (int)(id - row * (uint32_t)W), pz = (int)(row / (uint32_t)H), py = (int)(row - (uint32_t)pz * (uint32_t)H);
        last_pixel<SrgbLinEdge, SO, false>(L, lds, px, py, pz, rcd::lower_tri(px, py, W, H));
      }
      asm volatile("" ::: "memory");
      n = first;
    }
    n_listed = n;
  };
  auto strip_body = [&](int z, int xw, int ys, int x) __attribute__((always_inline)) {
    const int xmax = min(xw + 63, W - 1), ymax = min(ys + kLastRows - 1, H - 1);
    const bool all_lo = rcd::lower_tri(xw, ymax, W, H), all_up = !rcd::lower_tri(xmax, ys, W, H);
    if (!all_lo && !all_up) {   // the quad's diagonal crosses this strip: per-pixel form
      for (int y = ys; y <= ymax; ++y) last_pixel<SrgbLinEdge, SO, false>(L, lds, x, y, z, rcd::lower_tri(x, y, W, H));
      return;
    }
    const int side = all_lo ? 0 : 1;
    const int x0 = (int)cols[(LS_X0 * 2 + side) * W + x];
    const float wx = bits2f(cols[(LS_WX * 2 + side) * W + x]), bx = bits2f(cols[(LS_BX * 2 + side) * W + x]);
    const int xa = clampi(x0, 0, Win - 1), xb = clampi(x0 + 1, 0, Win - 1);
    // the gamma table serves pixels away from the border (border factor exactly 1): strips whose columns all are
    const bool tab_cols = gamma_tab != nullptr && __builtin_amdgcn_ballot_w64(bx != 0.0f) == 0ull;
    const uint32_t* img = reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z));
    // the sampler's horizontal lerp of a source row, three channels; the row's two texels are fetched a step ahead of their use
    auto fetch = [&](int r, uint32_t* ta, uint32_t* tb) {
      const uint32_t* p = img + clampi(r, 0, Hin - 1) * Win;
      *ta = p[xa];
      *tb = p[xb];
    };
    auto hfilter = [&](uint32_t ta, uint32_t tb, float* h) {   // (the decode table sits at LDS offset 0: royale_strip2.h dec_byte)
      {
        const float a = rcstrip2::dec_byte_plain<0>(ta), b = rcstrip2::dec_byte_plain<0>(tb);
        h[0] = fma_(wx, b - a, a);
      }
      {
        const float a = rcstrip2::dec_byte_plain<1>(ta), b = rcstrip2::dec_byte_plain<1>(tb);
        h[1] = fma_(wx, b - a, a);
      }
      {
        const float a = rcstrip2::dec_byte_plain<2>(ta), b = rcstrip2::dec_byte_plain<2>(tb);
        h[2] = fma_(wx, b - a, a);
      }
    };
    float w0[3], w1[3], w2[3], w3[3];   // rows y-1, y, y+1, y+2 of the current row pair
    uint32_t n2a, n2b, n3a, n3b;        // texels of rows y+1, y+2, in flight
    {
      uint32_t ta, tb, ua, ub;
      fetch(ys - 1, &ta, &tb);
      fetch(ys, &ua, &ub);
      fetch(ys + 1, &n2a, &n2b);
      fetch(ys + 2, &n3a, &n3b);
      hfilter(ta, tb, w0);
      hfilter(ua, ub, w1);
    }
    uint32_t failed = 0u;   // bit k: row ys + k of this column is not certain from the table
#pragma unroll
    for (int k = 0; k < kLastRows; k += 2) {
      const int y = ys + k;
      if (y >= H) break;
      hfilter(n2a, n2b, w2);
      hfilter(n3a, n3b, w3);
      if (k + 2 < kLastRows) {
        fetch(y + 3, &n2a, &n2b);
        fetch(y + 4, &n3a, &n3b);
      }
      const uint32_t* ra = rows + ((size_t)y * 2 + side) * LS_ROW_FIELDS;
      const uint32_t* rb = rows + ((size_t)min(y + 1, H - 1) * 2 + side) * LS_ROW_FIELDS;
      const bool a_up = (int)ra[LS_Y0] == y - 1, b_up = (int)rb[LS_Y0] == y;   // the pair starts one row above the target row
      const float wya = bits2f(ra[LS_WY]), wyb = bits2f(rb[LS_WY]), bya = bits2f(ra[LS_BY]), byb = bits2f(rb[LS_BY]);
      float ca[3], cb[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float la = a_up ? w0[ch] : w1[ch], ha = a_up ? w1[ch] : w2[ch];
        const float lb = b_up ? w1[ch] : w2[ch], hb = b_up ? w2[ch] : w3[ch];
        ca[ch] = fma_(wya, ha - la, la);
        cb[ch] = fma_(wyb, hb - lb, lb);
      }
      if (tab_cols && bya == 0.0f && byb == 0.0f) {   // uniform: both rows away from the border
        bool fa = false, fb = false;
        const uint32_t pa = 0xff000000u | last_gamma_byte(ca[0], &fa) | (last_gamma_byte(ca[1], &fa) << 8) | (last_gamma_byte(ca[2], &fa) << 16);
        const uint32_t pb = 0xff000000u | last_gamma_byte(cb[0], &fb) | (last_gamma_byte(cb[1], &fb) << 8) | (last_gamma_byte(cb[2], &fb) << 16);
        uint32_t* out = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z);
        if (!fa) out[(size_t)y * W + x] = pa; else failed |= 1u << k;
        if (y + 1 < H) {
          if (!fb) out[(size_t)(y + 1) * W + x] = pb; else failed |= 2u << k;
        }
      } else {
        // border dimming (get_border_dim_factor, as k_royale_last): 1 unless the pixel is within border_size of an edge
        auto dim = [&](float by) -> float {
          float f = 1.0f;
          if (bx != 0.0f || by != 0.0f) {
            const float pen = __builtin_sqrtf(bx * bx + by * by) / border_size;
            const float esc = maxps(1.0f - pen, 0.0f);
            f = minps(pow_(esc, border_darkness) * maxps(1.0f, border_compress), 1.0f);
          }
          return f;
        };
        const float fa = dim(bya), fb = dim(byb);
        // six output-gamma pows as three packed pairs
        const v2f rg_a = exp2_v<v2f>(log2_v(v2f{ca[0] * fa, ca[1] * fa}) * inv_gamma);
        const v2f rg_b = exp2_v<v2f>(log2_v(v2f{cb[0] * fb, cb[1] * fb}) * inv_gamma);
        const v2f bb = exp2_v<v2f>(log2_v(v2f{ca[2] * fa, cb[2] * fb}) * inv_gamma);
        SO::put(L, z, x, y, make_float4(rg_a.x, rg_a.y, bb.x, 1.0f), &lds);
        if (y + 1 < H) SO::put(L, z, x, y + 1, make_float4(rg_b.x, rg_b.y, bb.y, 1.0f), &lds);
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        w0[ch] = w2[ch];
        w1[ch] = w3[ch];
      }
    }
    // the pixels the table could not certify go to the wave's list (pass 1's scheme: positions from the ballot, no atomics)
    while (true) {
      const uint64_t any = __builtin_amdgcn_ballot_w64(failed != 0u);
      if (any == 0ull) break;
      const uint32_t n = (uint32_t)__builtin_popcountll(any);
      const uint32_t cur = __builtin_amdgcn_readfirstlane(n_listed);   // (lane 0's copy: taken before the branch below)
      if (cur + n > (uint32_t)kLastList) {   // (never expected: more than kLastList - 63 uncertain pixels in one strip) in place, as before
        if (failed) {
          const int k = __builtin_ctz(failed);
          failed &= failed - 1u;
          last_pixel<SrgbLinEdge, SO, false>(L, lds, x, ys + k, z, side == 0);
        }
        continue;
      }
      if (failed) {
        const int k = __builtin_ctz(failed);
        failed &= failed - 1u;
        const uint32_t pos = cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
        my_list[pos] = (uint32_t)((z * H + ys + k) * W + x);
      }
      n_listed = cur + n;
    }
  };
  for (int strip = (int)blockIdx.x * kLastWaves + wave; strip < G.total; strip += (int)gridDim.x * kLastWaves) {
    int z, xw, ys;
    G.locate(strip, &z, &xw, &ys);
    const int x = xw + lane;
    if (x < W) strip_body(z, xw, ys, x);
    drain(false);
  }
  drain(true);
}

// the gamma table for 1 / lcd_gamma into `tab` (kLastTabNodes records): slope and half curvature from the closed form in double
// precision (any coefficients are valid: the bound is measured against what is stored), node values and bounds on the device.
// `h` is the host staging of the asynchronous copy: it must outlive the stream's work (the caller synchronises).
bool fillLastGammaTable(float inv_gamma, float4* tab, hipStream_t s, std::vector<float4>* h) {
  const double g = (double)inv_gamma;
  h->resize((size_t)kLastTabNodes);
  for (int n = 0; n < kLastTabNodes; ++n) {
    const double c0 = (double)last_node_colour(n);
    (*h)[(size_t)n] = make_float4(0.0f, (float)(g * std::pow(c0, g - 1.0)), (float)(0.5 * g * (g - 1.0) * std::pow(c0, g - 2.0)), 0.0f);
  }
  if (hipMemcpyAsync(tab, h->data(), h->size() * sizeof(float4), hipMemcpyHostToDevice, s) != hipSuccess) return false;
  hipLaunchKernelGGL(k_last_gamma_err, dim3(1, kLastTabNodes), dim3(256), 0, s, inv_gamma, tab, 0);
  hipLaunchKernelGGL(k_last_gamma_err, dim3(16, kLastTabNodes), dim3(256), 0, s, inv_gamma, tab, 1);
  hipLaunchKernelGGL(k_last_gamma_finish, dim3((kLastTabNodes + 255) / 256), dim3(256), 0, s, tab);
  return hipGetLastError() == hipSuccess;
}

void buildLastTables(const PassLaunch& L, hipStream_t s, LastTables* T) {
  uint32_t* bad = nullptr;   // [0]: the strip form does not apply, [1]: the gamma table does not
  bool ok = hipMalloc(reinterpret_cast<void**>(&T->cols), (size_t)LS_COL_FIELDS * 2 * L.out_w * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&T->rows), (size_t)L.out_h * 2 * LS_ROW_FIELDS * 4) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&bad), 8) == hipSuccess;
  uint32_t hbad[2] = {1, 1};
  if (ok) ok = hipMemsetAsync(bad, 0, 8, s) == hipSuccess;
  if (ok) {
    const int n = L.out_w > L.out_h ? L.out_w : L.out_h;
    hipLaunchKernelGGL(k_last_geometry, dim3((n + 255) / 256), dim3(256), 0, s, L, T->cols, T->rows, bad);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(hbad, bad, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (bad) (void)hipFree(bad);
  T->usable = ok && hbad[0] == 0;
  RC_LOG_DEBUG("crt-royale last pass " + std::to_string(L.out_w) + "x" + std::to_string(L.out_h) + ": strip flags " + std::to_string(hbad[0]) + ", gamma table flags " + std::to_string(hbad[1]));
  if (T->usable && hbad[1] == 0 && hipMalloc(reinterpret_cast<void**>(&T->gamma_tab), kLastTabNodes * sizeof(float4)) == hipSuccess) {
    const float inv_gamma = 1.0f / L.params[1];
    std::vector<float4> h((size_t)kLastTabNodes);
    bool tok = fillLastGammaTable(inv_gamma, T->gamma_tab, s, &h);
    if (tok) {
      tok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;   // h must outlive the copy
    }
    if (!tok) {
      (void)hipFree(T->gamma_tab);
      T->gamma_tab = nullptr;
    }
  }
  if (!T->usable) {
    if (T->cols) (void)hipFree(T->cols);
    if (T->rows) (void)hipFree(T->rows);
    *T = LastTables();
  }
}

}  // namespace

namespace rck {
// The output-gamma table of 1 / lcd_gamma on the current device, for the general last-pass kernel (pass_royale_last_general.hip):
// built on first use, at most 16 kept (least recently used released after a device synchronisation); nullptr on failure.
const float4* royale_last_gamma_table(float inv_gamma, hipStream_t s) {
  struct Entry {
    float4* tab;
    uint64_t last_use;
  };
  static std::mutex mu;
  static std::map<std::pair<int, uint32_t>, Entry> cache;
  static uint64_t clock = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(dev, rcd::f2bits(inv_gamma));
  auto it = cache.find(key);
  if (it != cache.end()) {
    it->second.last_use = ++clock;
    return it->second.tab;
  }
  if (cache.size() >= 16) {
    auto victim = cache.begin();
    for (auto c = cache.begin(); c != cache.end(); ++c)
      if (c->second.last_use < victim->second.last_use) victim = c;
    (void)hipDeviceSynchronize();
    if (victim->second.tab) (void)hipFree(victim->second.tab);
    cache.erase(victim);
  }
  float4* tab = nullptr;
  std::vector<float4> h;
  if (hipMalloc(reinterpret_cast<void**>(&tab), kLastTabNodes * sizeof(float4)) == hipSuccess &&
      !(fillLastGammaTable(inv_gamma, tab, s, &h) && hipStreamSynchronize(s) == hipSuccess)) {
    (void)hipFree(tab);
    tab = nullptr;
  }
  cache[key] = Entry{tab, ++clock};   // (a failure is remembered too: the exact form runs instead)
  return tab;
}

#define RC_LAUNCH(fn, kernel)                                         \
  hipError_t fn(const PassLaunch& L, hipStream_t s) {                 \
    hipLaunchKernelGGL(kernel, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);      \
    return hipGetLastError();                                         \
  }
// KernelEntry::byte_map of pass 0: at 1:1 on a progressive 8-bit NEAREST source with an sRGB8 target the pass stores
// map[byte] per channel and alpha 255 (k_royale_first_copy); then d_dec256[byte] = the target's decode of map[byte], and the pass
// need not run for consumers that read the source through that table (rcd::kFoldedTableWords words: the mapped bytes follow).
bool royale_first_byte_map(const PassLaunch& L, hipStream_t s, float* d_dec256) {
  const bool bytes_in = (L.in.fmt == FMT_RGBX8 || L.in.fmt == FMT_RGBA8) && !L.in.linear && L.in.wrap == WRAP_EDGE && L.in.n_levels <= 1;
  if (!bytes_in || L.params[RP0_INTERLACED] != 0.0f || (L.flags & RC_FLAG_GENERAL_ONLY) || L.out_fmt != FMT_SRGB8 || L.in.w != L.out_w ||
      L.in.h != L.out_h || !rcstrip::separable(L, 0, 1))
    return false;
  static std::mutex mu;
  static std::map<rcstrip::GeoKey, rcstrip::GeoCached<FirstTables>> cache;
  if (!rcstrip::geo_tables<FirstTables>(L, s, mu, cache, buildFirstTables)) return false;   // (k_first_identity: every pixel's texel is the one under it)
  hipLaunchKernelGGL(k_first_decode_table, dim3(1), dim3(256), rcd::srgb_lds_bytes(L), s, L, d_dec256);
  return hipGetLastError() == hipSuccess;
}
hipError_t launch_royale_first(const PassLaunch& L, hipStream_t s) {
  const bool bytes_in = (L.in.fmt == FMT_RGBX8 || L.in.fmt == FMT_RGBA8) && !L.in.linear && L.in.wrap == WRAP_EDGE;
  if (bytes_in && L.params[RP0_INTERLACED] == 0.0f && !(L.flags & RC_FLAG_GENERAL_ONLY)) {
    // 1:1, rows of whole 16-byte groups, 16-byte aligned frames: the streaming form
    if ((L.out_fmt == FMT_SRGB8 || L.out_fmt == FMT_RGBA8) && L.in.w == L.out_w && L.in.h == L.out_h && ((L.out_w * L.out_h) & 3) == 0 &&
        (L.in.frame_stride & 15u) == 0 && (reinterpret_cast<uintptr_t>(L.in.base) & 15u) == 0 && rcstrip::separable(L, 0, 1)) {
      static std::mutex mu;
      static std::map<rcstrip::GeoKey, rcstrip::GeoCached<FirstTables>> cache;
      if (const auto T = rcstrip::geo_tables<FirstTables>(L, s, mu, cache, buildFirstTables)) {
        const long quads = (long)(L.out_w * L.out_h / 4) * L.n_frames;
        const long blocks = (quads + 255) / 256;
        hipLaunchKernelGGL(k_royale_first_copy, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, L, T->map);
        return hipGetLastError();
      }
    }
    if (L.out_fmt == FMT_SRGB8) {
      hipLaunchKernelGGL(k_royale_first_bytemap<FMT_SRGB8>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
      return hipGetLastError();
    }
    if (L.out_fmt == FMT_RGBA8) {
      hipLaunchKernelGGL(k_royale_first_bytemap<FMT_RGBA8>, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(k_royale_first, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
RC_LAUNCH(launch_royale_mask_v, k_royale_mask_v)
RC_LAUNCH(launch_royale_mask_h, k_royale_mask_h)

#define GO(...)                                                                 \
  do {                                                                          \
    hipLaunchKernelGGL((__VA_ARGS__), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);          \
    return hipGetLastError();                                                   \
  } while (0)
using OutS = St<FMT_SRGB8>;

hipError_t launch_royale_bloom_approx(const PassLaunch& L, hipStream_t s) {
  if (SrgbLinEdge::matches(L.extra[0]) && OutS::matches(L)) GO(k_royale_bloom_approx<SrgbLinEdge, OutS>);
  GO(k_royale_bloom_approx<SRT, StRT>);
}
hipError_t launch_blur9(const PassLaunch& L, hipStream_t s) {
  hipError_t tile_err = hipSuccess;
  if (launch_blur9_tile(L, s, &tile_err)) return tile_err;   // (pass_royale_blur.hip: texels decoded once per workgroup)
  if (SrgbLinEdge::matches(L.in) && OutS::matches(L)) GO(k_blur9<SrgbLinEdge, OutS>);
  GO(k_blur9<SRT, StRT>);
}
hipError_t launch_royale_scan_h(const PassLaunch& L, hipStream_t s) {
  using MaskS = S<FMT_RGBA8, 0, WRAP_EDGE>;
  if (MaskS::matches(L.in) && SrgbLinEdge::matches(L.extra[0]) && OutS::matches(L)) {
    if (!(L.flags & RC_FLAG_GENERAL_ONLY) && rcstrip::separable(L, 0, 2)) {
      static std::mutex mu;
      static std::map<rcstrip::GeoKey, rcstrip::GeoCached<ScanHTables>> cache;
      if (const auto T = rcstrip::geo_tables<ScanHTables>(L, s, mu, cache, buildScanHTables)) {
        const long strips = (long)((L.out_w + 127) / 128) * ((L.out_h + kShRows - 1) / kShRows) * L.n_frames;
        const long blocks = (strips + kSh2Waves - 1) / kSh2Waves;
        hipLaunchKernelGGL(k_royale_scan_h_strip2, dim3((unsigned)(blocks < 768 ? blocks : 768)), dim3(kSh2Waves * 64), rcstrip2::kStrip2LdsUser, s, L, T->cols, T->rows);
        return hipGetLastError();
      }
    }
    GO((k_royale_scan_h<MaskS, SrgbLinEdge, OutS, false>));
  }
  GO((k_royale_scan_h<SRT, SRT, StRT, false>));
}
hipError_t launch_royale_scan_h_fake(const PassLaunch& L, hipStream_t s) {
  using MaskS = S<FMT_RGBA8, 0, WRAP_EDGE>;
  if (MaskS::matches(L.in) && SrgbLinEdge::matches(L.extra[0]) && SrgbLinEdge::matches(L.extra[1]) && SrgbLinEdge::matches(L.extra[2]) &&
      OutS::matches(L))
    GO((k_royale_scan_h<MaskS, SrgbLinEdge, OutS, true>));
  GO((k_royale_scan_h<SRT, SRT, StRT, true>));
}
hipError_t launch_royale_brightpass(const PassLaunch& L, hipStream_t s) {
  if (SrgbNearEdge::matches(L.in) && SrgbLinEdge::matches(L.extra[0]) && OutS::matches(L)) {
    if (!(L.flags & RC_FLAG_GENERAL_ONLY) && rcstrip::separable(L, 0, 2)) {
      static std::mutex mu;
      static std::map<rcstrip::GeoKey, rcstrip::GeoCached<BpTables>> cache;
      if (const auto T = rcstrip::geo_tables<BpTables>(L, s, mu, cache, buildBpTables)) {
        const long strips = (long)((L.out_w + 127) / 128) * ((L.out_h + kBpRows - 1) / kBpRows) * L.n_frames;
        const long blocks = (strips + kBp2Waves - 1) / kBp2Waves;
        hipLaunchKernelGGL(k_royale_brightpass_strip2, dim3((unsigned)(blocks < 768 ? blocks : 768)), dim3(kBp2Waves * 64), rcstrip2::kStrip2LdsUser, s, L, T->cols, T->rows);
        return hipGetLastError();
      }
    }
    GO(k_royale_brightpass<SrgbNearEdge, SrgbLinEdge, OutS>);
  }
  GO(k_royale_brightpass<SRT, SRT, StRT>);
}
hipError_t launch_royale_last(const PassLaunch& L, hipStream_t s) {
  if (lastIsGeneral(L.params)) {
    return launch_royale_last_general(L, s);
  }
  if (L.in.n_levels > 1) {
    if (SrgbLinEdge::matches(L.in) && St<FMT_RGBA8>::matches(L)) GO((k_royale_last<SrgbLinEdge, St<FMT_RGBA8>, true>));
    GO((k_royale_last<SRT, StRT, true>));
  }
  if (SrgbLinEdge::matches(L.in) && St<FMT_RGBA8>::matches(L)) {
    if (!(L.flags & RC_FLAG_GENERAL_ONLY) && rcstrip::separable(L, 0, 1)) {
      static std::mutex mu;
      static std::map<rcstrip::GeoKey, rcstrip::GeoCached<LastTables>> cache;
      if (const auto T = rcstrip::geo_tables<LastTables>(L, s, mu, cache, buildLastTables)) {
        const long strips = (long)((L.out_w + 63) / 64) * ((L.out_h + kLastRows - 1) / kLastRows) * L.n_frames;
        const long blocks = (strips + kLastWaves - 1) / kLastWaves;
        hipLaunchKernelGGL((k_royale_last_strip<St<FMT_RGBA8>>), dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(kLastWaves * 64),
                           kLastLdsList + (unsigned)(kLastWaves * kLastList * 4), s, L, T->cols, T->rows, T->gamma_tab);
        return hipGetLastError();
      }
    }
    GO((k_royale_last<SrgbLinEdge, St<FMT_RGBA8>, false>));
  }
  GO((k_royale_last<SRT, StRT, false>));
}
}  // namespace rck
