// Pass kernels for three reference shader assets (arithmetic spec = the GLSL text):
//   stock.glsl                         shaders/shaders_glsl/stock.glsl
//   scanlines/shaders/scanline.glsl    VS line 50, FS lines 107-113
//   motionblur/shaders/mix_frames.glsl VS lines 51-55, FS lines 94-107 (samples PrevTexture)
//   crt/shaders/crt-pi.glsl            VS lines 96-103, FS lines 131-232
//     (compile-time switches as shipped: SCANLINES, MULTISAMPLE, GAMMA, MASK_TYPE 1)
// One thread per target pixel; blockIdx.z = frame of the batch.
#include "pass_launch.h"

using namespace rcd;

namespace {

// llvmpipe's blit fast path for a pure copy of an RGBA8 texture to a plain RGBA8 target with NEAREST
// + clamp to edge: 16.16 fixed-point stepping of the texture coordinate, re-anchored every 64 target
// pixels (formula measured on the GL, see oracle/rc_passes_basic.c blit_index).
__device__ __forceinline__ int blit_index(float a0, float d, int texsize, int x) {
  const float T = (float)texsize, K = 65536.0f;
  const float fd = d * T;
  const int D = (int)(fd * K);
  const int j = x & ~63;
  const float s0 = fd * (float)j + a0 * T;
  const int S0 = (int)(s0 * K);
  return clampi((S0 + (x - j) * D) >> 16, 0, texsize - 1);
}
__global__ void __launch_bounds__(256) k_stock_blit_nearest(const PassLaunch L) {
  RC_TILE_LOOP_BEGIN
  (void)lo;
  const int sx = blit_index(L.plane[0].a0_lo, L.plane[0].dx_lo, L.in.w, x), sy = blit_index(L.plane[1].a0_lo, L.plane[1].dy_lo, L.in.h, y);
  const uint32_t p = *reinterpret_cast<const uint32_t*>(frame_ptr(L.in, z) + texel_off(L.in.w, sx, sy, 4u));
  *reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z + texel_off(L.out_w, x, y, 4u)) = p;
  RC_TILE_LOOP_END
}

__global__ void __launch_bounds__(256) k_stock(const PassLaunch L) {
  __shared__ SrgbLds lds;
  load_srgb_tables(lds);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  store_rt(L, z, x, y, sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds), &lds);
  RC_TILE_LOOP_END
}

// params: SCANLINE_BASE_BRIGHTNESS, SCANLINE_SINE_COMP_A, SCANLINE_SINE_COMP_B, size
__global__ void __launch_bounds__(256) k_scanline(const PassLaunch L) {
  __shared__ SrgbLds lds;
  load_srgb_tables(lds);
  RC_TILE_LOOP_BEGIN
  const float base = L.params[0], comp_a = L.params[1], comp_b = L.params[2], size = L.params[3];
  const float pi = 3.141592654f;
  const float omega_x = (pi * size) * (float)L.out_w;
  const float omega_y = (2.0f * pi) * (float)L.in.h;
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 res = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float d = comp_a * sin_(u * omega_x) + comp_b * sin_(v * omega_y);
  const float k = base + d;
  store_rt(L, z, x, y, make_float4(res.x * k, res.y * k, res.z * k, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// extra[0] = PrevTexture; mix(colour, colourPrev, 0.5) with a constant weight = a*(1-0.5) + b*0.5
// plane[0], plane[1]: TEX0 = TexCoord * 1.0001
__global__ void __launch_bounds__(256) k_mix_frames(const PassLaunch L) {
  __shared__ SrgbLds lds;
  load_srgb_tables(lds);
  RC_TILE_LOOP_BEGIN
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float4 p = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds);
  store_rt(L, z, x, y, make_float4(c.x * 0.5f + p.x * 0.5f, c.y * 0.5f + p.y * 0.5f, c.z * 0.5f + p.z * 0.5f, 1.0f), &lds);
  RC_TILE_LOOP_END
}

// Conformance fixture tests/fixtures/conformance/feedback-persist.glsl (this repository's own shader):
// max(cur*0.75 + old0*0.25, old1*PERSIST); extra[0] = PassFeedback0, extra[1] = PassFeedback1
__global__ void __launch_bounds__(256) k_feedback_persist(const PassLaunch L) {
  __shared__ SrgbLds lds;
  load_srgb_tables(lds);
  RC_TILE_LOOP_BEGIN
  const float persist = L.params[0];
  const float u = vary(L.plane[0], x, y, lo), v = vary(L.plane[1], x, y, lo);
  const float4 c = sample_rt(L.in, frame_ptr(L.in, z), u, v, &lds);
  const float4 p0 = sample_rt(L.extra[0], frame_ptr(L.extra[0], z), u, v, &lds);
  const float4 p1 = sample_rt(L.extra[1], frame_ptr(L.extra[1], z), u, v, &lds);
  const float m0 = c.x * 0.75f + p0.x * 0.25f, m1 = c.y * 0.75f + p0.y * 0.25f, m2 = c.z * 0.75f + p0.z * 0.25f;
  const float q0 = p1.x * persist, q1 = p1.y * persist, q2 = p1.z * persist;
  store_rt(L, z, x, y, make_float4(m0 < q0 ? q0 : m0, m1 < q1 ? q1 : m1, m2 < q2 ? q2 : m2, 1.0f), &lds);
  RC_TILE_LOOP_END
}

__device__ __forceinline__ float crtpi_weight(float dist, float sw, float gap) {
  float w = 1.0f - (dist * dist) * sw;
  return w > gap ? w : gap;
}

// params: CURVATURE_X, CURVATURE_Y, MASK_BRIGHTNESS, SCANLINE_WEIGHT,
//         SCANLINE_GAP_BRIGHTNESS, BLOOM_FACTOR, INPUT_GAMMA, OUTPUT_GAMMA
// plane[0], plane[1]: TEX0 = TexCoord * 1.0001
template <int IN_FMT, int IN_LINEAR, int IN_WRAP, int OUT_FMT, bool GENERIC>
__global__ void __launch_bounds__(256) k_crt_pi(const PassLaunch L) {
  __shared__ SrgbLds lds;
  if (GENERIC || IN_FMT == FMT_SRGB8 || OUT_FMT == FMT_SRGB8) load_srgb_tables(lds);
  RC_TILE_LOOP_BEGIN
  const float mask_b = L.params[2], sw = L.params[3], gap = L.params[4], bloom = L.params[5];
  const float in_gamma = L.params[6], out_gamma = L.params[7];
  const float tsy = (float)L.in.h;
  const float filter_width = (tsy / (float)L.out_h) / 3.0f;
  const float inv_out_gamma = 1.0f / out_gamma;
  const float tcx = vary(L.plane[0], x, y, lo), tcy = vary(L.plane[1], x, y, lo);
  const float pix_y = tcy * tsy;
  const float temp_y = __builtin_floorf(pix_y) + 0.5f;
  const float y_coord = temp_y / tsy;
  float dy = pix_y - temp_y;
  float slw = crtpi_weight(dy, sw, gap);
  slw += crtpi_weight(dy - filter_width, sw, gap);
  slw += crtpi_weight(dy + filter_width, sw, gap);
  slw *= 0.3333333f;
  const float sign_y = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f);
  dy = dy * dy;
  dy = dy * dy;
  dy *= 8.0f;
  dy /= tsy;
  dy *= sign_y;
  const uint8_t* img = frame_ptr(L.in, z);
  const float4 c = GENERIC ? sample_rt(L.in, img, tcx, y_coord + dy, &lds)
                           : sample<IN_FMT, IN_LINEAR, IN_WRAP>(L.in, img, tcx, y_coord + dy, &lds);
  float r = pow_(c.x, in_gamma), g = pow_(c.y, in_gamma), b = pow_(c.z, in_gamma);
  slw *= bloom;
  r *= slw;
  g *= slw;
  b *= slw;
  r = pow_(r, inv_out_gamma);
  g = pow_(g, inv_out_gamma);
  b = pow_(b, inv_out_gamma);
  const float fx = ((float)x + 0.5f) * 1.0001f * 0.5f;
  const float which = fx - __builtin_floorf(fx);
  float4 o;
  if (which < 0.5f) {
    o = make_float4(r * mask_b, g * 1.0f, b * mask_b, 1.0f);
  } else {
    o = make_float4(r * 1.0f, g * mask_b, b * 1.0f, 1.0f);
  }
  if (GENERIC) store_rt(L, z, x, y, o, &lds);
  else store<OUT_FMT>(L, z, x, y, o, &lds);
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {

hipError_t launch_stock(const PassLaunch& L, hipStream_t s) {
  if (L.in.fmt == FMT_RGBA8 && !L.in.linear && L.in.wrap == WRAP_EDGE && L.out_fmt == FMT_RGBA8) {
    hipLaunchKernelGGL(k_stock_blit_nearest, px_grid(L), px_block(), 0, s, L);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_stock, px_grid(L), px_block(), 0, s, L);
  return hipGetLastError();
}
hipError_t launch_mix_frames(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_mix_frames, px_grid(L), px_block(), 0, s, L);
  return hipGetLastError();
}
hipError_t launch_feedback_persist(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_feedback_persist, px_grid(L), px_block(), 0, s, L);
  return hipGetLastError();
}
hipError_t launch_scanline(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_scanline, px_grid(L), px_block(), 0, s, L);
  return hipGetLastError();
}
hipError_t launch_crt_pi(const PassLaunch& L, hipStream_t s) {
  // the shipped preset's configuration (crt/crt-pi.glslp: linear, clamp_to_border, RGBA8 out,
  // on the RGB source frame) gets a specialised instantiation
  if (L.in.fmt == FMT_RGBX8 && L.in.linear && L.in.wrap == WRAP_BORDER && L.out_fmt == FMT_RGBA8)
    hipLaunchKernelGGL((k_crt_pi<FMT_RGBX8, 1, WRAP_BORDER, FMT_RGBA8, false>), px_grid(L), px_block(), 0, s, L);
  else
    hipLaunchKernelGGL((k_crt_pi<0, 0, 0, 0, true>), px_grid(L), px_block(), 0, s, L);
  return hipGetLastError();
}

}  // namespace rck
