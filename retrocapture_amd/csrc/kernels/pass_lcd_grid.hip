// handheld/shaders/lcd-cgwg/lcd-grid-v2.glsl (handheld/lcd-grid-v2.glslp and the lcd-grid-v2-<colour>[-motionblur] chains): every target
// pixel integrates the LCD subpixel profile over the four source texels around it - intsmear (FS 150-166), a 13th-degree odd
// polynomial per edge, six times horizontally (three subpixels of the left and of the right texel) and twice vertically -
// on texels fetched with texelFetchOffset and passed through pow(gain * t + blacklevel, gamma) + ambient; then the 3x3
// subpixel colour matrix (pow(*SUBPIX_*, outgamma)) and the output gamma.
//
// ~580 scalar operations that the GL's compiler reassociates, so the per-pixel body is the GL's own instruction order:
// gen/lcd_grid_v2_fs.inc comes from the NIR listing of Mesa llvmpipe (oracle/glrun/nir2c.py, recipe gen_lists.sh), one
// statement per instruction, over rc_device.h's llvmpipe-exact primitives.  One thread per target pixel; the four texels of a
// pixel are shared with its neighbours and stay in L1 / L2 (the source is far smaller than the target in every preset).
// Bound: VALU (24 pow and ~560 FP32 operations per pixel against 4 B written).
#include "pass_launch.h"
#include "rc_vecmath.h"

using namespace rcd;

namespace {

struct TexCtx {
  const Tex* t;
  const uint8_t* img;
  const SrgbLds* lds;
};

// texelFetch: the decoded texel, zeros outside the image
__device__ __forceinline__ void rcn_txf(void* vctx, int x, int y, float* dst) {
  const TexCtx* c = static_cast<const TexCtx*>(vctx);
  const Tex& t = *c->t;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x >= 0 && y >= 0 && x < t.w && y < t.h) {
    switch (t.fmt) {
      case FMT_SRGB8: r = texel<FMT_SRGB8>(t, c->img, x, y, c->lds); break;
      case FMT_RGBX8: r = texel<FMT_RGBX8>(t, c->img, x, y, c->lds); break;
      case FMT_F32: r = texel<FMT_F32>(t, c->img, x, y, c->lds); break;
      case FMT_F16: r = texel<FMT_F16>(t, c->img, x, y, c->lds); break;
      default: r = texel<FMT_RGBA8>(t, c->img, x, y, c->lds); break;
    }
  }
  dst[0] = r.x;
  dst[1] = r.y;
  dst[2] = r.z;
  dst[3] = r.w;
}
__device__ __forceinline__ void rcn_tex(void* vctx, float u, float v, float* dst) {
  const TexCtx* c = static_cast<const TexCtx*>(vctx);
  const float4 r = sample_rt(*c->t, c->img, u, v, c->lds);
  dst[0] = r.x;
  dst[1] = r.y;
  dst[2] = r.z;
  dst[3] = r.w;
}
// cvttps2dq: the "integer indefinite" value for NaN and out-of-range inputs
__device__ __forceinline__ int rcn_f2i(float x) { return (x != x || x >= 2147483648.0f || x < -2147483648.0f) ? (-2147483647 - 1) : (int)x; }
__device__ __forceinline__ float rcn_min(float a, float b) { return b != b ? a : (a < b ? a : b); }   // gallivm's fmin / fmax: the operand that is not NaN
__device__ __forceinline__ float rcn_max(float a, float b) { return b != b ? a : (a > b ? a : b); }
__device__ __forceinline__ float rcn_pow(float x, float y) { return x != x ? 0.0f : pow_(x, y); }     // llvmpipe: pow of a NaN base is 0

#define RCN_FN __device__ __forceinline__ static
#define RCN_NO_TABLES
#define RCN_BITS(u) bits2f(u)
#define RCN_RCP(x) (1.0f / (x))
#define RCN_DIV(a, b) ((a) / (b))
#define RCN_FLOOR(x) __builtin_floorf(x)
#define RCN_F2I(x) rcn_f2i(x)
#define RCN_MIN(a, b) rcn_min(a, b)
#define RCN_MAX(a, b) rcn_max(a, b)
#define RCN_POW(a, b) rcn_pow(a, b)
#define RCN_TXF(ctx, unit, x, y, dst) rcn_txf(ctx, x, y, dst)
#define RCN_TEX(ctx, unit, u, v, dst) rcn_tex(ctx, u, v, dst)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#pragma clang diagnostic ignored "-Wunused-variable"
#include "gen/lcd_grid_v2_fs.inc"
#include "gen/lcd_grid_fs.inc"
#pragma clang diagnostic pop

// params[0..14]: the shader's 15 #pragma parameters in declaration order = dwords 6..20 of its uniform block
__global__ void __launch_bounds__(256) k_lcd_grid_v2(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  float U[21];
  U[0] = (float)L.out_w;
  U[1] = (float)L.out_h;
  U[2] = (float)L.in.w;
  U[3] = (float)L.in.h;
  U[4] = U[2];
  U[5] = U[3];
#pragma unroll
  for (int k = 0; k < 15; ++k) U[6 + k] = L.params[k];
  RC_TILE_LOOP_BEGIN
  const float in[2] = {vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo)};
  float out[4];
  TexCtx ctx{&L.in, frame_ptr(L.in, z), &lds};
  lcd_grid_v2_fs(U, in, out, &ctx);
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], out[3]), &lds);
  RC_TILE_LOOP_END
}

// handheld/shaders/lcd-cgwg/lcd-grid.glsl (handheld/lcd-grid.glslp, nds.glslp, twenty console-border presets): the first version of the
// same idea - four texture() samples under the subpixel integrals, one GRID_STRENGTH, an input gamma.  ~560 operations in the GL's
// own order (gen/lcd_grid_fs.inc).  params[0..1]: GRID_STRENGTH, gamma = dwords 6, 7 of the uniform block.
__global__ void __launch_bounds__(256) k_lcd_grid(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  const float U[8] = {(float)L.out_w, (float)L.out_h, (float)L.in.w, (float)L.in.h, (float)L.in.w, (float)L.in.h, L.params[0], L.params[1]};
  RC_TILE_LOOP_BEGIN
  const float in[2] = {vary(L.plane[0], x, y, lo), vary(L.plane[1], x, y, lo)};
  float out[4];
  TexCtx ctx{&L.in, frame_ptr(L.in, z), &lds};
  lcd_grid_fs(U, in, out, &ctx);
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], out[3]), &lds);
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {
hipError_t launch_lcd_grid(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_lcd_grid, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_lcd_grid_v2(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL(k_lcd_grid_v2, px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
}  // namespace rck
