// Two hub shaders of the reference's preset tree whose per-pixel bodies are the GL's own instruction lists (gen/*.inc, produced by
// oracle/glrun/nir2c.py from Mesa llvmpipe's NIR; recipe oracle/glrun/gen_lists.sh; pass_lcd_grid.hip and pass_royale_last_general.hip
// explain why):
//   crt/shaders/tvout-tweaks.glsl (39 presets), FS 99-214: sinc resampling of Y, I and Q at three signal bandwidths - 32 sin per pixel -
//       composite cross-talk and TV colour levels behind run-time switches; ~840 operations, 17 (wave-uniform) branches.
//   misc/image-adjustment.glsl (48 presets): gamma, saturation / contrast / luminance, channel gains, overscan masks, film grain seeded
//       by FrameCount, sharpen; its vertex stage (zoom, shift, overscan) runs on the host per launch (list_setup.cpp); the two flip
//       parameters move the quad itself half off the target and are refused (kernel_registry.cpp).
// One thread per target pixel; uniform blocks are built per launch in the layouts the lists address.  Both are VALU-bound.
#include "list_params.h"
#include "pass_launch.h"
#include "rc_vecmath.h"

using namespace rcd;

namespace {

struct TexCtx {
  const Tex* t;
  const uint8_t* img;
  const SrgbLds* lds;
};
__device__ __forceinline__ void rcn_tex(void* vctx, float u, float v, float* dst) {
  const TexCtx* c = static_cast<const TexCtx*>(vctx);
  const float4 r = sample_rt(*c->t, c->img, u, v, c->lds);
  dst[0] = r.x;
  dst[1] = r.y;
  dst[2] = r.z;
  dst[3] = r.w;
}
__device__ __forceinline__ float rcn_min(float a, float b) { return b != b ? a : (a < b ? a : b); }   // gallivm's fmin / fmax: the operand that is not NaN
__device__ __forceinline__ float rcn_max(float a, float b) { return b != b ? a : (a > b ? a : b); }
__device__ __forceinline__ float rcn_pow(float x, float y) { return x != x ? 0.0f : pow_(x, y); }     // llvmpipe: pow of a NaN base is 0

#define RCN_FN __device__ __forceinline__ static
#define RCN_NO_TABLES
#define RCN_BITS(u) bits2f(u)
#define RCN_ABS(x) __builtin_fabsf(x)
#define RCN_RCP(x) (1.0f / (x))
#define RCN_DIV(a, b) ((a) / (b))
#define RCN_FLOOR(x) __builtin_floorf(x)
#define RCN_FRACT(x) ((x) - __builtin_floorf(x))
#define RCN_MIN(a, b) rcn_min(a, b)
#define RCN_MAX(a, b) rcn_max(a, b)
#define RCN_POW(a, b) rcn_pow(a, b)
#define RCN_SIN(x) sin_(x)
#define RCN_SQRT(x) __builtin_sqrtf(x)
#define RCN_EXP2(x) exp2_(x)
#define RCN_TEX(ctx, unit, u, v, dst) rcn_tex(ctx, u, v, dst)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#pragma clang diagnostic ignored "-Wunused-variable"
#include "gen/tvout_tweaks_fs.inc"
#include "gen/jinc2_sharper_fs.inc"
#include "gen/crt_lottes_fs.inc"
#include "gen/fakelottes_fs.inc"
#include "gen/image_adjustment_fs.inc"
#pragma clang diagnostic pop

// The uniform block of the list is handed over ready-made in params[kListU0 ..] (registry: setupTvoutTweaks / setupImageAdjustment).
// FC: the block's FrameCount slot (an int uniform the lists take as a float holding its value), or -1
template <int NU, int FC, void (*FS)(const float*, const float*, float*, void*)>
__global__ void __launch_bounds__(256) k_list_pass(const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  float U[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) U[k] = L.params[kListU0 + k];
  RC_TILE_LOOP_BEGIN
  if (FC >= 0) U[FC] = (float)(L.frame_count0 + z);
  // varying slots 0, 1 = TEX0; 32..35 = gl_FragCoord (pixel + 0.5), read by the lists that build a mask on screen position
  float in[36];
  in[0] = vary(L.plane[0], x, y, lo);
  in[1] = vary(L.plane[1], x, y, lo);
  in[32] = (float)x + 0.5f;
  in[33] = (float)y + 0.5f;
  in[34] = 0.5f;
  in[35] = 1.0f;
  float out[4] = {0.f, 0.f, 0.f, 0.f};   // a component the shader never writes (jinc2-sharper's alpha) is stored as 0 by the GL
  TexCtx ctx{&L.in, frame_ptr(L.in, z), &lds};
  FS(U, in, out, &ctx);
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], out[3]), &lds);
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {
hipError_t launch_tvout_tweaks(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL((k_list_pass<kTvoutU, -1, tvout_tweaks_fs>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
// windowed/shaders/jinc2-sharper.glsl (10 presets): a 4x4 jinc-windowed-jinc resampler - 16 taps, two sin of a sqrt distance each - with an
// anti-ringing clamp; ~430 operations.  Uniform block: TextureSize only.
hipError_t launch_jinc2_sharper(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL((k_list_pass<kJinc2U, -1, jinc2_sharper_fs>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
// crt/shaders/crt-lottes.glsl (~2 500 operations: 31 taps under gaussian pixel / scanline / bloom kernels, tube warp, four shadow masks on
// gl_FragCoord, 48 branches) and its one-tap cousin crt/shaders/fakelottes.glsl
hipError_t launch_crt_lottes(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL((k_list_pass<kLottesU, -1, crt_lottes_fs>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_fakelottes(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL((k_list_pass<kFakeLottesU, -1, fakelottes_fs>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
hipError_t launch_image_adjustment(const PassLaunch& L, hipStream_t s) {
  hipLaunchKernelGGL((k_list_pass<kImageAdjU, kImageAdjFrameCount, image_adjustment_fs>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);
  return hipGetLastError();
}
}  // namespace rck
