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
// sampler policies: the run-time selected sampler, and compile-time ones for what the shipped presets bind (a NEAREST or LINEAR clamp-to-edge
// RGBX8 source frame / RGBA8 target) - a list with 31 taps carries 31 copies of the sampler, so its size matters
struct SampRT {
  static __device__ __forceinline__ float4 get(const Tex& t, const uint8_t* img, float u, float v, const SrgbLds* l) { return sample_rt(t, img, u, v, l); }
  static bool matches(const Tex&) { return true; }
};
template <int FMT, int LIN>
struct SampEdge {
  static __device__ __forceinline__ float4 get(const Tex& t, const uint8_t* img, float u, float v, const SrgbLds* l) { return sample<FMT, LIN, WRAP_EDGE>(t, img, u, v, l); }
  static bool matches(const Tex& t) { return t.fmt == FMT && (t.linear != 0) == (LIN != 0) && t.wrap == WRAP_EDGE && t.n_levels <= 1; }
};
template <class SI>
__device__ __forceinline__ void rcn_tex(void* vctx, float u, float v, float* dst) {
  const TexCtx* c = static_cast<const TexCtx*>(vctx);
  const float4 r = SI::get(*c->t, c->img, u, v, c->lds);
  dst[0] = r.x;
  dst[1] = r.y;
  dst[2] = r.z;
  dst[3] = r.w;
}
__device__ __forceinline__ float rcn_min(float a, float b) { return b != b ? a : (a < b ? a : b); }   // gallivm's fmin / fmax: the operand that is not NaN
__device__ __forceinline__ float rcn_max(float a, float b) { return b != b ? a : (a > b ? a : b); }
__device__ __forceinline__ float rcn_pow(float x, float y) { return x != x ? 0.0f : pow_(x, y); }     // llvmpipe: pow of a NaN base is 0

#define RCN_FN static __device__ __forceinline__
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
#define RCN_TEX(ctx, unit, u, v, dst) rcn_tex<SI>(ctx, u, v, dst)
// the lists as static members of a template over the sampler policy
template <class SI>
struct Lists {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#pragma clang diagnostic ignored "-Wunused-variable"
#include "gen/tvout_tweaks_fs.inc"
#include "gen/jinc2_sharper_fs.inc"
#include "gen/crt_lottes_fs.inc"
#include "gen/fakelottes_fs.inc"
#include "gen/side_by_side_fs.inc"
#include "gen/sameboy_lcd_fs.inc"
#include "gen/crt_consumer_fs.inc"
#include "gen/reverse_aa_fs.inc"
#include "gen/advanced_aa_fs.inc"
#include "gen/image_adjustment_fs.inc"
#pragma clang diagnostic pop
};
enum { LIST_TVOUT, LIST_JINC2, LIST_LOTTES, LIST_FAKELOTTES, LIST_SBS, LIST_SAMEBOY_LCD, LIST_CONSUMER, LIST_REVERSE_AA, LIST_ADVANCED_AA, LIST_IMAGE_ADJ };
template <class SI, int WHICH>
__device__ __forceinline__ void run_list(const float* U, const float* in, float* out, void* ctx) {
  if (WHICH == LIST_TVOUT) Lists<SI>::tvout_tweaks_fs(U, in, out, ctx);
  else if (WHICH == LIST_JINC2) Lists<SI>::jinc2_sharper_fs(U, in, out, ctx);
  else if (WHICH == LIST_LOTTES) Lists<SI>::crt_lottes_fs(U, in, out, ctx);
  else if (WHICH == LIST_FAKELOTTES) Lists<SI>::fakelottes_fs(U, in, out, ctx);
  else if (WHICH == LIST_SBS) Lists<SI>::side_by_side_fs(U, in, out, ctx);
  else if (WHICH == LIST_SAMEBOY_LCD) Lists<SI>::sameboy_lcd_fs(U, in, out, ctx);
  else if (WHICH == LIST_CONSUMER) Lists<SI>::crt_consumer_fs(U, in, out, ctx);
  else if (WHICH == LIST_REVERSE_AA) Lists<SI>::reverse_aa_fs(U, in, out, ctx);
  else if (WHICH == LIST_ADVANCED_AA) Lists<SI>::advanced_aa_fs(U, in, out, ctx);
  else Lists<SI>::image_adjustment_fs(U, in, out, ctx);
}

// The uniform block of the list is handed over ready-made in params[kListU0 ..] (registry: setupTvoutTweaks / setupImageAdjustment).
// FC: the block's FrameCount slot (an int uniform the lists take as a float holding its value), or -1
template <int NU, int FC, int WHICH, class SI>
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
  if (WHICH == LIST_ADVANCED_AA) {   // six varyings (the neighbour coordinates its vertex stage prepares): planes 2..5 as well
#pragma unroll
    for (int k = 2; k < 6; ++k) in[k] = vary(L.plane[k], x, y, lo);
  }
  in[32] = (float)x + 0.5f;
  in[33] = (float)y + 0.5f;
  in[34] = 0.5f;
  in[35] = 1.0f;
  float out[4] = {0.f, 0.f, 0.f, 0.f};   // a component the shader never writes (jinc2-sharper's alpha) is stored as 0 by the GL
  TexCtx ctx{&L.in, frame_ptr(L.in, z), &lds};
  run_list<SI, WHICH>(U, in, out, &ctx);
  store_rt(L, z, x, y, make_float4(out[0], out[1], out[2], out[3]), &lds);
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {
template <int NU, int FC, int WHICH>
static hipError_t launch_list(const PassLaunch& L, hipStream_t s) {
#define RC_LIST_GO(SI)                                                                                                   \
  do {                                                                                                                   \
    hipLaunchKernelGGL((k_list_pass<NU, FC, WHICH, SI>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L);          \
    return hipGetLastError();                                                                                            \
  } while (0)
  using XN = SampEdge<FMT_RGBX8, 0>;
  using XL = SampEdge<FMT_RGBX8, 1>;
  using AN = SampEdge<FMT_RGBA8, 0>;
  using AL = SampEdge<FMT_RGBA8, 1>;
  if (!(L.flags & RC_FLAG_GENERAL_ONLY)) {
    if (XN::matches(L.in)) RC_LIST_GO(XN);
    if (XL::matches(L.in)) RC_LIST_GO(XL);
    if (AN::matches(L.in)) RC_LIST_GO(AN);
    if (AL::matches(L.in)) RC_LIST_GO(AL);
  }
  RC_LIST_GO(SampRT);
#undef RC_LIST_GO
}
// crt/shaders/tvout-tweaks.glsl, windowed/shaders/jinc2-sharper.glsl (10 presets: a 4x4 jinc-windowed-jinc resampler, 16 taps, two sin of a sqrt
// distance each, anti-ringing clamp), crt/shaders/crt-lottes.glsl (~2 500 operations: 31 taps under gaussian pixel / scanline / bloom kernels, tube
// warp, four shadow masks on gl_FragCoord, 48 branches), its one-tap cousin crt/shaders/fakelottes.glsl, misc/image-adjustment.glsl
hipError_t launch_tvout_tweaks(const PassLaunch& L, hipStream_t s) { return launch_list<kTvoutU, -1, LIST_TVOUT>(L, s); }
hipError_t launch_jinc2_sharper(const PassLaunch& L, hipStream_t s) { return launch_list<kJinc2U, -1, LIST_JINC2>(L, s); }
hipError_t launch_crt_lottes(const PassLaunch& L, hipStream_t s) { return launch_list<kLottesU, -1, LIST_LOTTES>(L, s); }
hipError_t launch_fakelottes(const PassLaunch& L, hipStream_t s) { return launch_list<kFakeLottesU, -1, LIST_FAKELOTTES>(L, s); }
// stereoscopic-3d/shaders/side-by-side-simple.glsl: one copy of the frame per eye, sampled only inside the frame (taps inside branches), added
hipError_t launch_side_by_side(const PassLaunch& L, hipStream_t s) { return launch_list<kSbsU, -1, LIST_SBS>(L, s); }
// handheld/shaders/sameboy-lcd.glsl: SameBoy's LCD filter, nine taps chosen by the sub-pixel position, a scanline term
hipError_t launch_sameboy_lcd(const PassLaunch& L, hipStream_t s) { return launch_list<kSameboyLcdU, -1, LIST_SAMEBOY_LCD>(L, s); }
// crt/shaders/crt-consumer.glsl: ~970 operations, 24 taps, 19 branches; FrameCount (noise) at dword 0 of its block
hipError_t launch_crt_consumer(const PassLaunch& L, hipStream_t s) { return launch_list<kConsumerU, 0, LIST_CONSUMER>(L, s); }
// anti-aliasing/shaders/reverse-aa.glsl: 3x3 neighbourhood, clamped tilt estimates, two sub-pixel corrections
hipError_t launch_reverse_aa(const PassLaunch& L, hipStream_t s) { return launch_list<kReverseAaU, -1, LIST_REVERSE_AA>(L, s); }
// anti-aliasing/shaders/advanced-aa.glsl: nine taps at vertex-stage coordinates, edge-directed blend; no uniform block
hipError_t launch_advanced_aa(const PassLaunch& L, hipStream_t s) { return launch_list<1, -1, LIST_ADVANCED_AA>(L, s); }
hipError_t launch_image_adjustment(const PassLaunch& L, hipStream_t s) { return launch_list<kImageAdjU, kImageAdjFrameCount, LIST_IMAGE_ADJ>(L, s); }
}  // namespace rck
