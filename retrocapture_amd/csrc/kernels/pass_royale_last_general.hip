// crt-royale pass 11, general form: geometry-aa-last-pass.glsl with curved geometry (geom_mode_runtime 1..3: sphere,
// alt. sphere, cylinder) or overscan != 1 - FS 5451-5531 takes the tex2Daa12x branch (3985-4063: twelve LINEAR taps on
// a quincunx-like grid, cubic weights of the run-time aa_cubic_c, red and blue offset by a third of a pixel) and, when
// curved, casts a ray per pixel and builds the pixel-to-uv tangent matrix (2527-3010).
//
// The default parameters take the flat form (pass_royale.hip: k_royale_last / k_royale_last_strip); this one exists so
// that every value of the 44 parameters runs, and runs with the GL's results.  The GL's compiler rearranges these
// ~1200 scalar operations freely (the sample grid is folded to constants, weight sums are factored, sub-expressions are
// shared across the geometry branches), so the per-pixel body is the GL's own final instruction order:
// gen/royale_last_fs.inc is produced by oracle/glrun/nir2c.py from the NIR listing of Mesa llvmpipe (recipe:
// oracle/glrun/gen_lists.sh) - one statement per instruction - and the float built-ins map to rc_device.h's
// llvmpipe-exact primitives.  Around it, as for every pass: the vertex stage is evaluated once per launch on the host
// (royale_setup.cpp, setupLast), the two varyings that change across the quad are plane equations, one thread per pixel.
//
// Roofline: the pass reads a 13-tap neighbourhood that stays in L2 and writes 4 B per pixel; it is VALU-bound by the
// ray cast and the twelve-tap filter.
#include "royale_common.h"

using namespace rcd;
using namespace rcroyale;

namespace {

// texture(): plain, or - mipmap_input, crt-royale-fake-bloom - with the LOD the GL derives per tex instruction from ONE
// set of coordinate differences per 2x2 pixel quad, taken at the quad's top-left pixel (rc_device.h, mip-mapped sampling).
// The body is then evaluated at the quad's top-left, top-right and bottom-left pixels first - as this pixel's triangle
// extrapolates them - recording the taps' coordinates (the arithmetic that only feeds colours drops out of those three
// evaluations), and the fourth evaluation filters tap k at lod[k].
enum { TEX_PLAIN = 0, TEX_REC_TL, TEX_REC_TR, TEX_REC_BL, TEX_MIP };
constexpr int kMaxTaps = 13;
struct TexCtx {
  const Tex* t;
  const uint8_t* img;
  const SrgbLds* lds;
  int z, n;
  int n_pow;         // pow calls of the body so far (compile-time once the body is inlined)
  bool defer_gamma;  // the three output-gamma pows - the body's last three - are left to the caller: the body returns their bases
  float tl[kMaxTaps][2], tr[kMaxTaps][2], lod[kMaxTaps];
};

#define RCN_FN __device__ __forceinline__ static
#define RCN_BITS(u) bits2f(u)
#define RCN_ABS(x) __builtin_fabsf(x)
#define RCN_RSQ(x) (1.0f / __builtin_sqrtf(x))
#define RCN_RCP(x) (1.0f / (x))
#define RCN_SQRT(x) __builtin_sqrtf(x)
#define RCN_SIGN(x) rcn_sign(x)
#define RCN_SIN(x) sin_(x)
#define RCN_COS(x) cos_(x)
#define RCN_DIV(a, b) ((a) / (b))
#define RCN_MIN(a, b) rcn_min(a, b)
#define RCN_MAX(a, b) rcn_max(a, b)
#define RCN_POW(a, b) rcn_pow_ctx(TEXCTX, a, b)

__device__ __forceinline__ float rcn_sign(float x) { return x == 0.0f ? 0.0f : __builtin_copysignf(1.0f, x); }
// fmin / fmax as gallivm builds them (MINPS / MAXPS, then the first operand where the second is NaN)
__device__ __forceinline__ float rcn_min(float a, float b) { return b != b ? a : (a < b ? a : b); }
__device__ __forceinline__ float rcn_max(float a, float b) { return b != b ? a : (a > b ? a : b); }
// llvmpipe's pow selects 0 where "x == 0" under an unordered compare: a NaN base gives 0
__device__ __forceinline__ float rcn_pow(float x, float y) { return x != x ? 0.0f : pow_(x, y); }

// The body calls pow four times: the border's (get_border_dim_factor) first, then OUT.rgb = pow(colour, 1 / lcd_gamma).  With
// defer_gamma those three return their base unchanged and the kernel encodes them from the gamma table (royale_common.h).
__device__ __forceinline__ float rcn_pow_ctx(void* ctx, float x, float y) {
  TexCtx* c = static_cast<TexCtx*>(ctx);
  const int k = c->n_pow++;
  return (c->defer_gamma && k >= 1) ? x : rcn_pow(x, y);
}

template <class SI, int MODE>
__device__ __forceinline__ void rcn_tex(void* ctx, float u, float v, float* dst) {
  TexCtx* c = static_cast<TexCtx*>(ctx);
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == TEX_PLAIN) {
    t = SI::get(*c->t, c->img, u, v, c->lds);
  } else {
    const int k = c->n++;   // a compile-time constant once the body is inlined: the taps sit in straight-line code
    if (MODE == TEX_REC_TL) {
      c->tl[k][0] = u;
      c->tl[k][1] = v;
    } else if (MODE == TEX_REC_TR) {
      c->tr[k][0] = u;
      c->tr[k][1] = v;
    } else if (MODE == TEX_REC_BL) {
      c->lod[k] = lod_from_quad(*c->t, c->tl[k][0], c->tr[k][0], c->tl[k][1], c->tr[k][1], c->tl[k][0], u, c->tl[k][1], v);
    } else {
      t = c->lod[k] > 0.0f ? sample_mip(*c->t, c->z, u, v, c->lod[k], c->lds) : SI::get(*c->t, c->img, u, v, c->lds);
    }
  }
  dst[0] = t.x;
  dst[1] = t.y;
  dst[2] = t.z;
  dst[3] = t.w;
}

// the generated body samples through RCN_TEX; the sampler policy and the tap mode are the enclosing template's
template <class SI, int MODE>
struct LastFs {
#define RCN_TEX(ctx, unit, u, v, dst) rcn_tex<SI, MODE>(ctx, u, v, dst)
#define RCN_NO_TABLES
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-but-set-variable"
#pragma clang diagnostic ignored "-Wunused-variable"
#include "gen/royale_last_fs.inc"
#pragma clang diagnostic pop
#undef RCN_NO_TABLES
#undef RCN_TEX
};

// GTAB: plain RGBA8 target with the gamma table in LDS behind the decode table - a pixel's bytes come from the table where they are
// certain (colour in [0, 1], royale_common.h last_gamma_byte), from the exact pow otherwise (the body's own operations).
template <class SI, class SO, bool MIP, bool GTAB>
__global__ void __launch_bounds__(256) k_royale_last_general(const PassLaunch L, const float4* __restrict__ gamma_tab) {
  RC_SRGB_LDS(lds, L);
  if (GTAB) {
    if ((uint32_t)(uintptr_t)(RC_AS3 uint32_t*)rc_dyn_lds_ != 0u) __builtin_trap();   // the table is addressed by absolute LDS offsets
    for (int i = (int)(threadIdx.y * blockDim.x + threadIdx.x); i < kLastTabNodes; i += (int)(blockDim.x * blockDim.y))
      reinterpret_cast<float4*>(rc_dyn_lds_ + kLastLdsTab / 4)[i] = gamma_tab[i];
    __syncthreads();
  }
  const float* P = L.params;
  // the fragment stage's uniform block, in the layout the listing addresses (gen/royale_last_fs.inc, royale_last_fs_uniforms)
  const float U[12] = {P[1], P[28], P[30], P[31], P[32], P[39], P[40], P[41], (float)L.in.w, (float)L.in.h, (float)L.in.w, (float)L.in.h};
  RC_TILE_LOOP_BEGIN
  float in[kLastVaryings], out[4];
#pragma unroll
  for (int k = 2; k < kLastVaryings; ++k) in[k] = P[RP11_VARYING0 + k];   // the same at all four vertices: constant planes
  TexCtx ctx;
  ctx.t = &L.in;
  ctx.img = frame_ptr(L.in, z);
  ctx.lds = &lds;
  ctx.z = z;
  ctx.defer_gamma = false;
  if (MIP) {
    const int x0 = x & ~1, y0 = y & ~1;
    in[0] = vary(L.plane[0], x0, y0, lo);
    in[1] = vary(L.plane[1], x0, y0, lo);
    ctx.n = 0;
    ctx.n_pow = 0;
    LastFs<SI, TEX_REC_TL>::royale_last_fs(U, in, out, &ctx);
    in[0] = vary(L.plane[0], x0 + 1, y0, lo);
    in[1] = vary(L.plane[1], x0 + 1, y0, lo);
    ctx.n = 0;
    ctx.n_pow = 0;
    LastFs<SI, TEX_REC_TR>::royale_last_fs(U, in, out, &ctx);
    in[0] = vary(L.plane[0], x0, y0 + 1, lo);
    in[1] = vary(L.plane[1], x0, y0 + 1, lo);
    ctx.n = 0;
    ctx.n_pow = 0;
    LastFs<SI, TEX_REC_BL>::royale_last_fs(U, in, out, &ctx);
  }
  in[0] = vary(L.plane[0], x, y, lo);
  in[1] = vary(L.plane[1], x, y, lo);
  ctx.n = 0;
  ctx.n_pow = 0;
  ctx.defer_gamma = GTAB;
  LastFs<SI, MIP ? TEX_MIP : TEX_PLAIN>::royale_last_fs(U, in, out, &ctx);
  if (GTAB) {
    // out[0..2] are the bases of the output gamma
    bool fail = !(out[0] >= 0.0f && out[0] <= 1.0f && out[1] >= 0.0f && out[1] <= 1.0f && out[2] >= 0.0f && out[2] <= 1.0f);
    const float c0 = fail ? 0.5f : out[0], c1 = fail ? 0.5f : out[1], c2 = fail ? 0.5f : out[2];   // (keeps the lookups inside the table)
    uint32_t px = last_gamma_byte(c0, &fail) | (last_gamma_byte(c1, &fail) << 8) | (last_gamma_byte(c2, &fail) << 16) | (unorm8(out[3]) << 24);
    if (fail) {
      const float ig = 1.0f / U[0];   // the body's RCN_RCP(lcd_gamma)
      px = unorm8(rcn_pow(out[0], ig)) | (unorm8(rcn_pow(out[1], ig)) << 8) | (unorm8(rcn_pow(out[2], ig)) << 16) | (unorm8(out[3]) << 24);
    }
    reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z)[(size_t)y * L.out_w + x] = px;
  } else {
    SO::put(L, z, x, y, make_float4(out[0], out[1], out[2], out[3]), &lds);
  }
  RC_TILE_LOOP_END
}

}  // namespace

namespace rck {
hipError_t launch_royale_last_general(const PassLaunch& L, hipStream_t s) {
  if (L.in.n_levels > 1) {   // mipmap_input: four evaluations per pixel, run-time sampler only
    hipLaunchKernelGGL((k_royale_last_general<SRT, StRT, true, false>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L, nullptr);
    return hipGetLastError();
  }
  if (SrgbLinEdge::matches(L.in) && St<FMT_RGBA8>::matches(L)) {
    const float4* tab = (L.flags & RC_FLAG_GENERAL_ONLY) ? nullptr : royale_last_gamma_table(1.0f / L.params[1], s);
    if (tab)
      hipLaunchKernelGGL((k_royale_last_general<SrgbLinEdge, St<FMT_RGBA8>, false, true>), px_grid(L), px_block(),
                         rcd::srgb_lds_bytes(L) + kLastTabNodes * sizeof(float4), s, L, tab);
    else
      hipLaunchKernelGGL((k_royale_last_general<SrgbLinEdge, St<FMT_RGBA8>, false, false>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L, nullptr);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((k_royale_last_general<SRT, StRT, false, false>), px_grid(L), px_block(), rcd::srgb_lds_bytes(L), s, L, nullptr);
  return hipGetLastError();
}
}  // namespace rck
