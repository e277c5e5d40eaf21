// Device-side building blocks shared by every pass kernel (gfx950 / CDNA4 only).
//
// A pass kernel computes, for each pixel of its render target, exactly what the GL
// fragment pipeline computes for the reference's draw call (reference
// src/shader/ShaderEngine.cpp:850-1475): varyings from the quad's plane equations, texture
// taps with the sampler state the preset sets, the shader's arithmetic, and the
// target-format store.  The float behaviour that defines "the same result" is that of the GL
// the reference is measured on (Mesa llvmpipe 23.2.1): its pow/exp2/log2/sin/cos
// polynomials (fused multiply-adds where llvmpipe fuses), its texel decode and two filter
// paths, its varying setup and its store rounding.  All of that is written here directly in
// HIP; the file is compiled with -ffp-contract=off so only the explicit __builtin_fmaf calls
// fuse, and float division / sqrt are IEEE-correct (hipcc default).  Denormals are flushed
// (-fgpu-flush-denormals-to-zero), like the GL's FTZ/DAZ execution mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rcd {

// ---------------------------------------------------------------- launch descriptors ----
// FMT_F16: opt-in storage of float_framebuffer targets as four binary16 values (engine setFloatTargetFp16): the pass
// still computes in float, a store rounds to nearest even, a fetch widens exactly.  Not the reference's format (RGBA32F).
enum Fmt : int { FMT_RGBA8 = 0, FMT_SRGB8 = 1, FMT_RGBX8 = 2, FMT_F32 = 3, FMT_F16 = 4 };
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
enum Wrap : int { WRAP_EDGE = 0, WRAP_BORDER = 1, WRAP_REPEAT = 2, WRAP_MIRROR = 3 };

struct Tex {
  const void* base;      // frame 0 of the batch
  uint64_t frame_stride; // bytes between consecutive frames (0: one image shared by all)
  int w, h;
  int fmt, linear, wrap;
  // mipmap_input (ShaderEngine.cpp:1022-1033): levels 1..n_levels-1 of the chain, packed one after the
  // other per frame (level k is max(1, w >> k) x max(1, h >> k)); n_levels <= 1: not mip-mapped
  int n_levels = 0;
  const void* mip_base = nullptr;
  uint64_t mip_frame_stride = 0;
  // A folded pass (shader_engine.cpp runChunk): the texture is an sRGB8 render target that was never written - its bytes
  // would be a per-channel byte map of the bytes at `base` (crt-royale's pass 0 at 1:1), so a consumer reads `base` and
  // takes `dec[byte]` (256 floats, device memory: the target's decode of the mapped byte) for the target's decoded texel,
  // alpha 1; the mapped byte itself - the byte the target would hold - follows as 256 words (`dec + 256`, kFoldedTableWords in
  // all).  Only kernels registered with KernelEntry::decode_table_inputs are handed such a texture.
  const float* dec = nullptr;
  int alpha_one = 0;
};

// Plane equation of one varying for the two triangles of the quad (see host varying.cpp).
struct Plane {
  float a0_lo, dx_lo, dy_lo, a0_up, dx_up, dy_up;
};

// PassLaunch::flags bits (bit 0 is crt-royale's RC_FLAG_UNDEF_VARYING_ZERO, kernels/royale_params.h)
constexpr int RC_FLAG_GENERAL_ONLY = 1 << 16; // launchers must pick the general kernel form
constexpr int RC_FLAG_ASYNC_TABLES = 1 << 18;  // per-geometry tables that take long to build (royale_strip.h geo_tables) are built off the frame path: the general form serves meanwhile
constexpr int RC_FLAG_STOCK_NO_BLIT = 1 << 17; // stock.glsl: the ordinary sampler even where llvmpipe's blit fast path would apply (mip generation)
constexpr int RC_FLAG_XBR_REGULAR = 1 << 8;
constexpr int RC_FLAG_NTSC_REGULAR = 1 << 9; // ntsc pass 2: tap k of target column x reads source column c(x)+k-24, c(x+1) = c(x)+2  // xbr: sampled columns/rows are centre-2..centre+2 for every target pixel

constexpr int kFoldedTableWords = 512;   // Tex::dec: 256 floats, then 256 mapped bytes
constexpr int kMaxExtra = 8;
constexpr int kMaxPlanes = 12;
constexpr int kMaxParams = 80;

struct PassLaunch {
  Tex in;                 // the pass's "Texture" sampler
  Tex extra[kMaxExtra];   // PassPrev / alias / OrigTexture / LUT samplers, kernel specific
  void* out;
  uint64_t out_frame_stride;
  void* scratch;                 // per-pass device scratch (KernelEntry::scratch_bytes per frame), or null
  uint64_t scratch_frame_stride;
  int out_w, out_h, out_fmt;
  int src_w, src_h;       // OriginalSize
  int vp_w, vp_h;         // viewport
  int frame_count0;       // FrameCount of frame 0 of this launch; frame z sees frame_count0+z
  int n_frames;
  int flags;              // kernel specific (e.g. RC_FLAG_UNDEF_VARYING_ZERO)
  // The size uniforms as the shader reads them where they are NOT the real sizes of `in` and the target: the history push
  // re-draws the final output through pass 0's program with the uniforms of pass 0's own draw still set (reference
  // ShaderEngine.cpp:1805-1834).  TextureSize == InputSize = uni_tex, OutputSize = uni_out; 0 = the real sizes.  Only kernels
  // registered with KernelEntry::stale_size_uniforms read them.
  int uni_tex_w, uni_tex_h, uni_out_w, uni_out_h;
  const uint32_t* srgb_enc;  // per-run table of the sRGB8 encode in device memory (srgb_encode.cpp); needed when out_fmt is sRGB8
  Plane plane[kMaxPlanes];
  float params[kMaxParams];
};

// -------------------------------------------------------------------- float primitives ----
// The float primitives are __host__ __device__: the host evaluates them for per-launch
// constants (e.g. blur weights of a run-time sigma) with the very same operations.
#define RC_HD __host__ __device__ __forceinline__
RC_HD float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RC_HD uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
RC_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

RC_HD float exp2_(float x) {
  x = x > 128.0f ? 128.0f : x;
  x = x < -126.99999f ? -126.99999f : x;
  float ip = __builtin_floorf(x);
  float fp = x - ip;
  float e = bits2f((uint32_t)(((int32_t)ip + 127) << 23));
  float x2 = fp * fp;
  float even = fma_(x2, 0.00898934009049466391101f, 0.240153617044375388211f);
  even = fma_(x2, even, 1.0f);
  float odd = fma_(x2, 0.00187757667519147912699f, 0.0558263180532956664775f);
  odd = fma_(x2, odd, 0.693153073200168932794f);
  return e * fma_(odd, fp, even);
}

// num / den for the one division inside log2: num = m - 1 in [0,1), den = m + 1 in [2,3].  On the
// device: reciprocal estimate, one Newton step, quotient, two fused corrections - no scaling or
// fix-up is needed in this range.  Checked exhaustively against IEEE division for all 2^23 mantissas
// with every reciprocal seed within 1 ulp (tests/test_fastmath.py repeats it on the real v_rcp_f32).
RC_HD float div_log2_(float num, float den) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(den);
  r = fma_(fma_(-den, r, 1.0f), r, r);
  const float q = num * r;
  return fma_(fma_(-den, q, num), r, q);
#else
  return num / den;
#endif
}

// n / d, correctly rounded, for operands away from the exponent extremes: n == 0 or 2^-60 <= |n| <=
// 2^60, and 2^-60 <= |d| <= 2^60.  This is the compiler's own IEEE division sequence (reciprocal
// estimate, Newton step, quotient, three fused corrections) without the v_div_scale / v_div_fixup /
// denormal-mode bracketing, which only act on operands outside that range: same bits, about half
// the issue slots.  Callers state why their operands qualify.
RC_HD float div_safe_(float n, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(d);
  r = fma_(fma_(-d, r, 1.0f), r, r);
  float q = n * r;
  q = fma_(fma_(-d, q, n), r, q);
  return fma_(fma_(-d, q, n), r, q);
#else
  return n / d;
#endif
}

RC_HD float log2_(float x) {
  uint32_t i = f2bits(x);
  // zero / denormal -> -inf, negative -> NaN, +inf -> +inf (the GL's "safe" log2)
  if ((i & 0x7f800000u) == 0u) return -__builtin_inff();
  if (i & 0x80000000u) return __builtin_nanf("");
  if (i == 0x7f800000u) return __builtin_inff();
  float logexp = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
  float mant = bits2f((i & 0x007fffffu) | 0x3f800000u);
  float y = div_log2_(mant - 1.0f, mant + 1.0f);
  float z = y * y;
  float z2 = z * z;
  float even = fma_(z2, 0.406718052498846252698f, 0.577440339438736392009f);
  even = fma_(z2, even, 2.88539009343309178325f);
  float odd = fma_(z2, 0.403343858251329912514f, 0.961791550404184197881f);
  float p = fma_(odd, z, even);
  return fma_(y, p, logexp);
}

// log2_ without the zero / negative / infinity selects, for arguments known to be positive
// normal numbers, or where the caller proves the selects cannot change its own result.
RC_HD float log2_core_(float x) {
  uint32_t i = f2bits(x);
  float logexp = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
  float mant = bits2f((i & 0x007fffffu) | 0x3f800000u);
  float y = div_log2_(mant - 1.0f, mant + 1.0f);
  float z = y * y;
  float z2 = z * z;
  float even = fma_(z2, 0.406718052498846252698f, 0.577440339438736392009f);
  even = fma_(z2, even, 2.88539009343309178325f);
  float odd = fma_(z2, 0.403343858251329912514f, 0.961791550404184197881f);
  float p = fma_(odd, z, even);
  return fma_(y, p, logexp);
}
// x / c for a compile-time constant c with rc = RN(1/c): one multiply and one fused correction
// step give the correctly rounded quotient (checked exhaustively over every mantissa for
// c = e and c = 3; quotients scale exactly across binades away from overflow / underflow).
RC_HD float div_const_(float x, float c, float rc) {
  float q = x * rc;
  float r = fma_(-c, q, x);
  return fma_(r, rc, q);
}
RC_HD float pow_(float x, float y) { return exp2_(log2_(x) * y); }
RC_HD float exp_(float x) { return exp2_(x * 1.4426950408889634f); }
RC_HD float log_(float x) { return log2_(x) * 0.69314718055994529f; }

template <bool COS>
RC_HD float sincos_(float x) {
  uint32_t xi = f2bits(x);
  float xa = bits2f(xi & 0x7fffffffu);
  uint32_t sign = xi & 0x80000000u;
  float y = xa * 1.27323954473516f;
  int32_t j = (int32_t)y;
  j = (j + 1) & ~1;
  float y2 = (float)j;
  int32_t j2 = COS ? j - 2 : j;
  uint32_t swap = COS ? (((uint32_t)~j2 & 4u) << 29) : (((uint32_t)j & 4u) << 29);
  bool poly_sin = (j2 & 2) == 0;
  float x3 = fma_(y2, -0.78515625f, xa);
  x3 = fma_(y2, -2.4187564849853515625e-4f, x3);
  x3 = fma_(y2, -3.77489497744594108e-8f, x3);
  float z = x3 * x3;
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
  float r = poly_sin ? ys : yc;
  uint32_t sb = COS ? swap : (sign ^ swap);
  return bits2f(f2bits(r) ^ sb);
}
RC_HD float sin_(float x) { return sincos_<false>(x); }
RC_HD float cos_(float x) { return sincos_<true>(x); }

// ----------------------------------------------------------------------------- varyings ----
__device__ __forceinline__ bool lower_tri(int x, int y, int W, int H) {
  // pixel centre below the BL-TR diagonal; a centre exactly ON the diagonal belongs to the lower-right
  // triangle (measured on float targets)
  return (2 * y + 1) * W <= (2 * x + 1) * H;
}
__device__ __forceinline__ float vary(const Plane& p, int x, int y, bool lower) {
  float a0 = lower ? p.a0_lo : p.a0_up, dx = lower ? p.dx_lo : p.dx_up, dy = lower ? p.dy_lo : p.dy_up;
  return fma_(dy, (float)y, fma_(dx, (float)x, a0));
}

// ------------------------------------------------------------------------------- tables ----
// sRGB8 -> linear float as the texture unit decodes it (table measured on the GL; the
// same numbers as oracle/rc_tables.inc but owned by the product), and the sRGB8 encode of an sRGB
// render target: llvmpipe's RSQRTPS-based conversion, re-tabulated per RSQRTPS run of the argument
// (srgb_encode.cpp): byte = entry >> 16, plus one from offset (entry & 0x3fff) of the run on (bit 14: flag for
// interval tests across a run boundary).
#include "rc_tables_device.inc"  // k_srgb_decode[256]

constexpr float kSrgbLinMax = 0.0031308f;          // linear segment up to here: byte = rint(x * kSrgbLinScale)
constexpr float kSrgbLinScale = 12.92f * 255.0f;
constexpr uint32_t kSrgbRun0 = 0x3b4d2e1cu >> 13;  // run (float bits >> 13) that contains kSrgbLinMax
constexpr uint32_t kSrgbRuns = (0x3f7fffffu >> 13) - kSrgbRun0 + 2u;  // 8599 runs up to the last float below 1, and a spare entry (255 << 16) for the run of 1.0

// Second form of the encode table (srgb_encode.cpp buildSrgbRunTable2; in device memory right after the first form):
// runs from the first float that can store a non-zero byte up to the run that starts at 1.0, the linear segment included.
constexpr uint32_t kSrgb2MinBits = 0x391d4000u;    // 1.4997e-4: x * 12.92 * 255 = 0.494 still rounds to 0
constexpr uint32_t kSrgb2Run0 = kSrgb2MinBits >> 13;
constexpr uint32_t kSrgb2Runs = (0x3f800000u >> 13) - kSrgb2Run0 + 1u;   // 13 079 entries (52 KB)

// Both tables live in dynamic LDS (the launch passes srgb_lds_bytes(L)): 1 KiB for the decode table,
// plus 34 KiB for the encode table only when the pass stores to an sRGB8 target.
struct SrgbLds {
  const float* dec;
  const uint32_t* enc;
};
// A small sRGB8 target (crt-royale's 320 x 240 passes: a few hundred pixels per workgroup) reads the encode table where it is,
// in device memory (34 KB, L2-resident): copying it into every workgroup's LDS would move more bytes than the pass itself.
__host__ __device__ inline bool srgb_enc_in_lds(const PassLaunch& L) {
  return L.out_fmt == FMT_SRGB8 && (long)L.out_w * L.out_h * L.n_frames >= (1L << 21);
}
inline unsigned srgb_lds_bytes(const PassLaunch& L) { return 1024u + (srgb_enc_in_lds(L) ? kSrgbRuns * 4u : 0u); }
// Every thread of the block must call this (RC_SRGB_LDS) before sampling sRGB textures / storing sRGB.
__device__ __forceinline__ SrgbLds load_srgb_tables(uint32_t* dyn, const PassLaunch& L, const float* dec_from = nullptr) {
  float* dec = reinterpret_cast<float*>(dyn);
  const uint32_t* enc = L.srgb_enc;
  const int nt = blockDim.x * blockDim.y, t0 = threadIdx.y * blockDim.x + threadIdx.x;
  for (int i = t0; i < 256; i += nt) dec[i] = dec_from ? dec_from[i] : k_srgb_decode[i];   // (dec_from: Tex::dec of the one sRGB8 texture the kernel samples)
  if (srgb_enc_in_lds(L)) {
    for (int i = t0; i < (int)kSrgbRuns; i += nt) dyn[256 + i] = L.srgb_enc[i];
    enc = dyn + 256;
  }
  __syncthreads();
  return SrgbLds{dec, enc};
}
#define RC_SRGB_LDS(name, L)                  \
  extern __shared__ uint32_t rc_dyn_lds_[];   \
  const rcd::SrgbLds name = rcd::load_srgb_tables(rc_dyn_lds_, (L))
// ... for a kernel whose only sRGB8 texture is `tex`, which may be a folded pass's view (Tex::dec)
#define RC_SRGB_LDS_OF(name, L, tex)          \
  extern __shared__ uint32_t rc_dyn_lds_[];   \
  const rcd::SrgbLds name = rcd::load_srgb_tables(rc_dyn_lds_, (L), (tex).dec)

// ------------------------------------------------------------------------------ sampling ----
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int modi(int v, int n) {
  int r = v % n;
  return r < 0 ? r + n : r;
}
template <int WRAP>
__device__ __forceinline__ int wrap_index(int i, int n) {
  if (WRAP == WRAP_REPEAT) return modi(i, n);
  if (WRAP == WRAP_MIRROR) {
    int p = modi(i, 2 * n);
    return p < n ? p : 2 * n - 1 - p;
  }
  if (WRAP == WRAP_BORDER) return (i < 0 || i >= n) ? -1 : i;
  return clampi(i, 0, n - 1);
}

// Byte offset of texel (x, y) inside one image: 32-bit arithmetic (the engine refuses frames of
// 4 GiB or more), so the address is a uniform 64-bit base plus a 32-bit lane offset instead of a
// 64-bit multiply-add per fetch.
__device__ __forceinline__ uint32_t texel_off(int w, int x, int y, uint32_t bytes) { return ((uint32_t)y * (uint32_t)w + (uint32_t)x) * bytes; }

template <int FMT>
__device__ __forceinline__ float4 texel(const Tex& t, const uint8_t* img, int x, int y, const SrgbLds* lds) {
  if (FMT == FMT_F32) {
    return *reinterpret_cast<const float4*>(img + texel_off(t.w, x, y, 16u));
  }
  if (FMT == FMT_F16) {
    const half4_t h = *reinterpret_cast<const half4_t*>(img + texel_off(t.w, x, y, 8u));
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
  uint32_t p = *reinterpret_cast<const uint32_t*>(img + texel_off(t.w, x, y, 4u));
  const float k = 1.0f / 255.0f;
  uint32_t r = p & 255u, g = (p >> 8) & 255u, b = (p >> 16) & 255u, a = p >> 24;
  if (FMT == FMT_SRGB8) return make_float4(lds->dec[r], lds->dec[g], lds->dec[b], t.alpha_one ? 1.0f : (float)a * k);
  if (FMT == FMT_RGBX8) return make_float4((float)r * k, (float)g * k, (float)b * k, 1.0f);
  return make_float4((float)r * k, (float)g * k, (float)b * k, (float)a * k);
}

template <int FMT, int WRAP>
__device__ __forceinline__ float4 fetch_wrapped(const Tex& t, const uint8_t* img, int x, int y, const SrgbLds* lds) {
  int xi = wrap_index<WRAP>(x, t.w), yi = wrap_index<WRAP>(y, t.h);
  if (WRAP == WRAP_BORDER && (xi < 0 || yi < 0)) return make_float4(0.f, 0.f, 0.f, 0.f);
  return texel<FMT>(t, img, xi, yi, lds);
}

__device__ __forceinline__ float lerp_(float w, float a, float b) { return fma_(w, b - a, a); }

template <int WRAP>
__device__ __forceinline__ float linear_coord(float s, int n) {
  if (WRAP == WRAP_REPEAT) s = s - __builtin_floorf(s);
  float u = s * (float)n;
  // CLAMP_TO_EDGE as llvmpipe orders it: min(u, n) first - which is n for a NaN coordinate (MINPS there, minnum here), so
  // a NaN coordinate filters at the last texel - then - 0.5, then max(.., 0)
  if (WRAP == WRAP_EDGE) return fmaxf(fminf(u, (float)n) - 0.5f, 0.0f);
  return u - 0.5f;
}

// The same coordinate for the finite inputs of the host-verified table / strip forms: clamp(s n, 0, n) - 0.5 keeps the texel pair
// (-1, 0) with weight 0.5 at the left / top edge where linear_coord's max(.., 0) gives (0, 1) with weight 0 - the same
// filtered value, both texels of the pair being texel 0 after the index clamp - and the pair those tables are written for.
__device__ __forceinline__ float linear_coord_edge_pair(float s, int n) { return fminf(fmaxf(s * (float)n, 0.0f), (float)n) - 0.5f; }

// GL_NEAREST
template <int FMT, int WRAP>
__device__ __forceinline__ float4 sample_nearest(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* lds) {
  if (WRAP == WRAP_REPEAT) {
    s = s - __builtin_floorf(s);
    v = v - __builtin_floorf(v);
  }
  // a NaN coordinate converts to INT_MIN on the GL's CPU: the border (clamp-to-edge, like the (int)NaN = 0 here, lands on texel 0)
  if (WRAP == WRAP_BORDER && (s != s || v != v)) return make_float4(0.f, 0.f, 0.f, 0.f);
  int x = (int)__builtin_floorf(s * (float)t.w), y = (int)__builtin_floorf(v * (float)t.h);
  if (WRAP == WRAP_REPEAT) {
    x = clampi(x, 0, t.w - 1);
    y = clampi(y, 0, t.h - 1);
  }
  return fetch_wrapped<FMT, WRAP>(t, img, x, y, lds);
}

// GL_LINEAR, float filter path (sRGB8 / F32 textures, or clamp_to_border)
template <int FMT, int WRAP>
__device__ __forceinline__ float4 sample_linear_f(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* lds) {
  float u = linear_coord<WRAP>(s, t.w), w = linear_coord<WRAP>(v, t.h);
  float x0f = __builtin_floorf(u), y0f = __builtin_floorf(w);
  float wx = u - x0f, wy = w - y0f;
  int x0 = (int)x0f, y0 = (int)y0f;
  float4 a = fetch_wrapped<FMT, WRAP>(t, img, x0, y0, lds), b = fetch_wrapped<FMT, WRAP>(t, img, x0 + 1, y0, lds);
  float4 c = fetch_wrapped<FMT, WRAP>(t, img, x0, y0 + 1, lds), d = fetch_wrapped<FMT, WRAP>(t, img, x0 + 1, y0 + 1, lds);
  return make_float4(lerp_(wy, lerp_(wx, a.x, b.x), lerp_(wx, c.x, d.x)), lerp_(wy, lerp_(wx, a.y, b.y), lerp_(wx, c.y, d.y)),
                     lerp_(wy, lerp_(wx, a.z, b.z), lerp_(wx, c.z, d.z)), lerp_(wy, lerp_(wx, a.w, b.w), lerp_(wx, c.w, d.w)));
}

// GL_LINEAR, 8-bit fixed-point filter path (RGBA8 / RGBX8 with edge / repeat / mirror wrap)
template <int FMT, int WRAP>
__device__ __forceinline__ float4 sample_linear_u8(const Tex& t, const uint8_t* img, float s, float v) {
  if (WRAP == WRAP_REPEAT) {
    s = s - __builtin_floorf(s);
    v = v - __builtin_floorf(v);
  }
  float u = s * (float)t.w - 0.5f, w = v * (float)t.h - 0.5f;
  if (WRAP == WRAP_EDGE) {
    u = fminf(fmaxf(u, 0.0f), (float)(t.w - 1));
    w = fminf(fmaxf(w, 0.0f), (float)(t.h - 1));
  }
  float x0f = __builtin_floorf(u), y0f = __builtin_floorf(w);
  int wx = (int)__builtin_rintf((u - x0f) * 256.0f), wy = (int)__builtin_rintf((w - y0f) * 256.0f);
  int x0 = wrap_index<WRAP>((int)x0f, t.w), x1 = wrap_index<WRAP>((int)x0f + 1, t.w);
  int y0 = wrap_index<WRAP>((int)y0f, t.h), y1 = wrap_index<WRAP>((int)y0f + 1, t.h);
  uint32_t p00 = *reinterpret_cast<const uint32_t*>(img + texel_off(t.w, x0, y0, 4u));
  uint32_t p10 = *reinterpret_cast<const uint32_t*>(img + texel_off(t.w, x1, y0, 4u));
  uint32_t p01 = *reinterpret_cast<const uint32_t*>(img + texel_off(t.w, x0, y1, 4u));
  uint32_t p11 = *reinterpret_cast<const uint32_t*>(img + texel_off(t.w, x1, y1, 4u));
  // a + ((w*(b - a) + 128) >> 8) == (a*(256 - w) + b*w + 128) >> 8 with every term non-negative and at most
  // 255*256 + 128 < 2^16: two channels are filtered per 32-bit operation (red|blue and green|alpha lanes).
  const uint32_t ux = (uint32_t)wx, uy = (uint32_t)wy, M = 0x00ff00ffu, R = 0x00800080u;
  const uint32_t top_rb = (((p00 & M) * (256u - ux) + (p10 & M) * ux + R) >> 8) & M;
  const uint32_t bot_rb = (((p01 & M) * (256u - ux) + (p11 & M) * ux + R) >> 8) & M;
  const uint32_t rb = ((top_rb * (256u - uy) + bot_rb * uy + R) >> 8) & M;
  const uint32_t top_ga = ((((p00 >> 8) & M) * (256u - ux) + ((p10 >> 8) & M) * ux + R) >> 8) & M;
  const uint32_t bot_ga = ((((p01 >> 8) & M) * (256u - ux) + ((p11 >> 8) & M) * ux + R) >> 8) & M;
  const uint32_t ga = ((top_ga * (256u - uy) + bot_ga * uy + R) >> 8) & M;
  const float k = 1.0f / 255.0f;
  return make_float4((float)(rb & 255u) * k, (float)(ga & 255u) * k, (float)(rb >> 16) * k,
                     FMT == FMT_RGBX8 ? 1.0f : (float)(ga >> 16) * k);
}

template <int FMT, int LINEAR, int WRAP>
__device__ __forceinline__ float4 sample(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* lds) {
  if (!LINEAR) return sample_nearest<FMT, WRAP>(t, img, s, v, lds);
  // 8-bit fixed-point filter for RGBA8 / RGBX8 with clamp-to-edge or repeat; border and mirrored
  // repeat filter in float (measured on the GL, tests/golden/wrap_*)
  if ((FMT == FMT_RGBA8 || FMT == FMT_RGBX8) && WRAP != WRAP_BORDER && WRAP != WRAP_MIRROR) return sample_linear_u8<FMT, WRAP>(t, img, s, v);
  return sample_linear_f<FMT, WRAP>(t, img, s, v, lds);
}

__device__ __forceinline__ int texel_bytes(int fmt) { return fmt == FMT_F32 ? 16 : (fmt == FMT_F16 ? 8 : 4); }
__device__ __forceinline__ const uint8_t* frame_ptr(const Tex& t, int z) {
  return static_cast<const uint8_t*>(t.base) + t.frame_stride * (uint64_t)z;
}

// --------------------------------------------------------------------------------- store ----
__device__ __forceinline__ uint32_t unorm8(float x) {
  if (!(x > 0.0f)) return 0u;
  x = x > 1.0f ? 1.0f : x;
  return (uint32_t)__builtin_rintf(x * 255.0f);
}
// sRGB8 encode (see the tables section): one LDS entry per RSQRTPS run of x.
__device__ __forceinline__ uint32_t srgb8(float x, const SrgbLds* t) {
  if (!(x > 0.0f)) return 0u;  // also NaN
  if (x >= 1.0f) return 255u;
  if (x <= kSrgbLinMax) return (uint32_t)__builtin_rintf(x * kSrgbLinScale);
  const uint32_t b = f2bits(x);
  const uint32_t e = t->enc[(b >> 13) - kSrgbRun0];
  return (e >> 16) + ((b & 0x1fffu) >= (e & 0x3fffu) ? 1u : 0u);
}

// The same byte from the second form of the table (`enc2`: kSrgb2Runs entries in LDS): clamp into the table's range - a NaN
// comes out of v_med3_f32 as the smallest operand, the lower bound, whose byte is 0 like srgb8's - one LDS read, one add.
__device__ __forceinline__ uint32_t srgb8_t2(float x, const uint32_t* enc2) {
  const uint32_t b = f2bits(__builtin_amdgcn_fmed3f(x, bits2f(kSrgb2MinBits), 1.0f));
  const uint32_t e = enc2[(b >> 13) - kSrgb2Run0];
  return ((e + (b & 0x1fffu)) >> 13) & 255u;
}

template <int OUT_FMT>
__device__ __forceinline__ void store(const PassLaunch& L, int z, int x, int y, float4 c, const SrgbLds* t) {
  uint8_t* o = static_cast<uint8_t*>(L.out) + L.out_frame_stride * (uint64_t)z;
  if (OUT_FMT == FMT_F32) {
    *reinterpret_cast<float4*>(o + texel_off(L.out_w, x, y, 16u)) = c;
  } else if (OUT_FMT == FMT_F16) {
    const half4_t h = {(_Float16)c.x, (_Float16)c.y, (_Float16)c.z, (_Float16)c.w};   // round to nearest even
    *reinterpret_cast<half4_t*>(o + texel_off(L.out_w, x, y, 8u)) = h;
  } else if (OUT_FMT == FMT_SRGB8) {
    uint32_t p = srgb8(c.x, t) | (srgb8(c.y, t) << 8) | (srgb8(c.z, t) << 16) | (unorm8(c.w) << 24);
    *reinterpret_cast<uint32_t*>(o + texel_off(L.out_w, x, y, 4u)) = p;
  } else {
    uint32_t p = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
    *reinterpret_cast<uint32_t*>(o + texel_off(L.out_w, x, y, 4u)) = p;
  }
}

}  // namespace rcd

// Grid-stride walk over the 64x4 tiles of all frames of a launch (see pass_launch.h).  Defines
// x, y, z (frame) and `lo` (triangle of the quad the pixel centre falls in) for the body.
#define RC_TILE_LOOP_BEGIN                                                              \
  const int tiles_x_ = (L.out_w + 63) >> 6, tiles_y_ = (L.out_h + 3) >> 2;               \
  const int tiles_per_frame_ = tiles_x_ * tiles_y_;                                     \
  const int n_tiles_ = tiles_per_frame_ * L.n_frames;                                   \
  for (int tile_ = blockIdx.x; tile_ < n_tiles_; tile_ += gridDim.x) {                  \
    const int z = tile_ / tiles_per_frame_;                                             \
    const int rem_ = tile_ - z * tiles_per_frame_;                                      \
    const int ty_ = rem_ / tiles_x_;                                                    \
    const int x = (rem_ - ty_ * tiles_x_) * 64 + (int)threadIdx.x, y = ty_ * 4 + (int)threadIdx.y; \
    if (x >= L.out_w || y >= L.out_h) continue;                                         \
    const bool lo = rcd::lower_tri(x, y, L.out_w, L.out_h);
#define RC_TILE_LOOP_END }

namespace rcd {
// Run-time selected sampler (all selectors are wave-uniform kernel arguments, so the
// branches are scalar).  Hot kernels instantiate sample<> directly instead.
template <int FMT, int LINEAR>
__device__ __forceinline__ float4 sample_rt_wrap(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* lds) {
  switch (t.wrap) {
    case WRAP_BORDER: return sample<FMT, LINEAR, WRAP_BORDER>(t, img, s, v, lds);
    case WRAP_REPEAT: return sample<FMT, LINEAR, WRAP_REPEAT>(t, img, s, v, lds);
    case WRAP_MIRROR: return sample<FMT, LINEAR, WRAP_MIRROR>(t, img, s, v, lds);
    default: return sample<FMT, LINEAR, WRAP_EDGE>(t, img, s, v, lds);
  }
}
template <int FMT>
__device__ __forceinline__ float4 sample_rt_filter(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* lds) {
  return t.linear ? sample_rt_wrap<FMT, 1>(t, img, s, v, lds) : sample_rt_wrap<FMT, 0>(t, img, s, v, lds);
}
__device__ __forceinline__ float4 sample_rt(const Tex& t, const uint8_t* img, float s, float v, const SrgbLds* lds) {
  switch (t.fmt) {
    case FMT_SRGB8: return sample_rt_filter<FMT_SRGB8>(t, img, s, v, lds);
    case FMT_RGBX8: return sample_rt_filter<FMT_RGBX8>(t, img, s, v, lds);
    case FMT_F32: return sample_rt_filter<FMT_F32>(t, img, s, v, lds);
    case FMT_F16: return sample_rt_filter<FMT_F16>(t, img, s, v, lds);
    default: return sample_rt_filter<FMT_RGBA8>(t, img, s, v, lds);
  }
}
__device__ __forceinline__ void store_rt(const PassLaunch& L, int z, int x, int y, float4 c, const SrgbLds* t) {
  if (L.out_fmt == FMT_F32) store<FMT_F32>(L, z, x, y, c, t);
  else if (L.out_fmt == FMT_F16) store<FMT_F16>(L, z, x, y, c, t);
  else if (L.out_fmt == FMT_SRGB8) store<FMT_SRGB8>(L, z, x, y, c, t);
  else store<FMT_RGBA8>(L, z, x, y, c, t);
}
// ------------------------------------------------------------------- mip-mapped sampling ----
// texture() on a mip-mapped input (GL_LINEAR_MIPMAP_LINEAR, ShaderEngine.cpp:1022-1033) as llvmpipe evaluates
// it: rho^2 from ONE set of coordinate differences per 2x2 quad, taken at its top-left pixel (callers with separable
// coordinates pass the differences along the pixel's own row / column: the same floats), lod = 0.5 * (exponent + mantissa - 1)
// of rho^2, the two nearest levels filtered LINEAR and blended with fma (oracle/rc_sampler.c).
__device__ __forceinline__ float fast_log2_(float x) {
  const uint32_t b = f2bits(x);
  const int e = (int)((b >> 23) & 255u) - 127;
  return (float)e + (bits2f((b & 0x7fffffu) | 0x3f800000u) - 1.0f);
}
__device__ __forceinline__ Tex mip_level(const Tex& t, int z, int level, const uint8_t** img) {
  Tex l = t;
  if (level == 0) {
    *img = frame_ptr(t, z);
    return l;
  }
  uint64_t off = 0;
  for (int k = 1; k < level; ++k) off += (uint64_t)max(t.w >> k, 1) * (uint64_t)max(t.h >> k, 1) * (uint64_t)texel_bytes(t.fmt);
  l.w = max(t.w >> level, 1);
  l.h = max(t.h >> level, 1);
  *img = static_cast<const uint8_t*>(t.mip_base) + t.mip_frame_stride * (uint64_t)z + off;
  return l;
}
__device__ __forceinline__ float lod_from_quad(const Tex& t, float s_dx0, float s_dx1, float v_dx0, float v_dx1, float s_dy0,
                                               float s_dy1, float v_dy0, float v_dy1) {
  const float fw = (float)t.w, fh = (float)t.h;
  const float ax = (s_dx1 - s_dx0) * fw, bx = (v_dx1 - v_dx0) * fh;
  const float ay = (s_dy1 - s_dy0) * fw, by = (v_dy1 - v_dy0) * fh;
  const float rx = ax * ax + bx * bx, ry = ay * ay + by * by;
  if (!t.linear) {
    // GL_NEAREST_MIPMAP_NEAREST (mipmap_input without filter_linear): the level itself, (exponent(rho^2) + 1) >> 1 - not
    // round(lod) of the float below, which differs one ulp below an odd power of two (oracle/rc_sampler.c, measured)
    const float rho2 = rx > ry ? rx : ry;
    const int lv = ((int)((f2bits(rho2) >> 23) & 255u) - 126) >> 1;
    return (float)(rho2 > 0.0f ? min(max(lv, 0), t.n_levels - 1) : 0);
  }
  float lod = 0.5f * fast_log2_(rx > ry ? rx : ry);
  if (!(lod > 0.0f)) lod = 0.0f;
  return fminf(lod, (float)(t.n_levels - 1));
}
__device__ __forceinline__ float4 sample_mip(const Tex& t, int z, float s, float v, float lod, const SrgbLds* lds) {
  const float fl = __builtin_floorf(lod), w = lod - fl;
  const int l0 = (int)fl, l1 = min(l0 + 1, t.n_levels - 1);
  const uint8_t *i0, *i1;
  if (!t.linear) {   // one level, NEAREST (lod_from_quad returned the level)
    const Tex tn = mip_level(t, z, l0, &i0);
    return sample_rt(tn, i0, s, v, lds);
  }
  const Tex t0 = mip_level(t, z, l0, &i0), t1 = mip_level(t, z, l1, &i1);
  const float4 c0 = sample_rt(t0, i0, s, v, lds), c1 = sample_rt(t1, i1, s, v, lds);
  // RGBA8 / GL_RGB textures on the 8-bit filter path: the blend between the two level samples (bytes) is 8-bit too,
  // weight floor(frac(lod) * 256), a + ((w (b - a) + 128) >> 8)  (oracle/rc_sampler.c o_sample_quad, measured)
  if (t.linear && (t.fmt == FMT_RGBA8 || t.fmt == FMT_RGBX8) && t.wrap != WRAP_BORDER && t.wrap != WRAP_MIRROR) {
    const int w8 = (int)__builtin_floorf(w * 256.0f);
    const float a[4] = {c0.x, c0.y, c0.z, c0.w}, b[4] = {c1.x, c1.y, c1.z, c1.w};
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int a8 = (int)__builtin_rintf(a[c] * 255.0f), b8 = (int)__builtin_rintf(b[c] * 255.0f);
      o[c] = (float)((a8 + ((w8 * (b8 - a8) + 128) >> 8)) & 255) * (1.0f / 255.0f);
    }
    return make_float4(o[0], o[1], o[2], o[3]);
  }
  return make_float4(fma_(w, c1.x - c0.x, c0.x), fma_(w, c1.y - c0.y, c0.y), fma_(w, c1.z - c0.z, c0.z), fma_(w, c1.w - c0.w, c0.w));
}

}  // namespace rcd
