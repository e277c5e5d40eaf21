// Shared by the crt-royale strip kernels that render TWO pixels per lane - columns x and x + 64 of a 128-column band -
// so that the float work of a pass runs as packed v_pk_{fma,mul,add}_f32 over the pixel pair (the same IEEE operation per
// component): LDS access through absolute offsets, the sRGB decode / encode tables at fixed places, buffer addressing.
//
// LDS layout of these kernels (no static LDS, so their dynamic LDS starts at offset 0 - checked once at kernel entry by
// strip2_load_tables): the 256-entry sRGB decode table at 0, the second form of the sRGB8 encode table (rc_device.h
// kSrgb2*) at 1024, the kernel's own data from kStrip2LdsUser on.
#pragma once
#include "royale_strip.h"

namespace rcstrip2 {
using namespace rcd;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));

#define RC_AS3 __attribute__((address_space(3)))
__device__ __forceinline__ float lds_f32(uint32_t off) { return *reinterpret_cast<const RC_AS3 float*>((uintptr_t)off); }
__device__ __forceinline__ uint32_t lds_u32(uint32_t off) { return *reinterpret_cast<const RC_AS3 uint32_t*>((uintptr_t)off); }
__device__ __forceinline__ v4f lds_v4f(uint32_t off) { return *reinterpret_cast<const RC_AS3 v4f*>((uintptr_t)off); }
__device__ __forceinline__ void lds_put_v4f(uint32_t off, v4f v) { *reinterpret_cast<RC_AS3 v4f*>((uintptr_t)off) = v; }
// (Replicating the decode table against gather bank conflicts was measured and rejected: 8 copies made P7 / P8 slower -
// neighbouring lanes of an image row hold equal or close bytes, which one table serves by broadcast or adjacent banks.)
constexpr uint32_t kStrip2LdsDec = 0u, kStrip2LdsEnc = 1024u;
constexpr uint32_t kStrip2LdsUser = ((kStrip2LdsEnc + kSrgb2Runs * 4u) + 15u) & ~15u;   // first free byte behind the two tables

// Both tables into LDS (every thread of the workgroup calls this first); `with_encode`: the pass stores to an sRGB8 target.
__device__ __forceinline__ void strip2_load_tables(uint32_t* dyn, const PassLaunch& L, bool with_encode) {
  if ((uint32_t)(uintptr_t)(RC_AS3 uint32_t*)dyn != 0u) __builtin_trap();
  const int nt = blockDim.x * blockDim.y, t0 = threadIdx.y * blockDim.x + threadIdx.x;
  for (int i = t0; i < 256; i += nt) dyn[i] = f2bits(k_srgb_decode[i]);
  if (with_encode)
    for (int i = t0; i < (int)kSrgb2Runs; i += nt) dyn[kStrip2LdsEnc / 4 + i] = L.srgb_enc[kSrgbRuns + i];
  __syncthreads();
}

// value of the next lane (lane 63: 0); the compiler folds the move into the subtraction that consumes it (v_sub_f32_dpp)
__device__ __forceinline__ float next_lane_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}
// byte N of t, shifted left by SH (one SDWA instruction): the offset of its entry in a table of 2^SH-byte entries
template <int N, int SH>
__device__ __forceinline__ uint32_t byte_shl(uint32_t t) {
  uint32_t r;
  static_assert(N >= 0 && N <= 3 && SH >= 1 && SH <= 8, "byte_shl");
  asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_%3" : "=v"(r) : "v"(t), "n"(SH), "n"(N));
  return r;
}
// the sRGB-decoded value of byte N of texel t (the decode table sits at LDS offset 0)
template <int N>
__device__ __forceinline__ float dec_byte(uint32_t t) { return lds_f32(kStrip2LdsDec + byte_shl<N, 2>(t)); }
template <int N>
__device__ __forceinline__ float dec_byte_plain(uint32_t t) { return dec_byte<N>(t); }   // (kernels that keep RC_SRGB_LDS's layout: the same offset)
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float x) { return v2f{x, x}; }
// srgb8_t2 (rc_device.h) on the table at kStrip2LdsEnc
__device__ __forceinline__ uint32_t srgb8_lds(float x) {
  const uint32_t b = f2bits(__builtin_amdgcn_fmed3f(x, bits2f(kSrgb2MinBits), 1.0f));
  const uint32_t e = lds_u32(((b >> 13) << 2) + (kStrip2LdsEnc - (kSrgb2Run0 << 2)));
  return ((e + (b & 0x1fffu)) >> 13) & 255u;
}
// three channels of a pixel pair into the two RGBA8 texels (alpha 255)
__device__ __forceinline__ void srgb8_pack2(const v2f* o, uint32_t* pa, uint32_t* pb) {
  *pa = 0xff000000u | srgb8_lds(o[0].x) | (srgb8_lds(o[1].x) << 8) | (srgb8_lds(o[2].x) << 16);
  *pb = 0xff000000u | srgb8_lds(o[0].y) | (srgb8_lds(o[1].y) << 8) | (srgb8_lds(o[2].y) << 16);
}
// n / d per component with div_safe_'s operations (rc_device.h: operands away from the exponent extremes), packed
__device__ __forceinline__ v2f div_safe2(v2f n, v2f d) {
  v2f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  r = fma2(fma2(-d, r, splat2(1.0f)), r, r);
  v2f q = n * r;
  q = fma2(fma2(-d, q, n), r, q);
  return fma2(fma2(-d, q, n), r, q);
}

// a frame of 4-byte texels as a raw buffer (wave-uniform base, 32-bit offsets, out-of-range accesses dropped)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const void* base, int w, int h) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, w * h * 4, 0x00020000);
}

}  // namespace rcstrip2
