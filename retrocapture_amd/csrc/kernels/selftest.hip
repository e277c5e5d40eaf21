// Device self-test of the division shortcuts in rc_device.h against the compiler's IEEE division,
// on the real v_rcp_f32 of the device it runs on (exposed as rc_selftest_fastmath in the C ABI).
#include "pass_launch.h"

using namespace rcd;

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // xorshift-multiply hash: test operand generator
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// counts[0]: div_log2_ over all 2^23 mantissas; counts[1]: div_safe_ on 2^26 operand pairs with
// exponents in [-60, 60] plus exact zeros; counts[2]: div_const_ for 3, e and the xbr widths
__global__ void __launch_bounds__(256) k_selftest(unsigned long long* counts) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  unsigned long long bad0 = 0, bad1 = 0, bad2 = 0;
  for (uint32_t m = gid; m < (1u << 23); m += stride) {
    const float mant = bits2f(0x3f800000u | m);
    const float n = mant - 1.0f, d = mant + 1.0f;
    if (f2bits(div_log2_(n, d)) != f2bits(n / d)) ++bad0;
  }
  for (uint32_t i = gid; i < (1u << 26); i += stride) {
    const uint32_t a = mix32(i * 2u + 1u), b = mix32(i * 2u + 0x9e3779b9u);
    const uint32_t ea = 127u - 60u + (a >> 23) % 121u, eb = 127u - 60u + (b >> 23) % 121u;
    float n = bits2f((a & 0x807fffffu) | (ea << 23));
    const float d = bits2f((b & 0x807fffffu) | (eb << 23));
    if ((i & 1023u) == 0u) n = 0.0f;
    if (f2bits(div_safe_(n, d)) != f2bits(n / d)) ++bad1;
  }
  const float cs[6] = {3.0f, 2.71828182845904523536f, 0.79999995f, 0.8f, 0.80000007f, 0.8000002f};
  for (uint32_t i = gid; i < (1u << 24); i += stride) {
    const float x = bits2f(0x3d000000u + (i << 3));  // mantissas across several binades
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (f2bits(div_const_(x, cs[k], 1.0f / cs[k])) != f2bits(x / cs[k])) ++bad2;
  }
  if (bad0) atomicAdd(&counts[0], bad0);
  if (bad1) atomicAdd(&counts[1], bad1);
  if (bad2) atomicAdd(&counts[2], bad2);
}

// the sRGB8 store of every pass kernel (rc_device.h srgb8) on a flat array of floats
__global__ void __launch_bounds__(256) k_selftest_srgb8(const float* src, uint8_t* dst, size_t n, const PassLaunch L) {
  RC_SRGB_LDS(lds, L);
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) dst[i] = (uint8_t)srgb8(src[i], &lds);
}

// ... and the second form of the table (rc_device.h srgb8_t2: the strip kernels' encode)
__global__ void __launch_bounds__(256) k_selftest_srgb8_t2(const float* src, uint8_t* dst, size_t n, const uint32_t* __restrict__ table2) {
  extern __shared__ uint32_t enc2[];
  for (uint32_t i = threadIdx.x; i < kSrgb2Runs; i += 256u) enc2[i] = table2[i];
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) dst[i] = (uint8_t)srgb8_t2(src[i], enc2);
}

// the box's streaming ceiling: 16 bytes per lane in, 16 out, grid-stride, four workgroups per CU (profiles/micro/copy_rate.hip: the
// fastest of the grid sizes and unroll factors tried, 5.6 TB/s; the hardware guide quotes 6.3 TB/s for this kind of kernel)
__global__ void __launch_bounds__(256) k_selftest_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}

}  // namespace

namespace rck {
hipError_t launch_selftest_copy(const void* d_src, void* d_dst, size_t bytes, hipStream_t s) {
  hipLaunchKernelGGL(k_selftest_copy, dim3(1024), dim3(256), 0, s, static_cast<const uint4*>(d_src), static_cast<uint4*>(d_dst), bytes / 16);
  return hipGetLastError();
}
hipError_t launch_selftest_srgb8(const float* d_src, uint8_t* d_dst, size_t n, const uint32_t* table, hipStream_t s, int form) {
  if (form == 2) {
    hipLaunchKernelGGL(k_selftest_srgb8_t2, dim3(512), dim3(256), kSrgb2Runs * 4u, s, d_src, d_dst, n, table + kSrgbRuns);
    return hipGetLastError();
  }
  PassLaunch L = {};
  L.out_fmt = FMT_SRGB8;
  L.srgb_enc = table;
  // form 1: the table where it is, in device memory (what a small sRGB8 target does); form 3: a target large enough for
  // load_srgb_tables to copy the table into every workgroup's LDS (rc_device.h srgb_enc_in_lds) - the path of every 1080p pass
  if (form == 3) {
    L.out_w = 2048;
    L.out_h = 1024;
    L.n_frames = 1;
    if (!rcd::srgb_enc_in_lds(L)) return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(k_selftest_srgb8, dim3(1024), dim3(256), rcd::srgb_lds_bytes(L), s, d_src, d_dst, n, L);
  return hipGetLastError();
}
hipError_t launch_selftest(unsigned long long* d_counts, hipStream_t s) {
  hipLaunchKernelGGL(k_selftest, dim3(4096), dim3(256), 0, s, d_counts);
  return hipGetLastError();
}
}  // namespace rck
