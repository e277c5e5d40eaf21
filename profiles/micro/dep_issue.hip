// How often can ONE wave issue VALU instructions, and what does that make of a SIMD with few waves?  Chains of dependent v_fma_f32
// (and v_pk_fma_f32), C independent chains interleaved per wave (C = 1: every instruction waits for the one before it), at 1 - 4
// waves per SIMD (one workgroup of 256 - 1024 threads per CU).  Prints cycles per wave64 instruction per SIMD at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/dep_issue profiles/micro/dep_issue.hip && /tmp/dep_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float __attribute__((ext_vector_type(2))) f2;

template <int C, bool PK>
__global__ void __launch_bounds__(1024) k(float* out, int reps, float m, float a) {
  float u[8];
  f2 w[8];
  for (int i = 0; i < 8; ++i) { u[i] = (float)(threadIdx.x + i); w[i] = f2{u[i], u[i] + 1.f}; }
  const f2 mm = {m, m}, aa = {a, a};
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < 64 / C; ++j) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[c]) : "v"(mm), "v"(aa));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(u[c]) : "v"(m), "v"(a));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += u[i] + w[i].x + w[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C, bool PK>
void run(float* d) {
  for (int threads = 256; threads <= 1024; threads += 256) {
    const int reps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<C, PK>), dim3(256), dim3(threads), 0, 0, d, reps, 0.999f, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<C, PK>), dim3(256), dim3(threads), 0, 0, d, reps, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)reps * 64.0 * (threads / 256);   // wave-instructions per SIMD
    printf("%s chains %d waves/SIMD %d: %.2f cycles per instruction per SIMD (%.2f per wave)\n", PK ? "v_pk_fma_f32" : "v_fma_f32   ", C, threads / 256,
           ms * 1e-3 * 2.4e9 / insts_per_simd, ms * 1e-3 * 2.4e9 / ((double)reps * 64.0));
  }
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<1, false>(d); run<2, false>(d); run<4, false>(d); run<8, false>(d);
  run<1, true>(d); run<2, true>(d); run<4, true>(d); run<8, true>(d);
  return 0;
}
