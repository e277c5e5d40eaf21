// Micro-benchmark behind DESIGN.md section 7's cost model: do a wave's LDS reads overlap with its VALU work, or do the returned
// dwords compete with VALU results?  One workgroup per CU, W waves per SIMD; per wave REPS rounds of NV dependent-free FMAs and NL
// conflict-free ds_read_b128 (or b32).  Build / run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_valu profiles/micro/lds_valu_overlap.hip && /tmp/lds_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NV, int NL, bool B128>
__global__ void __launch_bounds__(1024) k(float* out, int reps) {
  __shared__ v4f lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = v4f{(float)i, 1.0f, 2.0f, 3.0f};
  __syncthreads();
  float a0 = tid * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v4f acc = {0, 0, 0, 0};
  int idx = tid & 63;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (B128) {
        const v4f e = lds[(idx + i * 64) & 4095];
        acc += e;
      } else {
        acc.x += reinterpret_cast<const float*>(lds)[(idx + i * 64) & 16383];
      }
    }
#pragma unroll
    for (int i = 0; i < NV / 8; ++i) {
      a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
      a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
    }
    idx = (idx + 1) & 63;
  }
  out[blockIdx.x * blockDim.x + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc.x + acc.y + acc.z + acc.w;
}

template <int NV, int NL, bool B128>
float run(int waves_per_cu, int reps, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NL, B128>), dim3(256), dim3(waves_per_cu * 64), 0, 0, d, reps);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NL, B128>), dim3(256), dim3(waves_per_cu * 64), 0, 0, d, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  const int reps = 2000;
  for (int w : {4, 8, 12, 16}) {
    const float v = run<128, 0, true>(w, reps, d), l = run<0, 16, true>(w, reps, d), b = run<128, 16, true>(w, reps, d);
    const float l32 = run<0, 16, false>(w, reps, d), b32 = run<128, 16, false>(w, reps, d);
    const double insts = (double)reps * 128 * w / 4;   // VALU instructions per SIMD
    printf("waves/CU %2d: VALU-only %.3f ms (%.2f cyc/inst @2.4GHz)  b128-only %.3f ms  both %.3f ms (sum %.3f, max %.3f) | b32-only %.3f  both %.3f\n", w, v,
           v * 1e-3 * 2.4e9 / insts, l, b, v + l, v > l ? v : l, l32, b32);
  }
  return 0;
}
