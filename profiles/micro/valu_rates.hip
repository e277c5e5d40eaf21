// Issue cost of VALU instruction classes on gfx950 (cycles per wave64 instruction per SIMD at 16 waves per CU, eight independent
// chains per wave), for DESIGN.md section 7's cost model.  Build / run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates profiles/micro/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(1024) k(float* out, int reps, float c, uint32_t m) {
  const int tid = threadIdx.x;
  float a[8];
  v2f p[8];
  uint32_t u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = tid * 0.001f + i;
    p[i] = v2f{a[i], a[i] + 0.5f};
    u[i] = (uint32_t)tid * 2654435761u + i;
  }
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) a[i] = __builtin_fmaf(a[i], c, 0.5f);
        if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], v2f{c, c}, v2f{0.5f, 0.25f});
        if (OP == 2) u[i] = (u[i] + m) & 0x7fffffffu;                      // two integer instructions
        if (OP == 3) a[i] = a[i] > c ? a[i] - 1.0f : a[i] + c;              // compare + select + two adds (some fold)
        if (OP == 4) u[i] = u[i] * m + 1u;                                  // v_mul_lo_u32 (+ add)
        if (OP == 5) a[i] = __builtin_amdgcn_exp2f(a[i]) * c;               // transcendental + mul
        if (OP == 6) a[i] = a[i] * c;                                       // plain multiply
        if (OP == 7) u[i] = __builtin_amdgcn_ubfe(u[i], 3u, 9u) + u[i];     // bit-field extract + add
      }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int OP>
void run(const char* name, int per_iter, float* d) {
  const int reps = 1000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, 1.0001f, 12345u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, 1.0001f, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double groups = (double)reps * 16 * 8 * 4;   // source-level operations per SIMD (4 waves)
  printf("%-34s %.3f ms  %.2f cycles per source operation per SIMD @2.4GHz (%d instruction(s) each: %.2f per instruction)\n", name, ms,
         ms * 1e-3 * 2.4e9 / groups, per_iter, ms * 1e-3 * 2.4e9 / groups / per_iter);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_fma_f32", 1, d);
  run<6>("v_mul_f32", 1, d);
  run<1>("v_pk_fma_f32", 1, d);
  run<2>("v_add_u32 + v_and_b32", 2, d);
  run<7>("v_bfe_u32 + v_add_u32", 2, d);
  run<3>("v_cmp + v_cndmask + adds", 4, d);
  run<4>("v_mul_lo_u32 + v_add_u32", 2, d);
  run<5>("v_exp_f32 + v_mul_f32", 2, d);
  return 0;
}
