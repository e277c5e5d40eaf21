// Issue cost table, part 4: packed FP32 instructions with an SGPR-pair source (the form the compiler emits for a wave-uniform
// weight: v_pk_mul_f32 v[..], s[..], v[..] / v_pk_fma_f32 ... op_sel_hi:[0,1,1]) against the all-VGPR form.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rates4 profiles/micro/valu_rates4.hip && /tmp/valu_rates4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float __attribute__((ext_vector_type(2))) f2;
#define ONE2(STR, R) asm volatile(STR : "+v"(R) : "v"(mm), "s"(sc), "v"(m1) : "vcc");
#define CHAIN8(STR) ONE2(STR, w0) ONE2(STR, w1) ONE2(STR, w2) ONE2(STR, w3) ONE2(STR, w4) ONE2(STR, w5) ONE2(STR, w6) ONE2(STR, w7)

template <int OP>
__global__ void __launch_bounds__(1024) k(float* out, int reps, uint64_t sc, float m1) {
  f2 w0 = {1.f + threadIdx.x, 2.f}, w1 = w0 * 3.f, w2 = w0 * 5.f, w3 = w0 * 7.f, w4 = w0 * 9.f, w5 = w0 * 11.f, w6 = w0 * 13.f, w7 = w0 * 15.f;
  f2 mm = {1.0000001f, 0.9999999f};
  m1 += (float)(threadIdx.x >> 9);
  mm.x += (float)(threadIdx.x >> 9);
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (OP == 0) { CHAIN8("v_pk_mul_f32 %0, %1, %0") }
      if (OP == 1) { CHAIN8("v_pk_mul_f32 %0, %2, %0") }
      if (OP == 2) { CHAIN8("v_pk_mul_f32 %0, %2, %0 op_sel_hi:[0,1]") }
      if (OP == 3) { CHAIN8("v_pk_fma_f32 %0, %1, %0, %1") }
      if (OP == 4) { CHAIN8("v_pk_fma_f32 %0, %2, %0, %1 op_sel_hi:[0,1,1]") }
      if (OP == 5) { CHAIN8("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]") }
      if (OP == 6) { CHAIN8("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]") }
      if (OP == 7) { CHAIN8("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]") }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = w0.x + w1.y + w2.x + w3.y + w4.x + w5.y + w6.x + w7.y;
}

template <int OP>
void run(const char* name, float* d) {
  const int reps = 500;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const uint64_t sc = 0x3f8000013f800001ull;
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, sc, 1.0000001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, sc, 1.0000001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %.3f ms  %.2f cycles per instruction per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)reps * 16 * 8 * 4));
  fflush(stdout);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_pk_mul_f32 d, v[2], d", d);
  run<6>("v_pk_mul_f32 d, v[2], d op_sel_hi:[0,1] (splat)", d);
  run<1>("v_pk_mul_f32 d, s[2], d", d);
  run<2>("v_pk_mul_f32 d, s[2], d op_sel_hi:[0,1] (splat)", d);
  run<3>("v_pk_fma_f32 d, v[2], d, v[2]", d);
  run<7>("v_pk_fma_f32 d, v[2], d, d op_sel_hi:[0,1,1]", d);
  run<4>("v_pk_fma_f32 d, s[2], d, v[2] op_sel_hi:[0,1,1]", d);
  run<5>("v_pk_add_f32 d, d, -v[2]", d);
  return 0;
}
