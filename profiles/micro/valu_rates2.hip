// Issue cost of single VALU instructions on gfx950, pinned with inline assembly (cycles per wave64 instruction per SIMD at 16 waves
// per CU, eight independent chains per wave).  Build / run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rates2 profiles/micro/valu_rates2.hip && /tmp/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAIN8(STR)                                                                 \
  asm volatile(STR : "+v"(u0) : "v"(m), "s"(sc)); asm volatile(STR : "+v"(u1) : "v"(m), "s"(sc)); \
  asm volatile(STR : "+v"(u2) : "v"(m), "s"(sc)); asm volatile(STR : "+v"(u3) : "v"(m), "s"(sc)); \
  asm volatile(STR : "+v"(u4) : "v"(m), "s"(sc)); asm volatile(STR : "+v"(u5) : "v"(m), "s"(sc)); \
  asm volatile(STR : "+v"(u6) : "v"(m), "s"(sc)); asm volatile(STR : "+v"(u7) : "v"(m), "s"(sc));

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int reps, uint32_t m, uint32_t sc) {
  uint32_t u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 9, u5 = u0 * 11, u6 = u0 * 13, u7 = u0 * 15;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (OP == 0) { CHAIN8("v_and_b32 %0, %0, %1") }
      if (OP == 1) { CHAIN8("v_add_u32 %0, %0, %1") }
      if (OP == 2) { CHAIN8("v_lshrrev_b32 %0, 3, %0") }
      if (OP == 3) { CHAIN8("v_bfe_u32 %0, %0, 3, 9") }
      if (OP == 4) { CHAIN8("v_lshl_add_u32 %0, %0, 2, %1") }
      if (OP == 5) { CHAIN8("v_and_or_b32 %0, %0, %1, %2") }
      if (OP == 6) { CHAIN8("v_cndmask_b32 %0, %0, %1, vcc") }
      if (OP == 7) { CHAIN8("v_lshlrev_b32_sdwa %0, 2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") }
      if (OP == 8) { CHAIN8("v_fma_f32 %0, %0, %1, %1") }
      if (OP == 9) { CHAIN8("v_fma_f32 %0, %0, %2, %1") }
      if (OP == 10) { CHAIN8("v_med3_f32 %0, %0, %1, 1.0") }
      if (OP == 11) { CHAIN8("v_cvt_f32_ubyte1 %0, %0") }
      if (OP == 12) { CHAIN8("v_mov_b32 %0, %1") }
      if (OP == 13) { CHAIN8("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf") }
      if (OP == 14) { CHAIN8("v_rndne_f32 %0, %0") }
      if (OP == 15) { CHAIN8("v_max_u32 %0, %0, %1") }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7;
}

template <int OP>
void run(const char* name, uint32_t* d) {
  const int reps = 1000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, 0x7ffu, 0x3f800000u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, 0x7ffu, 0x3f800000u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms  %.2f cycles per instruction per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)reps * 16 * 8 * 4));
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<8>("v_fma_f32 (vgpr operands)", d);
  run<9>("v_fma_f32 (sgpr operand)", d);
  run<0>("v_and_b32", d);
  run<1>("v_add_u32", d);
  run<15>("v_max_u32", d);
  run<2>("v_lshrrev_b32", d);
  run<3>("v_bfe_u32", d);
  run<4>("v_lshl_add_u32", d);
  run<5>("v_and_or_b32", d);
  run<6>("v_cndmask_b32", d);
  run<7>("v_lshlrev_b32_sdwa", d);
  run<10>("v_med3_f32", d);
  run<11>("v_cvt_f32_ubyte1", d);
  run<14>("v_rndne_f32", d);
  run<12>("v_mov_b32", d);
  run<13>("v_mov_b32_dpp wave_shl:1", d);
  return 0;
}
