// Issue cost table, part 3: more VALU instructions pinned with inline assembly (cycles per wave64 instruction per SIMD at 16 waves
// per CU, eight independent chains per wave).  Two-instruction rows report the cost of the PAIR.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rates3 profiles/micro/valu_rates3.hip && /tmp/valu_rates3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ONE(STR, R) asm volatile(STR : "+v"(R) : "v"(m), "s"(sc), "v"(m2) : "vcc", "s20", "s21");
#define CHAIN8(STR) ONE(STR, u0) ONE(STR, u1) ONE(STR, u2) ONE(STR, u3) ONE(STR, u4) ONE(STR, u5) ONE(STR, u6) ONE(STR, u7)
#define ONE2(STR, R) asm volatile(STR : "+v"(R) : "v"(mm), "s"(sc) : "vcc");
#define CHAIN4x2(STR) ONE2(STR, w0) ONE2(STR, w1) ONE2(STR, w2) ONE2(STR, w3)

typedef float __attribute__((ext_vector_type(2))) f2;

template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int reps, uint32_t m, uint32_t sc, uint32_t m2) {
  uint32_t u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 9, u5 = u0 * 11, u6 = u0 * 13, u7 = u0 * 15;
  f2 w0 = {1.f + threadIdx.x, 2.f}, w1 = w0 * 3.f, w2 = w0 * 5.f, w3 = w0 * 7.f, mm = {1.0000001f, 0.9999999f};
  m += threadIdx.x >> 9; m2 += threadIdx.x >> 8;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (OP == 0) { CHAIN8("v_fmac_f32 %0, %1, %3") }
      if (OP == 1) { CHAIN8("v_mul_f32 %0, %0, %1") }
      if (OP == 2) { CHAIN8("v_add_f32 %0, %0, %1") }
      if (OP == 3) { CHAIN8("v_fma_f32 %0, %0, %1, %3") }
      if (OP == 4) { CHAIN8("v_fma_f32 %0, %1, %3, %0") }
      if (OP == 5) { CHAIN4x2("v_pk_fma_f32 %0, %0, %1, %1") CHAIN4x2("v_pk_fma_f32 %0, %0, %1, %1") }
      if (OP == 6) { CHAIN4x2("v_pk_mul_f32 %0, %0, %1") CHAIN4x2("v_pk_mul_f32 %0, %0, %1") }
      if (OP == 7) { CHAIN4x2("v_pk_add_f32 %0, %0, %1") CHAIN4x2("v_pk_add_f32 %0, %0, %1") }
      if (OP == 8) { CHAIN8("v_max_f32 %0, %0, %1") }
      if (OP == 9) { CHAIN8("v_min_u32 %0, %0, %1") }
      if (OP == 10) { CHAIN8("v_cvt_u32_f32 %0, %0") }
      if (OP == 11) { CHAIN8("v_cvt_f32_u32 %0, %0") }
      if (OP == 12) { CHAIN8("v_sub_u32 %0, %0, %1") }
      if (OP == 13) { CHAIN8("v_or_b32 %0, %0, %1") }
      if (OP == 14) { CHAIN8("v_xor_b32 %0, %0, %1") }
      if (OP == 15) { CHAIN8("v_lshlrev_b32 %0, 3, %0") }
      if (OP == 16) { CHAIN8("v_mul_u32_u24 %0, %0, %1") }
      if (OP == 17) { CHAIN8("v_mad_u32_u24 %0, %0, %1, %3") }
      if (OP == 18) { CHAIN8("v_add3_u32 %0, %0, %1, %3") }
      if (OP == 19) { CHAIN8("v_perm_b32 %0, %0, %1, %3") }
      if (OP == 20) { CHAIN8("v_alignbit_b32 %0, %0, %1, 8") }
      if (OP == 21) { CHAIN8("v_exp_f32 %0, %0") }
      if (OP == 22) { CHAIN8("v_log_f32 %0, %0") }
      if (OP == 23) { CHAIN8("v_rcp_f32 %0, %0") }
      if (OP == 24) { CHAIN8("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") }
      if (OP == 25) { CHAIN8("v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]") }
      if (OP == 26) { CHAIN8("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") }
      if (OP == 27) { CHAIN8("v_cmp_lt_u32 vcc, %0, %1") }
      if (OP == 28) { CHAIN8("v_mul_lo_u32 %0, %0, %1") }
      if (OP == 29) { CHAIN8("v_mul_hi_u32 %0, %0, %1") }
      if (OP == 30) { CHAIN8("v_fract_f32 %0, %0") }
      if (OP == 31) { CHAIN8("v_floor_f32 %0, %0") }
      if (OP == 32) { CHAIN8("v_sqrt_f32 %0, %0") }
      if (OP == 33) { CHAIN8("v_and_b32 %0, 0xff00ff, %0") }
      if (OP == 34) { CHAIN8("v_add_u32 %0, %2, %0") }
      if (OP == 35) { CHAIN8("v_mul_f32 %0, %2, %0") }
      if (OP == 36) { CHAIN8("v_fmac_f32 %0, %2, %1") }
      if (OP == 37) { CHAIN8("v_fmaak_f32 %0, %0, %1, 0x3f000000") }
      if (OP == 38) { CHAIN8("v_cvt_pk_u8_f32 %0, %1, 1, %0") }
      if (OP == 39) { CHAIN8("v_max3_f32 %0, %0, %1, %3") }
      if (OP == 40) { CHAIN8("v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %0") }
      if (OP == 41) { CHAIN8("v_lshrrev_b32 %0, %1, %0") }
      if (OP == 42) { CHAIN8("v_subrev_f32 %0, %1, %0") }
      if (OP == 43) { CHAIN8("v_cvt_f32_ubyte0 %0, %0") }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + (uint32_t)(w0.x + w1.y + w2.x + w3.y);
}

template <int OP>
void run(const char* name, uint32_t* d) {
  const int reps = 500;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, 0x3f800001u, 0x3f800000u, 0x3f7ffff0u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, reps, 0x3f800001u, 0x3f800000u, 0x3f7ffff0u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms  %.2f cycles per row entry per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)reps * 16 * 8 * 4));
  fflush(stdout);
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_fmac_f32 d, v, v", d);
  run<36>("v_fmac_f32 d, s, v", d);
  run<37>("v_fmaak_f32 d, d, v, literal", d);
  run<1>("v_mul_f32", d);
  run<35>("v_mul_f32 d, s, d", d);
  run<2>("v_add_f32", d);
  run<42>("v_subrev_f32", d);
  run<3>("v_fma_f32 d, d, v, v2 (VOP3, 3 vgprs)", d);
  run<4>("v_fma_f32 d, v, v2, d (fmac shape, VOP3)", d);
  run<5>("v_pk_fma_f32", d);
  run<6>("v_pk_mul_f32", d);
  run<7>("v_pk_add_f32", d);
  run<8>("v_max_f32", d);
  run<39>("v_max3_f32", d);
  run<9>("v_min_u32", d);
  run<10>("v_cvt_u32_f32", d);
  run<11>("v_cvt_f32_u32", d);
  run<43>("v_cvt_f32_ubyte0", d);
  run<38>("v_cvt_pk_u8_f32", d);
  run<30>("v_fract_f32", d);
  run<31>("v_floor_f32", d);
  run<12>("v_sub_u32", d);
  run<34>("v_add_u32 d, s, d", d);
  run<13>("v_or_b32", d);
  run<14>("v_xor_b32", d);
  run<33>("v_and_b32 d, literal, d", d);
  run<15>("v_lshlrev_b32 d, 3, d", d);
  run<41>("v_lshrrev_b32 d, v, d", d);
  run<16>("v_mul_u32_u24", d);
  run<17>("v_mad_u32_u24", d);
  run<18>("v_add3_u32", d);
  run<19>("v_perm_b32", d);
  run<20>("v_alignbit_b32", d);
  run<28>("v_mul_lo_u32", d);
  run<29>("v_mul_hi_u32", d);
  run<21>("v_exp_f32", d);
  run<22>("v_log_f32", d);
  run<23>("v_rcp_f32", d);
  run<32>("v_sqrt_f32", d);
  run<27>("v_cmp_lt_u32 vcc", d);
  run<24>("PAIR v_cmp_lt_u32 vcc + v_cndmask vcc", d);
  run<25>("PAIR v_cmp_lt_u32 s[20:21] + v_cndmask (VOP3)", d);
  run<26>("PAIR v_cmp_lt_f32 vcc + v_cndmask vcc", d);
  run<40>("PAIR v_readfirstlane + v_add_u32 d, s, d", d);
  return 0;
}
