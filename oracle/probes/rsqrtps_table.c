/* TEST INFRASTRUCTURE (oracle/probes): measure the x86 RSQRTPS approximation on the CPU that runs
 * Mesa llvmpipe here.  llvmpipe's sRGB8 encode (gallivm lp_build_linear_to_srgb: y = a*x^0.375 +
 * b*x^0.5 + c with x^0.5 = x*rsqrt(x), x^0.375 = rsqrt(rsqrt(x*x^0.5))) is built on this
 * instruction, so the bytes an sRGB8 render target stores are a function of its results.
 *
 * Finding (Intel Xeon of the build container): RSQRTPS(x) depends only on the exponent's parity
 * and the top 10 mantissa bits of x (2048 runs of 8192 consecutive floats over [1,4)), scaled by an
 * exact power of two, and is non-increasing.  This program verifies that for EVERY positive normal
 * float and prints the 2048-entry table as ((result bits >> 11) - 0x7e000) for x in [1,4)
 * (the results carry 12 significant bits).
 *
 *   gcc -O2 -o rsqrtps_table rsqrtps_table.c && ./rsqrtps_table > table.txt
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <xmmintrin.h>

static uint32_t rsq_bits(uint32_t b) {
  float x, r;
  memcpy(&x, &b, 4);
  r = _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(x)));
  uint32_t rb;
  memcpy(&rb, &r, 4);
  return rb;
}

int main(void) {
  static uint32_t tab[2048];
  /* idx = (bits >> 13) & 0x7ff: bit 10 = exponent LSB ([1,2): 1, [2,4): 0), bits 9..0 = mantissa top */
  for (uint32_t b = 0x3f800000u; b < 0x40800000u; b += 8192u) tab[(b >> 13) & 0x7ffu] = rsq_bits(b);
  long bad = 0;
  for (uint64_t b64 = 0x00800000u; b64 < 0x7f800000u; ++b64) {
    const uint32_t b = (uint32_t)b64;
    const int E = (int)(b >> 23);
    const int k = (E & 1) ? (E - 127) / 2 : (E - 128) / 2; /* exact: numerators are even */
    const uint32_t want = tab[(b >> 13) & 0x7ffu] - ((uint32_t)k << 23);
    if (rsq_bits(b) != want) ++bad;
  }
  int mono = 1;
  /* order of increasing x over [1,4): idx 1024..2047 then 0..1023 */
  uint32_t prev = 0xffffffffu;
  for (int i = 0; i < 2048; ++i) {
    const uint32_t v = tab[(i + 1024) & 2047];
    if (v > prev) mono = 0;
    prev = v;
  }
  fprintf(stderr, "mismatches over all positive normal floats: %ld; non-increasing: %d\n", bad, mono);
  for (int i = 0; i < 2048; ++i) {
    if ((tab[i] & 0x7ffu) != 0 || (tab[i] >> 11) < 0x7e000u || (tab[i] >> 11) - 0x7e000u > 0xffffu) {
      fprintf(stderr, "entry %d does not fit the 16-bit form\n", i);
      return 1;
    }
    printf("%u%s", (tab[i] >> 11) - 0x7e000u, (i % 16 == 15) ? ",\n" : ", ");
  }
  return bad != 0 || !mono;
}
