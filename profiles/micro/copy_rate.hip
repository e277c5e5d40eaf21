// The box's streaming ceiling for a plain copy kernel: 16 bytes per lane, variants of grid size and loads in flight per thread.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/copy_rate profiles/micro/copy_rate.hip && /tmp/copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) d[i + u * stride] = v[u];
  }
  for (; i < n; i += stride) d[i] = s[i];
}
template <int U>
void run(const uint4* a, uint4* b, size_t n, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(256), 0, 0, a, b, n);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(256), 0, 0, a, b, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("unroll %d blocks %5d: %.0f GB/s\n", U, blocks, 2.0 * n * 16 * 20 / (ms * 1e-3) / 1e9);
}
int main() {
  const size_t bytes = 1ull << 30, n = bytes / 16;
  uint4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 7, bytes);
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) { run<1>(a, b, n, blocks); run<2>(a, b, n, blocks); run<4>(a, b, n, blocks); }
  return 0;
}
