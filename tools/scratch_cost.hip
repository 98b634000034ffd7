// What does a kernel pay for touching scratch (private memory: spilled registers, dynamically indexed private arrays)?
// Two kernels that differ only in one 64-byte private array, 200 back-to-back launches each:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sc tools/scratch_cost.hip && /tmp/sc
// the social-attention backward kernel with spills spent its first ~4-5 us waiting.
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool SCRATCH>
__global__ __launch_bounds__(256) void k(float* out, const int* idx, int n) {
  float v = threadIdx.x;
  if (SCRATCH) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = v + i;
    const int j = idx[threadIdx.x & 15];  // unknown at compile time: the array lives in scratch
    a[j & 15] += 1.f;
    v = a[(j + 3) & 15] + a[(j + 7) & 15];
  } else {
    const int j = idx[threadIdx.x & 15];
    v = v + (float)((j + 3) & 15) + (float)((j + 7) & 15);
  }
  for (int i = 0; i < n; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
  float* out; int* idx;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&idx, 64);
  int h[16]; for (int i = 0; i < 16; ++i) h[i] = (i * 5) & 15;
  hipMemcpy(idx, h, 64, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {64, 256, 1024})
    for (int n : {0, 2000})
      for (int sc = 0; sc < 2; ++sc) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0, 0);
          for (int i = 0; i < 200; ++i) {
            if (sc) hipLaunchKernelGGL((k<true>), dim3(grid), dim3(256), 0, 0, out, idx, n);
            else hipLaunchKernelGGL((k<false>), dim3(grid), dim3(256), 0, 0, out, idx, n);
          }
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        printf("grid %4d, %4d FMAs per thread, %s: %.2f us per launch\n", grid, n, sc ? "scratch   " : "no scratch", best * 1e3 / 200);
      }
  return 0;
}
