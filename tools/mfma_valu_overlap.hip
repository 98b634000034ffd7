// Do two waves of one SIMD overlap one's matrix products with the other's vector arithmetic?  A loop of NM independent
// v_mfma_f32_16x16x4_f32 and NV dependent-free v_fma_f32 per iteration, one workgroup per CU, 1 / 2 / 4 waves per SIMD:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/mfma_valu_overlap.hip && /tmp/mvo
// prints shader cycles per iteration and wave (s_memtime) -- DESIGN.md section 5, round 4 (social rows).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], b, a);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);  // the slowest wave of the launch
}
template <int NM, int NV>
void run(float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int waves : {4, 8, 16}) {
    hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL((k<NM, NV>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);  // warm-up
    hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, NV>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %2d MFMA + %3d FMA per iteration, %d wave(s)/SIMD: slowest wave %7.1f cycles per iteration = %6.1f per wave-iteration of the SIMD; "
           "launch %.1f us = %6.1f cycles at 2.4 GHz per wave-iteration of the SIMD\n", NM, NV,
           waves / 4, (double)c / iters, (double)c / iters / (waves / 4), ms * 1e3, ms * 1e-3 * 2.4e9 / iters / (waves / 4));
  }
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  run<8, 0>(out, cyc);
  run<0, 64>(out, cyc);
  run<8, 64>(out, cyc);
  run<8, 128>(out, cyc);
  run<2, 64>(out, cyc);
  return 0;
}
