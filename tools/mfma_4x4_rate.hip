// Could the C = 8 kernels of the discriminator's scene CNN fill their matrix instructions with v_mfma_f32_4x4x1_16B_f32 (16
// independent 4x4 blocks: no idle columns when only 8 channels exist) instead of half-empty 16x16x4 tiles?  Rate of both
// instructions with their A operand in registers and with ONE ds_read_b32 per instruction (what a convolution's patch operand
// costs without sharing between lanes):   hipcc --offload-arch=gfx950 -O3 -o /tmp/r tools/mfma_4x4_rate.hip && /tmp/r
// MI355X, round 5:  4x4x1 registers 121 TFLOP/s | 4x4x1 + LDS read 52 | 16x16x4 registers 134 | 16x16x4 + LDS read 109.
// conv1_pool_kernel<8> delivers 41 useful TFLOP/s today (0.52 MFMA-busy, half of every tile idle): the 4x4x1 form would be
// LDS-bound at 52 before its epilogue -- a quarter faster at best, not the 2x the idle columns suggest.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: 4x4x1 regs only, 1: 4x4x1 + one ds_read_b32 per product, 2: 16x16x4 regs only, 3: 16x16x4 + ds_read per product
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.f + (i & 7);
  __syncthreads();
  f32x4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float b = 1.f + (threadIdx.x & 3);
  int off = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a;
      if (MODE & 1) a = lds[(off + 64 * j + 37 * it) & 4095]; else a = b + j;
      if (MODE < 2) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
  }
  f32x4 s = acc[0];
  for (int j = 1; j < 8; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int MODE> void run(const char* name, double macs_per_inst) {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<1024, 256>>>(out, 10);
  hipEventRecord(e0); k<MODE><<<1024, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = 1024.0 * 4 * iters * 8;  // wave-instructions
  printf("%-34s %.3f ms  %.2f TFLOP/s  (%.1f ns per wave-instruction per SIMD-slot)\n", name, ms, insts * macs_per_inst * 2 / ms / 1e9, ms * 1e6 / (insts / 1024.0));
  hipFree(out);
}
int main() {
  run<0>("4x4x1 registers", 256); run<1>("4x4x1 + ds_read_b32 each", 256);
  run<2>("16x16x4 registers", 1024); run<3>("16x16x4 + ds_read_b32 each", 1024);
  return 0;
}
