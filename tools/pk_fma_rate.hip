// What does a packed-f32 FMA cost on gfx950?  Issue rate of v_pk_fma_f32 (two FMAs per lane) against v_fma_f32, 24
// independent accumulators per lane, operands in registers:   hipcc --offload-arch=gfx950 -O3 -o /tmp/r tools/pk_fma_rate.hip && /tmp/r
// MODE 0: v_fma_f32 | 1: v_pk_fma_f32, three distinct register pairs | 2: v_pk_fma_f32 with the multiplier broadcast from one
// register (op_sel_hi) | 3: v_pk_fma_f32 with one multiplicand shared by every instruction of the block (the Gram walker's shape).
// Result (MI355X, round 5): see the header of image_gram_ac_kernel in csrc/cnn2.hip and DESIGN.md section 5.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x2 acc[24], w[24];
  for (int j = 0; j < 24; ++j) {
    acc[j] = f32x2{0.f, 0.f};
    w[j] = f32x2{1.f + j + threadIdx.x, 2.f + j};
  }
  f32x2 a = f32x2{1.f + (threadIdx.x & 3), 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(a.x), "v"(w[j].x));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(w[(j + 7) % 24]), "v"(w[j]));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "v"(a), "v"(w[j]));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(w[j]));
    }
  }
  f32x2 s = acc[0];
  for (int j = 1; j < 24; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
template <int MODE> void run(const char* name, double fma_per_lane) {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<2048, 256>>>(out, 10);
  hipEventRecord(e0); k<MODE><<<2048, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = 2048.0 * 4 * iters * 24;  // wave-instructions; 1,024 SIMDs
  printf("%-52s %.3f ms  %.1f TFLOP/s  %.2f ns per wave-instruction per SIMD\n", name, ms, insts * 64 * fma_per_lane * 2 / ms / 1e9,
         ms * 1e6 / (insts / 1024.0));
  hipFree(out);
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32, three distinct pairs", 2);
  run<2>("v_pk_fma_f32, broadcast multiplier (op_sel_hi)", 2); run<3>("v_pk_fma_f32, one shared multiplicand pair", 2);
  return 0;
}
