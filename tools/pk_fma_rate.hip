// What does a packed-f32 FMA cost on gfx950?  Issue rate of v_pk_fma_f32 (two FMAs per lane) against v_fma_f32, 24
// independent accumulators per lane, operands in registers:   hipcc --offload-arch=gfx950 -O3 -o /tmp/r tools/pk_fma_rate.hip && /tmp/r
// MODE 4-6: v_fmac_f32 with the multiplier in a scalar register / four dependent chains (the C = 8 convolution's shape).
// MODE 0: v_fma_f32 | 1: v_pk_fma_f32, three distinct register pairs | 2: v_pk_fma_f32 with the multiplier broadcast from one
// register (op_sel_hi) | 3: v_pk_fma_f32 with one multiplicand shared by every instruction of the block (the Gram walker's shape).
// Result (MI355X, round 5): v_fma_f32 1.23 ns per wave-instruction and SIMD (106 TFLOP/s); v_pk_fma_f32 2.1-2.2 ns (118-125):
// packing halves the instruction count, not the time; v_fmac_f32 with a scalar-register multiplier 1.1-1.2 ns, also with only four
// dependent chains from two waves per SIMD on (one wave: 2.5 ns).  DESIGN.md sections 3 and 5.
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
  const float sw = __builtin_amdgcn_readfirstlane(iters) * 1e-9f;  // a weight in a scalar register
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(a.x), "v"(w[j].x));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(w[(j + 7) % 24]), "v"(w[j]));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "v"(a), "v"(w[j]));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(w[j]));
      if (MODE == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j].x) : "s"(sw), "v"(w[j].x));
      if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j & 3].x) : "s"(sw), "v"(w[j].x));
      if (MODE == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j & 3].x) : "v"(a.x), "v"(w[j].x));
    }
  }
  f32x2 s = acc[0];
  for (int j = 1; j < 24; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
template <int MODE> void run(const char* name, double fma_per_lane, int wgs = 2048) {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<wgs, 256>>>(out, 10);
  hipEventRecord(e0); k<MODE><<<wgs, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)wgs * 4 * iters * 24;  // wave-instructions; 1,024 SIMDs
  printf("%-52s %.3f ms  %.1f TFLOP/s  %.2f ns per wave-instruction per SIMD\n", name, ms, insts * 64 * fma_per_lane * 2 / ms / 1e9,
         ms * 1e6 / (insts / 1024.0));
  hipFree(out);
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32, three distinct pairs", 2);
  run<2>("v_pk_fma_f32, broadcast multiplier (op_sel_hi)", 2); run<3>("v_pk_fma_f32, one shared multiplicand pair", 2);
  run<4>("v_fmac_f32, scalar-register multiplier, 24 chains", 1); run<5>("v_fmac_f32, scalar-register multiplier, 4 chains", 1);
  run<6>("v_fmac_f32, vector multiplier, 4 chains", 1);
  // one, two, four waves per SIMD: the latency of a dependent FMA shows when the other waves cannot cover it
  run<5>("  4 chains, scalar multiplier, 1 wave per SIMD", 1, 256); run<5>("  4 chains, scalar multiplier, 2 waves per SIMD", 1, 512);
  run<5>("  4 chains, scalar multiplier, 4 waves per SIMD", 1, 1024); run<4>("  24 chains, scalar multiplier, 1 wave per SIMD", 1, 256);
  run<0>("  24 chains, v_fma_f32, 1 wave per SIMD", 1, 256); run<0>("  24 chains, v_fma_f32, 2 waves per SIMD", 1, 512);
  return 0;
}
