// store-pattern microbenchmark: the rollout forward's saved-state stores (tile-blocked records) with `spin` dependent FMAs per
// step in front of them, against the same loop without the stores:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_bw tools/store_bw.hip && /tmp/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int LAYOUT, bool NT, bool ST = true>
__global__ __launch_bounds__(256) void k(float* Gt, float* Cs, int ntiles, int T, int spin) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int H = 32;
  for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += gridDim.x * 4) {
    float acc = tile;
    for (int t = 0; t < T; ++t) {
      for (int s = 0; s < spin; ++s) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);  // stand-in for the step's arithmetic
      const size_t rec = LAYOUT == 0 ? (size_t)tile * T + t : (size_t)t * ntiles + tile;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4 g = {acc, acc, acc, acc};
        f32x4* pg = reinterpret_cast<f32x4*>(Gt + ((rec * H + 4 * u + fk) * 16 + fi) * 4);
        f32x2* pc = reinterpret_cast<f32x2*>(Cs + ((rec * H + 4 * u + fk) * 16 + fi) * 2);
        if (!ST) { if (acc == 12345.678f) *pg = g; } else if (NT) { __builtin_nontemporal_store(g, pg); __builtin_nontemporal_store(f32x2{acc, acc}, pc); }
        else { *pg = g; *pc = f32x2{acc, acc}; }
      }
    }
  }
}
int main() {
  const int rows = 163840, ntiles = rows / 16, T = 12;
  float *Gt, *Cs;
  hipMalloc(&Gt, (size_t)ntiles * T * 32 * 64 * 4);
  hipMalloc(&Cs, (size_t)ntiles * T * 32 * 32 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = (double)ntiles * T * 32 * 96 * 4 / 1e9;
  for (int spin : {0, 100, 300, 600})
  for (int grid : {640, 2560})
  for (int v : {0, 4}) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL((k<0, false>), dim3(grid), dim3(256), 0, 0, Gt, Cs, ntiles, T, spin);
      if (v == 1) hipLaunchKernelGGL((k<0, true>), dim3(grid), dim3(256), 0, 0, Gt, Cs, ntiles, T, spin);
      if (v == 2) hipLaunchKernelGGL((k<1, false>), dim3(grid), dim3(256), 0, 0, Gt, Cs, ntiles, T, spin);
      if (v == 3) hipLaunchKernelGGL((k<1, true>), dim3(grid), dim3(256), 0, 0, Gt, Cs, ntiles, T, spin);
      if (v == 4) hipLaunchKernelGGL((k<0, false, false>), dim3(grid), dim3(256), 0, 0, Gt, Cs, ntiles, T, spin);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("spin %5d grid %4d layout %s %s: %.1f us  %.2f TB/s\n", spin, grid, v == 4 ? "NO STORES" : "[tile][t]", "", ms * 1e3, gb / ms);
  }
  return 0;
}
