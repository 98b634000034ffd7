// Every random number of one training iteration in ONE launch (Philox4x32-10, counter based: no state per lane).
//
// Replaces, on the device-RNG path (`--rng device`, what bench.py measures), the draws the reference makes on the
// host generators (/root/reference/mggan):
//   utils.py:18-25      get_gan_labels     two uniforms per call             -> `labels[n_labels]`  u ~ U[0,1)
//   utils.py:152-165    get_global_noise   one N(0,1)^Z vector PER SCENE,    -> `noise[n_sets][b][Z]`, already repeated
//                       repeated for the scene's pedestrians                    for the pedestrians of each scene
//   standard.py:217-225 Categorical.sample uniforms for the inverse-CDF      -> `unif[n_unif]`      u ~ U[0,1)
// Same distributions, not the same stream as torch's generators (seed-comparable runs use --rng host).
// Counter layout: (index lo, index hi, stream id, iteration); key = seed.  `state[0]` = seed, `state[1]` = iteration
// counter: every workgroup reads it, the LAST one to finish (ticket) advances it, so a captured HIP graph draws fresh
// numbers at every replay without any host involvement.
#include "common.h"
#include "../../include/mggan_hip.h"

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }          // [0,1)
__device__ __forceinline__ float u01_open0(uint32_t x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-8f; }  // (0,1]

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float r = sqrtf(-2.0f * __logf(u01_open0(a)));
  float s, c;
  __sincosf(6.283185307179586f * u01(b), &s, &c);
  n0 = r * c;
  n1 = r * s;
}

__global__ __launch_bounds__(256) void draw_iteration_kernel(long long* state, unsigned int* ticket, int n_labels,
                                                             float* labels, int n_sets, int b, int Z,
                                                             const int* __restrict__ ped_scene, float* noise,
                                                             long n_unif, float* unif) {
  const uint2 key = make_uint2((uint32_t)state[0], (uint32_t)((unsigned long long)state[0] >> 32));
  const uint32_t iter_lo = (uint32_t)state[1], iter_hi = (uint32_t)((unsigned long long)state[1] >> 32);
  const int zq = (Z + 3) / 4;
  const long nA = (n_labels + 3) / 4, nB = (long)n_sets * b * zq, nC = (n_unif + 3) / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nA) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)i, iter_hi, 0u, iter_lo), key);
    const uint32_t v[4] = {r.x, r.y, r.z, r.w};
    for (int q = 0; q < 4; ++q)
      if (4 * i + q < n_labels) labels[4 * i + q] = u01(v[q]);
  } else if (i < nA + nB) {
    const long j = i - nA;
    const int q = (int)(j % zq);
    const long sp = j / zq;
    const int ped = (int)(sp % b), set = (int)(sp / b);
    // one vector per (sample set, SCENE): the counter is keyed by the scene, every pedestrian of it draws the same
    const unsigned long long idx = ((unsigned long long)set << 32) | (unsigned long long)(uint32_t)(ped_scene[ped] * zq + q);
    const uint4 r = philox4x32_10(make_uint4((uint32_t)idx, (uint32_t)(idx >> 32) ^ (iter_hi << 16), 1u, iter_lo), key);
    float n[4];
    box_muller(r.x, r.y, n[0], n[1]);
    box_muller(r.z, r.w, n[2], n[3]);
    float* o = noise + ((size_t)set * b + ped) * Z + 4 * q;
    for (int t = 0; t < 4; ++t)
      if (4 * q + t < Z) o[t] = n[t];
  } else if (i < nA + nB + nC) {
    const long j = i - nA - nB;
    const uint4 r = philox4x32_10(make_uint4((uint32_t)j, (uint32_t)((unsigned long long)j >> 32) ^ (iter_hi << 16), 2u, iter_lo), key);
    const uint32_t v[4] = {r.x, r.y, r.z, r.w};
    for (int q = 0; q < 4; ++q)
      if (4 * j + q < n_unif) unif[4 * j + q] = u01(v[q]);
  }
  // every lane of this workgroup has read state[] (its value feeds the stores above); the last workgroup to arrive
  // advances the iteration counter and re-arms the ticket for the next launch / graph replay
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(ticket, 1u);
    if (t == gridDim.x - 1) {
      state[1] += 1;
      *ticket = 0u;
    }
  }
}

extern "C" int mggan_draw_iteration(long long* state, unsigned int* ticket, int n_labels, float* labels, int n_sets, int b,
                                    int Z, const int* ped_scene, float* noise, long n_unif, float* unif,
                                    hipStream_t stream) {
  MG_CHECK_ARG(state && ticket, "draw_iteration: null state");
  MG_CHECK_ARG((n_labels == 0 || labels) && (n_unif == 0 || unif), "draw_iteration: null output");
  MG_CHECK_ARG(n_sets == 0 || b == 0 || (noise && ped_scene && Z > 0), "draw_iteration: noise needs ped_scene and Z > 0");
  const long n = (n_labels + 3) / 4 + (long)n_sets * b * ((Z + 3) / 4) + (n_unif + 3) / 4;
  MG_LAUNCH(draw_iteration_kernel, dim3(n > 0 ? cdiv(n, 256) : 1), dim3(256), 0, stream, state, ticket, n_labels,
                     labels, n_sets, b, Z, ped_scene, noise, n_unif, unif);
  MG_LAUNCH_CHECK("draw_iteration");
  return MGGAN_OK;
}
