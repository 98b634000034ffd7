// Shared host/device helpers for libmggan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MGGAN_OK 0
#define MGGAN_ERR_ARG (-1)
#define MGGAN_ERR_LAUNCH (-2)
#define MGGAN_ERR_WORKSPACE (-3)

void mggan_set_error(const char* fmt, ...);

#define MG_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      mggan_set_error(__VA_ARGS__);        \
      return MGGAN_ERR_ARG;                \
    }                                      \
  } while (0)

#define MG_LAUNCH_CHECK(name)                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      mggan_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return MGGAN_ERR_LAUNCH;                                             \
    }                                                                      \
  } while (0)

#define MG_CHECK_HIP(call, name)                                           \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      mggan_set_error("%s: %s", name, hipGetErrorString(e_));              \
      return MGGAN_ERR_LAUNCH;                                             \
    }                                                                      \
  } while (0)

// Every kernel launch of the library goes through MG_LAUNCH: while the launch log is on (mggan_launch_log, a
// measurement aid: bench.py / tools name the HIP kernel an entry ran by its SYMBOL, the name rocprofv3 reports) the host
// function pointer and the launch size are noted; otherwise it is hipLaunchKernelGGL and one load of a flag.
extern int g_mggan_launch_log;
void mggan_note_launch(const void* host_fn, dim3 grid, dim3 block);
#define MG_LAUNCH(kernel, grid, block, ...)                                                   \
  do {                                                                                        \
    if (g_mggan_launch_log) mggan_note_launch((const void*)(kernel), (grid), (block));        \
    hipLaunchKernelGGL(kernel, (grid), (block), __VA_ARGS__);                                 \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// activation codes shared with the Python side
#define ACT_NONE 0
#define ACT_LEAKY 1    // slope 0 == ReLU
#define ACT_SIGMOID 2
#define ACT_SIGMOID_EPS 3  // sigmoid(x)*(1-2e-7)+1e-7 : D output, discriminators.py:83-84,203-204
#define MG_D_EPS 1e-7f

// v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division sequence
__device__ __forceinline__ float mg_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float mg_sigmoid(float x) { return mg_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float mg_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); accurate to ~2 ulp with __expf, saturates cleanly
  float e = __expf(2.0f * x);
  return fmaf(-2.0f, mg_rcp(e + 1.0f), 1.0f);
}
__device__ __forceinline__ float mg_act(float x, int act, float slope) {
  if (act == ACT_LEAKY) return x > 0.f ? x : x * slope;
  if (act == ACT_SIGMOID) return mg_sigmoid(x);
  if (act == ACT_SIGMOID_EPS) return mg_sigmoid(x) * (1.f - 2.f * MG_D_EPS) + MG_D_EPS;
  return x;
}
// derivative expressed through the OUTPUT y of the activation
__device__ __forceinline__ float mg_act_grad_from_out(float y, int act, float slope) {
  if (act == ACT_LEAKY) return y > 0.f ? 1.f : slope;
  if (act == ACT_SIGMOID) return y * (1.f - y);
  if (act == ACT_SIGMOID_EPS) {
    const float s = (y - MG_D_EPS) / (1.f - 2.f * MG_D_EPS);
    return (1.f - 2.f * MG_D_EPS) * s * (1.f - s);
  }
  return 1.f;
}

// Padded batches (the trainer's shape buckets, mggan/abstract_train.py): a batch is padded to its bucket's pedestrian
// count with inert "phantom" pedestrians at the end, and the kernels that mix rows (BatchNorm statistics, loss means,
// generator counts) are told how many leading rows are real through a 16-byte record in DEVICE memory -- a replayed graph
// reads it afresh, the launch geometry stays that of the bucket:  {int n_real; int s_real; float b_pad / n_real; pad}.
// A null pointer means "every row is real".
__device__ __forceinline__ int mg_real_rows(const int* dims, int n) {
  if (!dims) return n;
  const int r = dims[0];
  return r < n ? (r < 0 ? 0 : r) : n;
}
__device__ __forceinline__ int mg_real_scenes(const int* dims, int n) {
  if (!dims) return n;
  const int r = dims[1];
  return r < n ? (r < 0 ? 0 : r) : n;
}
// normalisers computed on the host from the padded count (1 / (c b_pad)) times this = 1 / (c n_real)
__device__ __forceinline__ float mg_pad_corr(const int* dims) { return dims ? __int_as_float(dims[2]) : 1.f; }
// image loops of the scene CNN kernels: only the real images, and the element count of the statistics shrinks with them
#define MG_REAL_IMAGES(B, dims)                       \
  const int B_padded_ = (B);                          \
  (B) = mg_real_rows((dims), (B));
#define MG_REAL_IMAGES_COUNT(B, dims, fin)            \
  MG_REAL_IMAGES(B, dims)                             \
  if ((B) != B_padded_ && B_padded_ > 0) (fin).count = (fin).count * (double)(B) / (double)B_padded_;

// The lane index from the execution-mask count (v_mbcnt) instead of the work-item id register: a value that can be had again
// anywhere at two instructions need not be kept (or spilled) across a loop that leaves no register free.  With the wave
// index in a scalar register (mg_wave) the pair replaces threadIdx.x in the tails of such kernels.
__device__ __forceinline__ int mg_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int mg_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global
// store (s_waitcnt vmcnt(0)), which costs a full HBM write round trip per barrier in kernels that stream
// their saved activations out inside a time loop.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sum / max over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane: two
// v_permlane{16,32}_swap + two VALU ops (gfx950) instead of two ds_bpermute round trips through the LDS crossbar.
// swap16(x, x) -> {[r0, r0, r2, r2], [r1, r1, r3, r3]}, swap32(y, y) -> {[lo, lo], [hi, hi]} (checked on hardware
// against __shfl_xor: bit-identical).
typedef unsigned mg_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float quarters_sum(float v) {
  const mg_u2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = __uint_as_float(s[0]) + __uint_as_float(s[1]);
  const mg_u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float quarters_max(float v) {
  const mg_u2 s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
  const mg_u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

// All-lanes sum / max over the 16 lanes of a DPP row (lanes l ^ 1, 2, 4, 8 ...): four row_ror DPP operations on the
// VALU instead of four ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float mg_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += mg_dpp<0x128>(v);  // row_ror:8
  v += mg_dpp<0x124>(v);  // row_ror:4
  v += mg_dpp<0x122>(v);
  v += mg_dpp<0x121>(v);
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, mg_dpp<0x128>(v));
  v = fmaxf(v, mg_dpp<0x124>(v));
  v = fmaxf(v, mg_dpp<0x122>(v));
  v = fmaxf(v, mg_dpp<0x121>(v));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
