// In-graph all-reduce between the GPUs of one node (scene-sharded training) -- a plain kernel, so the
// sharded iteration stays ONE HIP graph with its branch streams; RCCL calls would cut the capture at every exchange
// point (18 per iteration), and every message here is <= 360 KB, i.e. latency bound.
//
// One-shot exchange over peer-mapped memory (xGMI is point to point: every rank has a direct link to every other):
// each rank owns an UNCACHED arena that its peers map through hipIpc*.  For a vector of n elements, workgroup k of
// rank r (a) stores its chunk k into slot r of EVERY rank's arena, (b) system-scope fence, then stamps flag[r][k] of
// every arena with the collective's sequence number, (c) waits until all W flags of chunk k in its OWN arena carry that
// number, (d) adds the W slots in rank order -- every rank computes bit-identical sums, replicas cannot drift.
// Two buffers alternate by sequence parity: a rank can only start collective s+2 after it finished s+1, which needed
// every peer's flag of s+1, which a peer stamps only after it finished reading s.  Waits are bounded (wall clock,
// mggan_comm_set_timeout, 30 s by default): a lost peer sets the arena's error word and a host-mapped word
// (mggan_comm_host_error: readable without a device sync) instead of hanging the GPU, and the collective then leaves NaN
// (INT_MIN) behind -- never a sum over stale slots.
// No reference counterpart (the reference is single-process); replaces torch.distributed.all_reduce on this path.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "../../include/mggan_hip.h"

#include "comm_dev.h"

static long long g_timeout_ticks = (long long)(COMM_TIMEOUT_DEFAULT_S * COMM_TICKS_PER_S);
static unsigned* g_host_error = nullptr;
int comm_light() {
  static const int v = [] {
    const char* e = getenv("MGGAN_COMM_FENCES");
    return (e && e[0] == 'l') ? 1 : 0;
  }();
  return v;
}
long long comm_timeout_ticks() { return g_timeout_ticks; }
unsigned* comm_host_error() { return g_host_error; }

// Vector 1 (`data`, n elements of T) in chunks 0 .. nb1-1; an optional f64 TAIL (`data2`, n2 doubles, behind vector 1 in the
// slot at the next 256-byte boundary) in the chunks that follow: the per-rank f64 sums that must travel with a gradient
// buffer (the layer-1 BatchNorm adjoint sums and the raw conv1 weight-gradient sums of the sharded scene CNN) ride in the
// same collective instead of being one of their own.
template <typename T>
__global__ __launch_bounds__(256) void comm_allreduce_kernel(CommArgs a) {
  __shared__ int lost_s;
  CommHeader* hdr = (CommHeader*)a.arena[a.rank];
  const unsigned seq = __hip_atomic_load(&hdr->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int k = blockIdx.x, nb1 = (int)((a.n + COMM_CHUNK - 1) / COMM_CHUNK);
  if (k < nb1) {
    const long e0 = (long)k * COMM_CHUNK;
    comm_chunk<T>(a, (T*)a.data, e0, min((long)COMM_CHUNK, a.n - e0), 0, k, seq, hdr, &lost_s);
  } else {
    const long e0 = (long)(k - nb1) * COMM_CHUNK;
    const size_t off = ((size_t)a.n * sizeof(T) + 255) / 256 * 256;
    comm_chunk<double>(a, (double*)a.data2, e0, min((long)COMM_CHUNK, a.n2 - e0), off, k, seq, hdr, &lost_s);
  }
  // the last workgroup closes the collective
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(&hdr->done, 1u);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(&hdr->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&hdr->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

extern "C" {

size_t mggan_comm_arena_bytes(long max_elems) {
  const int max_blocks = cdiv(max_elems * 2, COMM_CHUNK);  // 4-byte elements fill a slot with twice as many
  return comm_data_off(max_blocks) + (size_t)2 * COMM_MAX_RANKS * max_elems * 8;
}

int mggan_comm_alloc(size_t bytes, void** out) {
  MG_CHECK_ARG(out && bytes > 0, "comm_alloc: bad arguments");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    mggan_set_error("comm_alloc: hipExtMallocWithFlags(%zu, uncached) failed: %s", bytes, hipGetErrorString(e));
    return MGGAN_ERR_LAUNCH;
  }
  e = hipMemset(p, 0, bytes);
  if (e != hipSuccess) {
    mggan_set_error("comm_alloc: hipMemset failed: %s", hipGetErrorString(e));
    return MGGAN_ERR_LAUNCH;
  }
  *out = p;
  return MGGAN_OK;
}

int mggan_comm_free(void* p) {
  if (p && hipFree(p) != hipSuccess) return MGGAN_ERR_LAUNCH;
  return MGGAN_OK;
}

/* handle: 64 bytes (hipIpcMemHandle_t) */
int mggan_comm_ipc_handle(void* p, void* handle) {
  MG_CHECK_ARG(p && handle, "comm_ipc_handle: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipError_t e = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle, p);
  if (e != hipSuccess) {
    mggan_set_error("comm_ipc_handle: hipIpcGetMemHandle failed: %s", hipGetErrorString(e));
    return MGGAN_ERR_LAUNCH;
  }
  return MGGAN_OK;
}

int mggan_comm_ipc_open(const void* handle, void** out) {
  MG_CHECK_ARG(handle && out, "comm_ipc_open: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    mggan_set_error("comm_ipc_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    return MGGAN_ERR_LAUNCH;
  }
  *out = p;
  return MGGAN_OK;
}

int mggan_comm_ipc_close(void* p) {
  if (p && hipIpcCloseMemHandle(p) != hipSuccess) return MGGAN_ERR_LAUNCH;
  return MGGAN_OK;
}

/* arenas: `world` pointers (arena of rank j mapped into this process); data: n elements reduced in place */
static int comm_allreduce_launch(void* const* arenas, int rank, int world, long max_elems, void* data, long n, int dtype,
                                 double* data2, long n2, hipStream_t stream) {
  MG_CHECK_ARG(arenas && (data || n == 0) && world >= 1 && world <= COMM_MAX_RANKS && rank >= 0 && rank < world,
               "comm_allreduce: bad ranks");
  MG_CHECK_ARG(dtype >= 0 && dtype <= 2, "comm_allreduce: dtype %d (0 f32, 1 f64, 2 i32)", dtype);
  MG_CHECK_ARG(n >= 0 && n2 >= 0 && (data2 || n2 == 0), "comm_allreduce: bad vector lengths");
  const size_t esz = dtype == 1 ? 8 : 4;
  const size_t off2 = ((size_t)n * esz + 255) / 256 * 256;
  MG_CHECK_ARG((n2 ? off2 + (size_t)n2 * 8 : (size_t)n * esz) <= (size_t)max_elems * 8,
               "comm_allreduce: %ld + %ld elements exceed the arena slot (%ld bytes)", n, n2, max_elems * 8);
  if (n == 0 && n2 == 0) return MGGAN_OK;
  CommArgs a;
  for (int j = 0; j < COMM_MAX_RANKS; ++j) a.arena[j] = j < world ? arenas[j] : nullptr;
  a.data = data; a.n = n; a.max_elems = max_elems; a.rank = rank; a.world = world;
  a.max_blocks = cdiv(max_elems * 2, COMM_CHUNK); a.dtype = dtype;
  a.data2 = data2; a.n2 = n2;
  a.timeout_ticks = g_timeout_ticks; a.host_error = g_host_error; a.light = comm_light();
  const int grid = cdiv(n, COMM_CHUNK) + cdiv(n2, COMM_CHUNK);
  MG_CHECK_ARG(grid <= a.max_blocks, "comm_allreduce: %d chunks exceed the arena's flags (%d)", grid, a.max_blocks);
  if (dtype == 0) MG_LAUNCH(comm_allreduce_kernel<float>, dim3(grid), dim3(256), 0, stream, a);
  else if (dtype == 1) MG_LAUNCH(comm_allreduce_kernel<double>, dim3(grid), dim3(256), 0, stream, a);
  else MG_LAUNCH(comm_allreduce_kernel<int>, dim3(grid), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("comm_allreduce");
  return MGGAN_OK;
}

int mggan_comm_allreduce(void* const* arenas, int rank, int world, long max_elems, void* data, long n, int dtype,
                         hipStream_t stream) {
  return comm_allreduce_launch(arenas, rank, world, max_elems, data, n, dtype, nullptr, 0, stream);
}

/* ... with an f64 tail (n2 doubles) summed in the same collective */
int mggan_comm_allreduce2(void* const* arenas, int rank, int world, long max_elems, void* data, long n, int dtype,
                          double* data2, long n2, hipStream_t stream) {
  return comm_allreduce_launch(arenas, rank, world, max_elems, data, n, dtype, data2, n2, stream);
}

/* A channel's CommArgs in DEVICE memory, for the kernels that fold a small exchange into their last workgroup
   (mggan_conv1_pool / mggan_conv2_fwd2 / mggan_scene_attention_bwd: `comm`); carries the wait bound and the host error
   word in force when it is created.  *out: device pointer, released with mggan_comm_channel_free. */
int mggan_comm_channel_create(void* const* arenas, int rank, int world, long max_elems, void** out) {
  MG_CHECK_ARG(arenas && out && world >= 1 && world <= COMM_MAX_RANKS && rank >= 0 && rank < world && max_elems > 0,
               "comm_channel_create: bad arguments");
  const CommArgs a = comm_make_args(arenas, rank, world, max_elems);
  void* p = nullptr;
  MG_CHECK_HIP(hipMalloc(&p, sizeof(CommArgs)), "comm_channel_create: hipMalloc");
  hipError_t e = hipMemcpy(p, &a, sizeof(CommArgs), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p);
    mggan_set_error("comm_channel_create: hipMemcpy failed: %s", hipGetErrorString(e));
    return MGGAN_ERR_LAUNCH;
  }
  *out = p;
  return MGGAN_OK;
}

int mggan_comm_channel_free(void* p) {
  if (p && hipFree(p) != hipSuccess) return MGGAN_ERR_LAUNCH;
  return MGGAN_OK;
}

/* bound of every wait inside a collective, in seconds (> 0); applies to launches and captures made afterwards */
int mggan_comm_set_timeout(double seconds) {
  MG_CHECK_ARG(seconds > 0.0 && seconds < 1e6, "comm_set_timeout: %g s", seconds);
  g_timeout_ticks = (long long)(seconds * COMM_TICKS_PER_S);
  return MGGAN_OK;
}

/* One host-mapped error word per process, set by any collective whose wait timed out: *out is a HOST pointer the
   caller may read at any time without synchronising the device (allocated on first use; reset = store 0). */
int mggan_comm_host_error(unsigned int** out) {
  MG_CHECK_ARG(out, "comm_host_error: null pointer");
  if (!g_host_error) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) {
      mggan_set_error("comm_host_error: hipHostMalloc failed: %s", hipGetErrorString(e));
      return MGGAN_ERR_LAUNCH;
    }
    memset(p, 0, 64);
    g_host_error = (unsigned*)p;
  }
  *out = g_host_error;
  return MGGAN_OK;
}

/* error word of an arena (host read; synchronises the device) */
int mggan_comm_error(const void* arena, unsigned int* out) {
  MG_CHECK_ARG(arena && out, "comm_error: null pointer");
  CommHeader h;
  if (hipMemcpy(&h, arena, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return MGGAN_ERR_LAUNCH;
  *out = h.error;
  return MGGAN_OK;
}

}  // extern "C"
