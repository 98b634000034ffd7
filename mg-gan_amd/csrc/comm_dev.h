// Device-side pieces of the peer-mapped all-reduce (csrc/comm.hip) that other kernels embed: the arena layout and a
// small-vector exchange done by ONE workgroup (the BatchNorm statistics of the sharded scene CNN are 33 doubles: the
// kernel that folds the partial rows also exchanges them and finalizes, csrc/cnn2.hip).
#pragma once
#include "common.h"

#define COMM_MAX_RANKS 8
#define COMM_CHUNK 1024                 // elements per workgroup (4 per lane)
#define COMM_TICKS_PER_S 100000000ll     // wall_clock64 ticks (100 MHz)
#define COMM_TIMEOUT_DEFAULT_S 30.0      // ranks drift apart for seconds in ordinary runs (per-rank data loading, validation,
                                         // checkpoints): the bound is for a LOST peer, not a late one (mggan_comm_set_timeout)

struct CommHeader {        // at the start of every arena (local use only)
  unsigned seq;            // collectives completed on this channel
  unsigned done;           // workgroups of the running collective that have finished
  unsigned error;          // set when a wait timed out
  unsigned pad;
};

struct CommArgs {
  void* arena[COMM_MAX_RANKS];  // arena of rank j for this channel, mapped into this process (arena[rank] = own)
  void* data;                   // vector to reduce in place
  long n;
  double* data2;                // optional f64 tail reduced in the same collective (csrc/comm.hip), n2 elements
  long n2;
  long max_elems;               // capacity of one slot in elements of the widest type (8 bytes)
  int rank, world, max_blocks, dtype;  // dtype 0: f32, 1: f64, 2: i32
  long long timeout_ticks;             // bound of every wait
  unsigned* host_error;                // host-mapped word, set (with the arena's) when a wait timed out; may be null
  int light;                           // MGGAN_COMM_FENCES=light: no system-scope fence / release / acquire around the flags (below)
};

// The arenas are UNCACHED (fine-grained) memory: a store to them is written through and a load bypasses the caches, and the
// workgroup barrier between the data stores and the flag store already waits for every wave's stores to be acknowledged
// (s_waitcnt vmcnt(0)).  The formal protocol (default, `light == 0`) still brackets the flags with a system-scope fence, a
// release store and acquire loads -- each a write-back / invalidate of the caches this kernel's OTHER traffic dirtied, 1-3 us
// apiece on the chains of the small shard.  `light` drops them: flags and data are ordered by the barrier and by the
// point-to-point link's write ordering alone.  Measured on one rank only; the default stays strict until
// the protocol has met a second device.
__device__ __forceinline__ void comm_publish_fence(const CommArgs& a) {
  if (!a.light) __threadfence_system();
}
__device__ __forceinline__ void comm_store_flag(const CommArgs& a, unsigned* pf, unsigned seq) {
  if (a.light) __hip_atomic_store(pf, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(pf, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

long long comm_timeout_ticks();  // csrc/comm.hip
unsigned* comm_host_error();
int comm_light();

// What a timed-out collective leaves behind: never a sum over stale slots.
template <typename T> __device__ __forceinline__ T comm_poison();
template <> __device__ __forceinline__ float comm_poison<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double comm_poison<double>() { return __builtin_nan(""); }
template <> __device__ __forceinline__ int comm_poison<int>() { return (int)0x80000000; }

// Waits until *wf == seq or the bound passes; an arena that has timed out once gives up at once from then on (a broken
// link costs one bound, not one per collective).  -> true when the flag arrived.
__device__ __forceinline__ bool comm_wait_flag(unsigned* wf, unsigned seq, CommHeader* hdr, const CommArgs& a) {
  const long long t0 = wall_clock64();
  const bool dead = __hip_atomic_load(&hdr->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  while ((a.light ? __hip_atomic_load(wf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                  : __hip_atomic_load(wf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) != seq) {
    __builtin_amdgcn_s_sleep(2);
    if (dead || wall_clock64() - t0 > a.timeout_ticks) {
      __hip_atomic_store(&hdr->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.host_error) __hip_atomic_store(a.host_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
  }
  return true;
}

__host__ __device__ inline size_t comm_flags_off() { return 64; }
__host__ __device__ inline size_t comm_data_off(int max_blocks) {
  const size_t f = 64 + (size_t)2 * COMM_MAX_RANKS * max_blocks * sizeof(unsigned);
  return (f + 255) / 256 * 256;
}


// All-reduce (sum, rank order) of n <= 64 doubles held in LDS by one workgroup, as chunk 0 of a collective of its own
// on the channel of `a` (same sequence / buffer protocol as comm_allreduce_kernel).  Every thread must call it.
__device__ __forceinline__ void comm_allreduce_small(const CommArgs& a, double* vals, int n) {
  char* mine = (char*)a.arena[a.rank];
  CommHeader* hdr = (CommHeader*)mine;
  const unsigned seq = __hip_atomic_load(&hdr->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int buf = seq & 1, W = a.world, r = a.rank;
  const size_t slot_bytes = (size_t)a.max_elems * 8, doff = comm_data_off(a.max_blocks);
  __syncthreads();
  for (int j = 0; j < W; ++j) {
    double* dst = (double*)((char*)a.arena[j] + doff + ((size_t)buf * COMM_MAX_RANKS + r) * slot_bytes);
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = vals[i];
  }
  __shared__ int comm_lost;
  if (threadIdx.x == 0) comm_lost = 0;
  comm_publish_fence(a);
  __syncthreads();
  if ((int)threadIdx.x < W) {
    unsigned* pf = (unsigned*)((char*)a.arena[threadIdx.x] + comm_flags_off()) + ((size_t)buf * COMM_MAX_RANKS + r) * a.max_blocks;
    comm_store_flag(a, pf, seq);
    unsigned* wf = (unsigned*)(mine + comm_flags_off()) + ((size_t)buf * COMM_MAX_RANKS + threadIdx.x) * a.max_blocks;
    if (!comm_wait_flag(wf, seq, hdr, a)) comm_lost = 1;
  }
  __syncthreads();
  const bool lost = comm_lost != 0;
  const double* base = (const double*)(mine + doff + (size_t)buf * COMM_MAX_RANKS * slot_bytes);
  const size_t stride = slot_bytes / sizeof(double);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double s = __builtin_nontemporal_load(base + i);
    for (int j = 1; j < W; ++j) s += __builtin_nontemporal_load(base + (size_t)j * stride + i);
    vals[i] = lost ? comm_poison<double>() : s;  // a lost peer shows up as NaN statistics, never as a partial sum
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&hdr->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One chunk of one vector: `k` = the chunk's flag index within the collective, `e0`/`cnt` = its elements of `data`,
// `off` = byte offset of the vector inside a slot.
template <typename T>
__device__ __forceinline__ void comm_chunk(const CommArgs& a, T* data, long e0, long cnt, size_t off, int k, unsigned seq,
                                           CommHeader* hdr, int* lost_s) {
  char* mine = (char*)a.arena[a.rank];
  const int buf = seq & 1, W = a.world, r = a.rank;
  const size_t slot_bytes = (size_t)a.max_elems * 8, doff = comm_data_off(a.max_blocks);
  T* src = data + e0;
  // (a) my chunk into slot r of every rank
  for (int j = 0; j < W; ++j) {
    T* dst = (T*)((char*)a.arena[j] + doff + ((size_t)buf * COMM_MAX_RANKS + r) * slot_bytes + off) + e0;
    for (long i = threadIdx.x; i < cnt; i += blockDim.x) dst[i] = src[i];
  }
  if (threadIdx.x == 0) *lost_s = 0;
  comm_publish_fence(a);
  __syncthreads();
  // (b) stamp, (c) wait
  if ((int)threadIdx.x < W) {
    unsigned* pf = (unsigned*)((char*)a.arena[threadIdx.x] + comm_flags_off()) + ((size_t)buf * COMM_MAX_RANKS + r) * a.max_blocks + k;
    comm_store_flag(a, pf, seq);
    unsigned* wf = (unsigned*)(mine + comm_flags_off()) + ((size_t)buf * COMM_MAX_RANKS + threadIdx.x) * a.max_blocks + k;
    if (!comm_wait_flag(wf, seq, hdr, a)) *lost_s = 1;
  }
  __syncthreads();
  // (d) fixed-order sum of the W slots of my own arena -- or, when a peer never arrived, poison: a failed exchange must
  // be visible in the weights (NaN), not look like a gradient
  const bool lost = *lost_s != 0;
  const T* base = (const T*)(mine + doff + (size_t)buf * COMM_MAX_RANKS * slot_bytes + off) + e0;
  const size_t stride = slot_bytes / sizeof(T);
  for (long i = threadIdx.x; i < cnt; i += blockDim.x) {
    T s = __builtin_nontemporal_load(base + i);
    for (int j = 1; j < W; ++j) s += __builtin_nontemporal_load(base + (size_t)j * stride + i);
    src[i] = lost ? comm_poison<T>() : s;
  }
}

static inline CommArgs comm_make_args(void* const* arenas, int rank, int world, long max_elems) {
  CommArgs a;
  for (int j = 0; j < COMM_MAX_RANKS; ++j) a.arena[j] = j < world ? arenas[j] : nullptr;
  a.data = nullptr; a.n = 0; a.data2 = nullptr; a.n2 = 0; a.max_elems = max_elems; a.rank = rank; a.world = world;
  a.max_blocks = cdiv(max_elems * 2, COMM_CHUNK); a.dtype = 1;
  a.timeout_ticks = comm_timeout_ticks(); a.host_error = comm_host_error(); a.light = comm_light();
  return a;
}
