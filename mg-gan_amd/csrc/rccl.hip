// RCCL all-reduce INSIDE the iteration graph (scene-sharded training; north_star: "RCCL all-reduce of
// discriminator/PM gradients over xGMI").  torch.distributed's collectives cannot be captured together with this library's
// launches (its watchdog queries events of the capturing stream), so they used to cut the iteration into 12 graph segments;
// here ncclAllReduce is bound straight from librccl.so -- the copy that is already in the process (torch links it) -- and
// issued on the caller's stream like any other launch of the library: eager, or captured into the ONE iteration graph.
// A gradient buffer and its f64 tail (csrc/comm.hip: mggan_comm_allreduce2 is the peer-mapped form of the same exchange)
// are two ncclAllReduce calls inside one ncclGroupStart/End: RCCL fuses a group into one kernel launch.
// The communicator is this library's own (mggan_rccl_comm_init from a 128-byte ncclUniqueId that rank 0 draws and the
// host side broadcasts once, mggan/devcomm.py: RcclComm); nothing of torch's process group is touched.
// No reference counterpart (the reference is single-process).
#include <dlfcn.h>
#include <string.h>
#include "common.h"
#include "../../include/mggan_hip.h"

namespace {

// the five entry points used, with the ABI of rccl.h (NCCL 2.27): ncclResult_t = int (0 = ncclSuccess),
// ncclDataType_t { ncclInt32 = 2, ncclFloat32 = 7, ncclFloat64 = 8 }, ncclRedOp_t { ncclSum = 0 }, ncclUniqueId = 128 bytes
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*CommDestroyFn)(void* comm);
typedef int (*AllReduceFn)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream);
typedef int (*GroupFn)(void);
typedef const char* (*ErrStrFn)(int);
typedef int (*CommAsyncErrFn)(void* comm, int* async_error);

struct Api {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
  ErrStrFn err_str = nullptr;
  CommAsyncErrFn async_err = nullptr;
  char why[256] = {0};
};

Api g_api;

Api* api() {
  Api& a = g_api;
  static bool tried = false;
  if (tried) return a.handle ? &a : nullptr;
  tried = true;
  // the copy already mapped into the process first (torch's librccl.so, soname librccl.so.1): two RCCL images in one
  // process would each bring their own kernels (0.5 GB) and their own idea of the topology
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !a.handle; ++pass)
    for (const char* n : names) {
      a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (a.handle) break;
    }
  if (!a.handle) {
    const char* e = dlerror();
    snprintf(a.why, sizeof(a.why), "dlopen(librccl.so.1): %s", e ? e : "not found");
    return nullptr;
  }
  a.get_unique_id = (GetUniqueIdFn)dlsym(a.handle, "ncclGetUniqueId");
  a.comm_init_rank = (CommInitRankFn)dlsym(a.handle, "ncclCommInitRank");
  a.comm_destroy = (CommDestroyFn)dlsym(a.handle, "ncclCommDestroy");
  a.all_reduce = (AllReduceFn)dlsym(a.handle, "ncclAllReduce");
  a.group_start = (GroupFn)dlsym(a.handle, "ncclGroupStart");
  a.group_end = (GroupFn)dlsym(a.handle, "ncclGroupEnd");
  a.err_str = (ErrStrFn)dlsym(a.handle, "ncclGetErrorString");
  a.async_err = (CommAsyncErrFn)dlsym(a.handle, "ncclCommGetAsyncError");
  if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_reduce || !a.group_start || !a.group_end) {
    snprintf(a.why, sizeof(a.why), "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / "
                                   "ncclAllReduce / ncclGroupStart / ncclGroupEnd");
    a.handle = nullptr;
    return nullptr;
  }
  return &a;
}

const char* why_not() { return g_api.why[0] ? g_api.why : "librccl.so could not be loaded"; }

#define RCCL_CALL(a, expr, what)                                                                       \
  do {                                                                                                 \
    const int r_ = (expr);                                                                             \
    if (r_ != 0) {                                                                                     \
      mggan_set_error("%s: %s (ncclResult %d)", what, (a)->err_str ? (a)->err_str(r_) : "RCCL error", r_); \
      return MGGAN_ERR_LAUNCH;                                                                         \
    }                                                                                                  \
  } while (0)

}  // namespace

extern "C" {

/* 1 when librccl.so resolves in this process (dlopen, the copy torch has loaded if there is one), else 0 */
int mggan_rccl_available(void) { return api() ? 1 : 0; }

/* id: 128 bytes (ncclUniqueId), drawn on ONE rank and handed to every rank's mggan_rccl_comm_init */
int mggan_rccl_unique_id(void* id) {
  MG_CHECK_ARG(id, "rccl_unique_id: null pointer");
  Api* a = api();
  MG_CHECK_ARG(a, "rccl_unique_id: %s", why_not());
  static_assert(sizeof(UniqueId) == 128, "ncclUniqueId is 128 bytes");
  UniqueId u;
  RCCL_CALL(a, a->get_unique_id(&u), "ncclGetUniqueId");
  memcpy(id, &u, sizeof(u));
  return MGGAN_OK;
}

/* collective: every rank calls it with the same id on its own (current) device; *comm: opaque communicator */
int mggan_rccl_comm_init(const void* id, int rank, int world, void** comm) {
  MG_CHECK_ARG(id && comm && world >= 1 && rank >= 0 && rank < world, "rccl_comm_init: bad arguments");
  Api* a = api();
  MG_CHECK_ARG(a, "rccl_comm_init: %s", why_not());
  UniqueId u;
  memcpy(&u, id, sizeof(u));
  void* c = nullptr;
  RCCL_CALL(a, a->comm_init_rank(&c, world, u, rank), "ncclCommInitRank");
  *comm = c;
  return MGGAN_OK;
}

int mggan_rccl_comm_destroy(void* comm) {
  Api* a = api();
  if (!a || !comm) return MGGAN_OK;
  RCCL_CALL(a, a->comm_destroy(comm), "ncclCommDestroy");
  return MGGAN_OK;
}

/* Sum over the ranks, in place, on `stream` (capturable): `data` (n elements; dtype 0 f32, 1 f64, 2 i32) and -- in the
   same group, i.e. one RCCL launch -- an optional f64 tail `data2` (n2 doubles; NULL / 0: none). */
int mggan_rccl_allreduce(void* comm, void* data, long n, int dtype, double* data2, long n2, hipStream_t stream) {
  MG_CHECK_ARG(comm && (data || n == 0) && (data2 || n2 == 0) && n >= 0 && n2 >= 0, "rccl_allreduce: bad arguments");
  MG_CHECK_ARG(dtype >= 0 && dtype <= 2, "rccl_allreduce: dtype %d (0 f32, 1 f64, 2 i32)", dtype);
  Api* a = api();
  MG_CHECK_ARG(a, "rccl_allreduce: %s", why_not());
  if (n == 0 && n2 == 0) return MGGAN_OK;
  static const int kType[3] = {7 /* ncclFloat32 */, 8 /* ncclFloat64 */, 2 /* ncclInt32 */};
  const bool group = n > 0 && n2 > 0;
  if (group) RCCL_CALL(a, a->group_start(), "ncclGroupStart");
  int r1 = 0, r2 = 0;
  if (n > 0) r1 = a->all_reduce(data, data, (size_t)n, kType[dtype], 0 /* ncclSum */, comm, stream);
  if (n2 > 0) r2 = a->all_reduce(data2, data2, (size_t)n2, 8, 0, comm, stream);
  if (group) RCCL_CALL(a, a->group_end(), "ncclGroupEnd");  // (always closed, also behind a failed call)
  RCCL_CALL(a, r1, "ncclAllReduce");
  RCCL_CALL(a, r2, "ncclAllReduce (f64 tail)");
  return MGGAN_OK;
}

/* *out = the communicator's asynchronous error state (0 = none): a host call, no device sync */
int mggan_rccl_async_error(void* comm, int* out) {
  MG_CHECK_ARG(comm && out, "rccl_async_error: null pointer");
  Api* a = api();
  MG_CHECK_ARG(a, "rccl_async_error: %s", why_not());
  *out = 0;
  if (a->async_err) RCCL_CALL(a, a->async_err(comm, out), "ncclCommGetAsyncError");
  return MGGAN_OK;
}

}  // extern "C"
