// Data-parallel gradient exchange: direct RCCL calls behind the C-ABI (replaces the reduce half of nn.DataParallel,
// train3D.py:119; SURVEY.md section 8e: one bucketed all-reduce of the gradients per optimizer step).
//
// Why not torch.distributed's ProcessGroupNCCL: the step is replayed from a HIP graph with the collectives captured as side
// branches.  ProcessGroupNCCL runs a watchdog thread that polls the completion events of eagerly issued collectives; once a
// capture has pulled the group's stream in, such a poll fails with hipErrorCapturedEvent and the watchdog aborts the process
// (round 2 dodged it with a sleep).  A communicator driven from here has no thread of its own: ncclAllReduce only enqueues the
// RCCL kernel on the stream it is given - inside a capture it becomes a graph node like any other launch.
//
// librccl is NOT linked: the process already holds PyTorch's copy (torch/lib/librccl.so, bound to PyTorch's HIP runtime, which is
// the runtime that owns our streams and pointers).  ltu_comm_load dlopens that file by path and resolves the five entry points.
// The only process-global state are these immutable function pointers; communicators are opaque handles owned by the caller.
#include <dlfcn.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace {

struct RcclUniqueId { char internal[128]; };       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;                            // ncclComm_t
typedef int (*fn_get_unique_id)(RcclUniqueId*);
typedef int (*fn_comm_init_rank)(RcclComm*, int, RcclUniqueId, int);
typedef int (*fn_comm_destroy)(RcclComm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
enum { RCCL_FLOAT32 = 7, RCCL_UINT8 = 1, RCCL_AVG = 4 };      // ncclFloat32, ncclUint8, ncclAvg (rccl.h:448-470)

struct RcclApi {
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_broadcast broadcast = nullptr;
};
RcclApi g_api;
std::atomic<bool> g_loaded{false};
std::mutex g_load_mu;

inline int rccl_rc(int r) { return r == 0 ? LTU_OK : LTU_E_COMM - r; }      // ncclResult_t r > 0 -> LTU_E_COMM - r

}  // namespace

extern "C" int ltu_comm_load(const char* librccl_path) {
  if (g_loaded.load(std::memory_order_acquire)) return LTU_OK;
  if (librccl_path == nullptr) return LTU_E_ARG;
  std::lock_guard<std::mutex> lk(g_load_mu);
  if (g_loaded.load(std::memory_order_relaxed)) return LTU_OK;
  void* h = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) return LTU_E_COMM;
  RcclApi a;
  a.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  a.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  a.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  a.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  a.broadcast = (fn_broadcast)dlsym(h, "ncclBroadcast");
  if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_reduce || !a.broadcast) return LTU_E_COMM;
  g_api = a;
  g_loaded.store(true, std::memory_order_release);
  return LTU_OK;
}

extern "C" int ltu_comm_unique_id(void* id128) {
  if (!g_loaded.load(std::memory_order_acquire)) return LTU_E_COMM;
  if (id128 == nullptr) return LTU_E_ARG;
  return rccl_rc(g_api.get_unique_id(reinterpret_cast<RcclUniqueId*>(id128)));
}

extern "C" int ltu_comm_init(void** comm, const void* id128, int world, int rank) {
  if (!g_loaded.load(std::memory_order_acquire)) return LTU_E_COMM;
  if (comm == nullptr || id128 == nullptr || world < 1 || rank < 0 || rank >= world) return LTU_E_ARG;
  RcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  RcclComm c = nullptr;
  const int r = g_api.comm_init_rank(&c, world, id, rank);
  *comm = c;
  return rccl_rc(r);
}

extern "C" int ltu_comm_allreduce_avg(void* comm, float* buf, long long n, ltu_stream_t s) {
  if (!g_loaded.load(std::memory_order_acquire)) return LTU_E_COMM;
  if (comm == nullptr || buf == nullptr || n < 0) return LTU_E_ARG;
  if (n == 0) return LTU_OK;
  return rccl_rc(g_api.all_reduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_AVG, (RcclComm)comm, (hipStream_t)s));
}

extern "C" int ltu_comm_broadcast(void* comm, void* buf, long long nbytes, int root, ltu_stream_t s) {
  if (!g_loaded.load(std::memory_order_acquire)) return LTU_E_COMM;
  if (comm == nullptr || buf == nullptr || nbytes < 0) return LTU_E_ARG;
  if (nbytes == 0) return LTU_OK;
  return rccl_rc(g_api.broadcast(buf, buf, (size_t)nbytes, RCCL_UINT8, root, (RcclComm)comm, (hipStream_t)s));
}

extern "C" int ltu_comm_destroy(void* comm) {
  if (!g_loaded.load(std::memory_order_acquire)) return LTU_E_COMM;
  if (comm == nullptr) return LTU_OK;
  return rccl_rc(g_api.comm_destroy((RcclComm)comm));
}
