// RCCL behind the C ABI (dsee_comm_*): the three exchanges of the data-parallel path -- gradient sum, start-state broadcast,
// SyncBN statistics -- for a host that binds include/deepsee_hip.h directly and has no torch.distributed.
//
// RCCL is resolved at first use with dlopen (DSEE_RCCL_LIB, else the librccl.so.1 of the process / of /opt/rocm/lib), so
// libdeepsee_hip.so itself has no link-time dependency on it: one-GPU users never load a collective library, and a process
// that already holds torch's copy of RCCL shares it (same SONAME) instead of mapping a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "../../include/deepsee_hip.h"
#include "dsee_common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;
char g_load_error[256] = "";

template <typename F>
bool bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  if (!fn) snprintf(g_load_error, sizeof(g_load_error), "RCCL library lacks %s", name);
  return fn != nullptr;
}

void load_rccl() {
  const char* forced = getenv("DSEE_RCCL_LIB");
  const char* names[] = {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
    snprintf(g_load_error, sizeof(g_load_error), "dlopen(%s): %s", n, dlerror());
    if (n == forced) return;          // an explicit choice that fails is an error, not a reason to look elsewhere
  }
  if (!g_rccl.handle) return;
  const bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
                  bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.AllReduce, "ncclAllReduce") &&
                  bind(g_rccl.Broadcast, "ncclBroadcast") && bind(g_rccl.AllGather, "ncclAllGather") &&
                  bind(g_rccl.GetErrorString, "ncclGetErrorString");
  if (!ok) {
    dlclose(g_rccl.handle);
    g_rccl.handle = nullptr;
  }
}

int need_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.handle) {
    dsee_set_error("dsee_comm: RCCL not available (%s)", g_load_error);
    return DSEE_ELAUNCH;
  }
  return DSEE_OK;
}

struct Comm {
  ncclComm_t nccl;
  int world, rank, device;
};

#define DSEE_RCCL(call)                                                                   \
  do {                                                                                    \
    ncclResult_t r_ = (call);                                                             \
    if (r_ != ncclSuccess) {                                                              \
      dsee_set_error("%s: %s", #call, g_rccl.GetErrorString(r_));                         \
      return DSEE_ELAUNCH;                                                                \
    }                                                                                     \
  } while (0)

}  // namespace

extern "C" {

int dsee_comm_unique_id(void* id_out) {
  DSEE_CHECK_ARG(id_out);
  static_assert(sizeof(ncclUniqueId) == DSEE_COMM_ID_BYTES, "DSEE_COMM_ID_BYTES must match ncclUniqueId");
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  DSEE_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return DSEE_OK;
}

int dsee_comm_init(void** comm_out, const void* id, int world, int rank) {
  DSEE_CHECK_ARG(comm_out && id && world >= 1 && rank >= 0 && rank < world);
  if (int rc = need_rccl()) return rc;
  Comm* c = new Comm();
  c->world = world;
  c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    dsee_set_error("dsee_comm_init: no current HIP device");
    return DSEE_ELAUNCH;
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, uid, rank);
  if (r != ncclSuccess) {
    dsee_set_error("ncclCommInitRank(world %d, rank %d): %s", world, rank, g_rccl.GetErrorString(r));
    delete c;
    return DSEE_ELAUNCH;
  }
  *comm_out = c;
  return DSEE_OK;
}

int dsee_comm_world(const void* comm) { return comm ? static_cast<const Comm*>(comm)->world : 0; }

int dsee_comm_rank(const void* comm) { return comm ? static_cast<const Comm*>(comm)->rank : -1; }

int dsee_comm_allreduce_sum(void* comm, float* buf, long n, hipStream_t st) {
  DSEE_CHECK_ARG(comm && buf && n >= 0);
  if (n == 0) return DSEE_OK;
  Comm* c = static_cast<Comm*>(comm);
  DSEE_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->nccl, st));
  return DSEE_OK;
}

int dsee_comm_broadcast(void* comm, void* buf, long nbytes, int root, hipStream_t st) {
  DSEE_CHECK_ARG(comm && buf && nbytes >= 0);
  Comm* c = static_cast<Comm*>(comm);
  DSEE_CHECK_ARG(root >= 0 && root < c->world);
  if (nbytes == 0) return DSEE_OK;
  DSEE_RCCL(g_rccl.Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, c->nccl, st));
  return DSEE_OK;
}

int dsee_comm_allgather(void* comm, const void* send, void* recv, long nbytes_per_rank, hipStream_t st) {
  DSEE_CHECK_ARG(comm && send && recv && nbytes_per_rank >= 0);
  if (nbytes_per_rank == 0) return DSEE_OK;
  Comm* c = static_cast<Comm*>(comm);
  DSEE_RCCL(g_rccl.AllGather(send, recv, (size_t)nbytes_per_rank, ncclUint8, c->nccl, st));
  return DSEE_OK;
}

int dsee_comm_destroy(void* comm) {
  if (!comm) return DSEE_OK;
  Comm* c = static_cast<Comm*>(comm);
  ncclResult_t r = g_rccl.CommDestroy(c->nccl);
  delete c;
  if (r != ncclSuccess) {
    dsee_set_error("ncclCommDestroy: %s", g_rccl.GetErrorString(r));
    return DSEE_ELAUNCH;
  }
  return DSEE_OK;
}

}  // extern "C"
