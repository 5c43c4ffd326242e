// Multi-GPU through the C ABI: the one exchange step of the sharded path — sum of the packed statistic block over the
// ranks — on RCCL directly, so that a host that is not Python (plain C, cgo, JNI) can shard the rows over the GPUs of
// a node exactly like mimo_amd/sharded.py does through torch.distributed.  One process per GPU; the application moves
// the 128-byte unique id from rank 0 to the others (MPI, a file, a socket — as with any NCCL program).
//
// librccl is opened at run time (dlopen), not linked: the library still loads on a box without RCCL, and inside a
// PyTorch process the copy PyTorch already brought in is the one that is used.
#include "../../include/mimo_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

namespace mimo_comm {

struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

static Api& api() {
  static Api a = [] {
    Api x;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.handle) break;
    }
    if (!x.handle) return x;
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.handle, "ncclAllReduce"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.GetErrorString;
    return x;
  }();
  return a;
}

// returns 0 or a negative MIMO_E_* code; msg receives a short description on failure
int unique_id(char* out128, char* msg, size_t msglen) {
  Api& a = api();
  if (!a.ok) {
    const char* why = dlerror();          // (a second call would return NULL: the message is consumed by the first)
    snprintf(msg, msglen, "librccl could not be opened (%s)", why ? why : "symbols missing");
    return MIMO_E_UNSUPPORTED;
  }
  ncclUniqueId id;
  const ncclResult_t r = a.GetUniqueId(&id);
  if (r != ncclSuccess) { snprintf(msg, msglen, "ncclGetUniqueId: %s", a.GetErrorString(r)); return MIMO_E_HIP; }
  static_assert(sizeof id.internal == 128, "unique id size");
  memcpy(out128, id.internal, 128);
  return MIMO_OK;
}

int init(void** comm, const char* id128, int rank, int world, char* msg, size_t msglen) {
  Api& a = api();
  if (!a.ok) { snprintf(msg, msglen, "librccl could not be opened"); return MIMO_E_UNSUPPORTED; }
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  ncclComm_t c = nullptr;
  const ncclResult_t r = a.CommInitRank(&c, world, id, rank);
  if (r != ncclSuccess) { snprintf(msg, msglen, "ncclCommInitRank(rank %d of %d): %s", rank, world, a.GetErrorString(r)); return MIMO_E_HIP; }
  *comm = c;
  return MIMO_OK;
}

int destroy(void* comm) {
  if (comm && api().ok) (void)api().CommDestroy(static_cast<ncclComm_t>(comm));
  return MIMO_OK;
}

int allreduce_sum_f64(void* comm, double* buf, size_t count, hipStream_t stream, char* msg, size_t msglen) {
  Api& a = api();
  const ncclResult_t r = a.AllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), stream);
  if (r != ncclSuccess) { snprintf(msg, msglen, "ncclAllReduce: %s", a.GetErrorString(r)); return MIMO_E_HIP; }
  return MIMO_OK;
}

}  // namespace mimo_comm
