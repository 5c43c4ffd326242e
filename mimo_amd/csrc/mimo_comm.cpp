// Multi-GPU through the C ABI: the one exchange step of the sharded path — sum of the packed statistic block over the
// ranks — on RCCL directly, so that a host that is not Python (plain C, cgo, JNI) can shard the rows over the GPUs of
// a node exactly like mimo_amd/sharded.py does through torch.distributed.  One process per GPU; the application moves
// the 128-byte unique id from rank 0 to the others (MPI, a file, a socket — as with any NCCL program).
//
// The sum is taken in RANK ORDER ("for determinism sum in fixed rank order"): every rank all-gathers the
// G blocks (8 x 1.08 MB at the C5 shape: latency-bound either way) and adds them itself, block 0 first — the association does
// not depend on RCCL's choice of algorithm, channel count or NCCL_* settings, as a ring / tree all-reduce's does, so a sweep is
// bit-identical from launch to launch on 8 GPUs as it is on one.  MIMO_COMM_RANK_ORDER=0 selects ncclAllReduce(sum).
//
// librccl is opened at run time (dlopen), not linked: the library still loads on a box without RCCL, and inside a
// PyTorch process the copy PyTorch already brought in is the one that is used.
#include "../../include/mimo_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace mimo_comm {

struct Api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
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
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.handle, "ncclAllGather"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.GetErrorString;
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

// what mimo_ctx::comm points to: the communicator and the gather buffer of the rank-ordered sum
struct Comm {
  ncclComm_t nccl = nullptr;
  int world = 1;
  double* gather = nullptr;      // [world][cap] device
  size_t cap = 0;
};

static bool rank_order() {
  static const bool on = [] { const char* e = getenv("MIMO_COMM_RANK_ORDER"); return !e || atoi(e) != 0; }();
  return on;
}

// out[e] = ((g[0][e] + g[1][e]) + g[2][e]) + ...  — the same association on every rank, whatever the transport did
__global__ void rank_order_sum_kernel(const double* __restrict__ g, int world, size_t count, double* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  double s = g[e];
  for (int r = 1; r < world; ++r) s += g[(size_t)r * count + e];
  out[e] = s;
}

int init(void** comm, const char* id128, int rank, int world, char* msg, size_t msglen) {
  Api& a = api();
  if (!a.ok) { snprintf(msg, msglen, "librccl could not be opened"); return MIMO_E_UNSUPPORTED; }
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  ncclComm_t c = nullptr;
  const ncclResult_t r = a.CommInitRank(&c, world, id, rank);
  if (r != ncclSuccess) { snprintf(msg, msglen, "ncclCommInitRank(rank %d of %d): %s", rank, world, a.GetErrorString(r)); return MIMO_E_HIP; }
  Comm* cm = new (std::nothrow) Comm();
  if (!cm) { (void)a.CommDestroy(c); snprintf(msg, msglen, "out of host memory"); return MIMO_E_NOMEM; }
  cm->nccl = c; cm->world = world;
  *comm = cm;
  return MIMO_OK;
}

int destroy(void* comm) {
  Comm* cm = static_cast<Comm*>(comm);
  if (!cm) return MIMO_OK;
  if (cm->nccl && api().ok) (void)api().CommDestroy(cm->nccl);
  if (cm->gather) (void)hipFree(cm->gather);
  delete cm;
  return MIMO_OK;
}

int allreduce_sum_f64(void* comm, double* buf, size_t count, hipStream_t stream, char* msg, size_t msglen) {
  Api& a = api();
  Comm* cm = static_cast<Comm*>(comm);
  const bool force = getenv("MIMO_COMM_RANK_ORDER_FORCE") != nullptr;     // (tests: the gather + ordered sum with one rank)
  if (!rank_order() || (cm->world == 1 && !force)) {
    const ncclResult_t r = a.AllReduce(buf, buf, count, ncclDouble, ncclSum, cm->nccl, stream);
    if (r != ncclSuccess) { snprintf(msg, msglen, "ncclAllReduce: %s", a.GetErrorString(r)); return MIMO_E_HIP; }
    return MIMO_OK;
  }
  if (cm->cap < count) {
    if (cm->gather) { (void)hipStreamSynchronize(stream); (void)hipFree(cm->gather); cm->gather = nullptr; cm->cap = 0; }
    if (hipMalloc(reinterpret_cast<void**>(&cm->gather), (size_t)cm->world * count * sizeof(double)) != hipSuccess) {
      snprintf(msg, msglen, "hipMalloc of the gather buffer (%zu doubles x %d ranks) failed", count, cm->world);
      return MIMO_E_NOMEM;
    }
    cm->cap = count;
  }
  const ncclResult_t r = a.AllGather(buf, cm->gather, count, ncclDouble, cm->nccl, stream);
  if (r != ncclSuccess) { snprintf(msg, msglen, "ncclAllGather: %s", a.GetErrorString(r)); return MIMO_E_HIP; }
  hipLaunchKernelGGL(rank_order_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, cm->gather, cm->world, count, buf);
  if (hipGetLastError() != hipSuccess) { snprintf(msg, msglen, "rank_order_sum_kernel launch failed"); return MIMO_E_HIP; }
  return MIMO_OK;
}

}  // namespace mimo_comm
