// comm.hip — multi-GPU exchange steps of the hot path: one process per GPU, RCCL over xGMI.
//
// The reference is single-process (no collective exists in it); what shards are BA points (one
// all-reduce(sum) of the reduced camera system S | b, D*D + D doubles, per BA iteration, T:893-1071) and RANSAC
// hypotheses (all-reduce(max) of a packed (count, iteration) key, T:664-677).  Both payloads are tiny (<= 29 KB), i.e.
// latency bound, so they are single collectives on the calling context's stream, in HBM, with no host bounce.
//
// librccl is bound lazily with dlopen (a single-GPU run never loads it; inside a PyTorch process the already loaded
// librccl.so.1 is the one that is found).  The unique id is created by rank 0 (sfmx_comm_get_unique_id) and carried to
// the other ranks by the application (bench.py: torch.distributed; CLI: a file).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "sfmx_internal.h"

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
  });
  return api;
}
}  // namespace

// dtype: 0 = f64, 1 = u64; op: 0 = sum, 1 = max.  In place on device memory, ordered on the context's stream.
int sfmx_comm_allreduce_dev(sfmx_ctx* c, sfmx_comm* comm, void* dev, size_t count, int dtype, int op) {
  if (!comm || comm->world <= 1 || count == 0) return SFMX_OK;
  RcclApi& api = rccl();
  const ncclResult_t r = api.AllReduce(dev, dev, count, dtype == 0 ? ncclFloat64 : ncclUint64, op == 0 ? ncclSum : ncclMax,
                                       static_cast<ncclComm_t>(comm->nccl), c->stream);
  if (r != ncclSuccess) {
    c->err = std::string("ncclAllReduce: ") + api.GetErrorString(r);
    return SFMX_ERR_HIP;
  }
  return SFMX_OK;
}

extern "C" {

int sfmx_comm_get_unique_id(void* id_out) {
  if (!id_out) return SFMX_ERR_INVALID;
  RcclApi& api = rccl();
  if (!api.ok) return SFMX_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (api.GetUniqueId(&id) != ncclSuccess) return SFMX_ERR_HIP;
  static_assert(sizeof(id) == SFMX_COMM_ID_BYTES, "unique id size");
  memcpy(id_out, &id, sizeof id);
  return SFMX_OK;
}

int sfmx_comm_create(int device, const void* id_bytes, int rank, int world, sfmx_comm** out) {
  if (!out || world < 1 || rank < 0 || rank >= world) return SFMX_ERR_INVALID;
  *out = nullptr;
  sfmx_comm* cm = new sfmx_comm;
  cm->rank = rank;
  cm->world = world;
  cm->device = device;
  if (world > 1) {
    RcclApi& api = rccl();
    if (!api.ok || !id_bytes) { delete cm; return api.ok ? SFMX_ERR_INVALID : SFMX_ERR_UNSUPPORTED; }
    if (hipSetDevice(device) != hipSuccess) { delete cm; return SFMX_ERR_NO_DEVICE; }
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclComm_t nc = nullptr;
    if (api.CommInitRank(&nc, world, id, rank) != ncclSuccess) { delete cm; return SFMX_ERR_HIP; }
    cm->nccl = nc;
  }
  *out = cm;
  return SFMX_OK;
}

void sfmx_comm_destroy(sfmx_comm* cm) {
  if (!cm) return;
  if (cm->nccl) (void)rccl().CommDestroy(static_cast<ncclComm_t>(cm->nccl));
  delete cm;
}
int sfmx_comm_rank(const sfmx_comm* cm) { return cm ? cm->rank : 0; }
int sfmx_comm_world(const sfmx_comm* cm) { return cm ? cm->world : 1; }

void sfmx_shard_range(int n, int rank, int world, int* lo, int* hi) {
  if (world < 1) world = 1;
  const int base = n / world, extra = n % world;
  const int l = rank * base + (rank < extra ? rank : extra);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (rank < extra ? 1 : 0);
}

// small host payloads (a packed key, nine matrix entries): staged through the context, one collective
static int allreduce_host(sfmx_ctx* c, sfmx_comm* cm, void* host_inout, int n, int dtype, int op) {
  SFMX_REQUIRE(c, c && host_inout && n >= 0);
  if (!cm || cm->world <= 1 || n == 0) return SFMX_OK;
  const size_t nb = (size_t)n * 8;
  SFMX_HIP(c, c->d[2].ensure(nb));
  SFMX_HIP(c, c->h[2].ensure(nb));
  memcpy(c->h[2].p, host_inout, nb);
  SFMX_HIP(c, hipMemcpyAsync(c->d[2].p, c->h[2].p, nb, hipMemcpyHostToDevice, c->stream));
  const int rc = sfmx_comm_allreduce_dev(c, cm, c->d[2].p, (size_t)n, dtype, op);
  if (rc) return rc;
  SFMX_HIP(c, hipMemcpyAsync(c->h[2].p, c->d[2].p, nb, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(host_inout, c->h[2].p, nb);
  return SFMX_OK;
}
int sfmx_comm_allreduce_f64(sfmx_ctx* c, sfmx_comm* cm, double* host_inout, int n, int op) { return allreduce_host(c, cm, host_inout, n, 0, op); }
int sfmx_comm_allreduce_u64_max(sfmx_ctx* c, sfmx_comm* cm, uint64_t* host_inout, int n) { return allreduce_host(c, cm, host_inout, n, 1, 1); }

}  // extern "C"
