// sfmhip_rccl.cpp -- RCCL binding of the sharded BA (include/sfmhip_rccl.h): the all-reduce callback of
// sfmhip_ba_set_allreduce implemented as ncclAllReduce(sum, ncclDouble) over xGMI.
#include "../../../include/sfmhip_rccl.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <map>
#include <mutex>
#include <string.h>

static_assert(sizeof(ncclUniqueId) == SFMHIP_RCCL_ID_BYTES, "ncclUniqueId size");

namespace {
thread_local int g_last_nccl = 0;
struct Route {
  ncclComm_t comm;
  hipStream_t stream;
};
// one route per problem object; owned here (the ABI's callback takes a plain user pointer)
std::mutex g_mu;
std::map<sfmhip_ba*, Route*> g_routes;

int allreduce_cb(void* device_f64_buffer, size_t count, void* user) {
  const Route* r = (const Route*)user;
  const ncclResult_t rc = ncclAllReduce(device_f64_buffer, device_f64_buffer, count, ncclDouble, ncclSum, r->comm, r->stream);
  if (rc != ncclSuccess) {
    g_last_nccl = (int)rc;
    return -1;
  }
  return 0;
}
}  // namespace

extern "C" int sfmhip_rccl_unique_id(void* id_out) {
  if (!id_out) return SFMHIP_ERR_ARG;
  ncclUniqueId id;
  const ncclResult_t rc = ncclGetUniqueId(&id);
  if (rc != ncclSuccess) {
    g_last_nccl = (int)rc;
    return SFMHIP_ERR_COMM;
  }
  memcpy(id_out, &id, sizeof id);
  return SFMHIP_OK;
}

extern "C" int sfmhip_rccl_comm_create(sfmhip_ctx* ctx, int rank, int world, const void* id, void** comm) {
  if (!ctx || !id || !comm || world < 1 || rank < 0 || rank >= world) return SFMHIP_ERR_ARG;
  if (hipSetDevice(sfmhip_device(ctx)) != hipSuccess) return SFMHIP_ERR_HIP;
  ncclUniqueId nid;
  memcpy(&nid, id, sizeof nid);
  ncclComm_t c = nullptr;
  const ncclResult_t rc = ncclCommInitRank(&c, world, nid, rank);
  if (rc != ncclSuccess) {
    g_last_nccl = (int)rc;
    return SFMHIP_ERR_COMM;
  }
  *comm = (void*)c;
  return SFMHIP_OK;
}

extern "C" int sfmhip_ba_use_rccl(sfmhip_ba* ba, sfmhip_ctx* ctx, void* comm, int rank, int world) {
  if (!ba || !ctx || !comm) return SFMHIP_ERR_ARG;
  Route* r = new Route{(ncclComm_t)comm, (hipStream_t)sfmhip_stream(ctx)};
  const int rc = sfmhip_ba_set_allreduce(ba, allreduce_cb, r, rank, world);
  if (rc != SFMHIP_OK) {
    delete r;
    return rc;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  Route*& slot = g_routes[ba];
  delete slot;
  slot = r;
  return SFMHIP_OK;
}

extern "C" void sfmhip_rccl_comm_destroy(void* comm) {
  if (!comm) return;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_routes.begin(); it != g_routes.end();) {
      if (it->second->comm == (ncclComm_t)comm) {
        delete it->second;
        it = g_routes.erase(it);
      } else {
        ++it;
      }
    }
  }
  ncclCommDestroy((ncclComm_t)comm);
}

extern "C" int sfmhip_rccl_last_error(void) { return g_last_nccl; }
