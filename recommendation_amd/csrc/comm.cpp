// Collectives of the sharded embedding path behind the C ABI: one communicator per process and GPU over RCCL
// (xGMI inside a node).  RCCL is loaded at the first call (dlopen), so the library itself has no link-time
// dependency on it and loads on a machine without RCCL; a process that already holds an RCCL (PyTorch's) gets
// that same copy.  Everything is enqueued on the caller's stream; nothing here synchronises the host.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

struct ctr_comm {
  ncclComm_t nccl;
  int world, rank, device;
};

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_error[256] = "";

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl.so not found: %s", dlerror());
    return;
  }
  bool ok = true;
  auto sym = [&](const char* name) {
    void* p = dlsym(g_rccl.handle, name);
    if (!p) {
      ok = false;
      snprintf(g_rccl_error, sizeof(g_rccl_error), "librccl.so lacks %s", name);
    }
    return p;
  };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
  g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
  g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!ok) g_rccl.handle = nullptr;
}

int need_rccl(const char* who) {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.handle) {
    ctr::set_error("%s: %s", who, g_rccl_error);
    return CTR_EINVAL;
  }
  return CTR_OK;
}

}  // namespace

#define CTR_NCCL(call)                                                                  \
  do {                                                                                  \
    const ncclResult_t r__ = (call);                                                    \
    if (r__ != ncclSuccess) {                                                           \
      ctr::set_error("%s: RCCL: %s", __func__, g_rccl.GetErrorString(r__));             \
      return CTR_EINVAL;                                                                \
    }                                                                                   \
  } while (0)

extern "C" int ctr_comm_unique_id(void* id_out) {
  CTR_REQUIRE(id_out, "ctr_comm_unique_id: null pointer");
  if (int rc = need_rccl("ctr_comm_unique_id")) return rc;
  static_assert(sizeof(ncclUniqueId) == CTR_COMM_ID_BYTES, "CTR_COMM_ID_BYTES must be the size of RCCL's unique id");
  CTR_NCCL(g_rccl.GetUniqueId((ncclUniqueId*)id_out));
  return CTR_OK;
}

extern "C" int ctr_comm_init(const void* id, int world, int rank, ctr_comm_t** comm_out) {
  CTR_REQUIRE(id && comm_out, "ctr_comm_init: null pointer");
  CTR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ctr_comm_init: rank %d of %d", rank, world);
  if (int rc = need_rccl("ctr_comm_init")) return rc;
  int dev = -1;
  CTR_HIP(hipGetDevice(&dev));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  CTR_NCCL(g_rccl.CommInitRank(&c, world, uid, rank));
  ctr_comm* out = new ctr_comm{c, world, rank, dev};
  *comm_out = out;
  return CTR_OK;
}

extern "C" int ctr_comm_destroy(ctr_comm_t* comm) {
  if (!comm) return CTR_OK;
  if (int rc = need_rccl("ctr_comm_destroy")) return rc;
  const ncclResult_t r = g_rccl.CommDestroy(comm->nccl);
  delete comm;
  if (r != ncclSuccess) {
    ctr::set_error("ctr_comm_destroy: RCCL: %s", g_rccl.GetErrorString(r));
    return CTR_EINVAL;
  }
  return CTR_OK;
}

extern "C" int ctr_comm_world(const ctr_comm_t* comm) { return comm ? comm->world : 0; }
extern "C" int ctr_comm_rank(const ctr_comm_t* comm) { return comm ? comm->rank : -1; }

extern "C" int ctr_allreduce_sum_f32(ctr_comm_t* comm, float* buf, int64_t n, void* stream) {
  CTR_REQUIRE(comm && n >= 0, "ctr_allreduce_sum_f32: bad argument");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(buf, "ctr_allreduce_sum_f32: null pointer");
  CTR_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm->nccl, (hipStream_t)stream));
  return CTR_OK;
}

extern "C" int ctr_all2allv(ctr_comm_t* comm, const void* send, const int64_t* send_offset, const int64_t* send_bytes,
                            void* recv, const int64_t* recv_offset, const int64_t* recv_bytes, void* stream) {
  CTR_REQUIRE(comm && send_offset && send_bytes && recv_offset && recv_bytes, "ctr_all2allv: null pointer");
  const int W = comm->world;
  for (int p = 0; p < W; ++p)
    CTR_REQUIRE(send_bytes[p] >= 0 && recv_bytes[p] >= 0 && send_offset[p] >= 0 && recv_offset[p] >= 0 &&
                    (send_bytes[p] == 0 || send) && (recv_bytes[p] == 0 || recv),
                "ctr_all2allv: bad piece for peer %d", p);
  hipStream_t st = (hipStream_t)stream;
  CTR_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < W; ++p) {
    ncclResult_t r = ncclSuccess;
    if (send_bytes[p] > 0)
      r = g_rccl.Send((const char*)send + send_offset[p], (size_t)send_bytes[p], ncclUint8, p, comm->nccl, st);
    if (r == ncclSuccess && recv_bytes[p] > 0)
      r = g_rccl.Recv((char*)recv + recv_offset[p], (size_t)recv_bytes[p], ncclUint8, p, comm->nccl, st);
    if (r != ncclSuccess) {
      g_rccl.GroupEnd();
      ctr::set_error("ctr_all2allv: RCCL: %s", g_rccl.GetErrorString(r));
      return CTR_EINVAL;
    }
  }
  CTR_NCCL(g_rccl.GroupEnd());
  return CTR_OK;
}

extern "C" int ctr_all2all(ctr_comm_t* comm, const void* send, void* recv, int64_t bytes_per_peer, void* stream) {
  CTR_REQUIRE(comm && bytes_per_peer >= 0, "ctr_all2all: bad argument");
  if (bytes_per_peer == 0) return CTR_OK;
  CTR_REQUIRE(send && recv, "ctr_all2all: null pointer");
  const int W = comm->world;
  hipStream_t st = (hipStream_t)stream;
  CTR_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < W; ++p) {
    ncclResult_t r = g_rccl.Send((const char*)send + (int64_t)p * bytes_per_peer, (size_t)bytes_per_peer, ncclUint8, p,
                                 comm->nccl, st);
    if (r == ncclSuccess)
      r = g_rccl.Recv((char*)recv + (int64_t)p * bytes_per_peer, (size_t)bytes_per_peer, ncclUint8, p, comm->nccl, st);
    if (r != ncclSuccess) {
      g_rccl.GroupEnd();
      ctr::set_error("ctr_all2all: RCCL: %s", g_rccl.GetErrorString(r));
      return CTR_EINVAL;
    }
  }
  CTR_NCCL(g_rccl.GroupEnd());
  return CTR_OK;
}
