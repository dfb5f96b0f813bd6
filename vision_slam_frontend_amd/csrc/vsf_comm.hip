// vsf_comm.hip -- the multi-GPU exchange of the hot path behind the C ABI (the seam lists
// "vsf_gather_* for multi-GPU"; BASELINE configs[3]: "RCCL-over-xGMI gather of VisionFeature / FeatureMatch outputs").
//
// What crosses GPUs is what the reference's algorithm forces: the per-frame mean epipolar residuals
// (the static threshold of RemoveAmbigStereo crosses frames, slam_frontend.cc:353, 392-394), the last `window` filtered
// frames of every rank (the temporal GetFeatureMatches of cc:424-434 needs frames k-1 .. k-window) and the compact
// VisionFeature / FeatureMatch payloads that one process assembles into the SLAMProblem (cc:498-503).  Three entry points
// carry all of it: an all-gather of equal-sized blocks and a sized gather to a root, both stream-ordered on the context's
// stream, plus the communicator's life cycle.  RCCL is bound at run time (dlopen of librccl -- the copy already in the
// process if there is one, e.g. PyTorch's --, never at link time: a single-GPU user of libvsf_hip.so does not need it).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>

#include <rccl/rccl.h>

#include "vsf_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already loaded first (PyTorch ships its own librccl.so: two RCCL instances in one process would each
    // open the devices' IPC resources), then the ROCm installation's
    for (const char* name : {"librccl.so", "librccl.so.1"})
      if (!r.handle) r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!r.handle) r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) return;
    auto sym = [&](const char* n) { return dlsym(r.handle, n); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart &&
           r.GroupEnd && r.GetVersion;
  });
  return r;
}

}  // namespace

struct vsf_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0, version = 0;
  int last_nccl = 0;
};

static_assert(VSF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's ncclUniqueId");

extern "C" {

vsf_status vsf_comm_unique_id(uint8_t* id) {
  if (!id) return VSF_ERR_INVALID_ARG;
  Rccl& r = rccl();
  if (!r.ok) return VSF_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (r.GetUniqueId(&u) != ncclSuccess) return VSF_ERR_HIP;
  std::memcpy(id, u.internal, VSF_COMM_ID_BYTES);
  return VSF_OK;
}

vsf_status vsf_comm_create(vsf_ctx* ctx, const uint8_t* id, int rank, int world, vsf_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return VSF_ERR_INVALID_ARG;
  *out = nullptr;
  Rccl& r = rccl();
  if (!r.ok) return VSF_ERR_UNSUPPORTED;
  vsf_comm* c = new (std::nothrow) vsf_comm();
  if (!c) return VSF_ERR_INVALID_ARG;
  c->rank = rank;
  c->world = world;
  c->device = vsf_ctx_device(ctx);
  if (hipSetDevice(c->device) != hipSuccess) {
    delete c;
    return VSF_ERR_HIP;
  }
  ncclUniqueId u;
  std::memcpy(u.internal, id, VSF_COMM_ID_BYTES);
  const ncclResult_t e = r.CommInitRank(&c->comm, world, u, rank);  // (collective: every rank of the world calls it)
  if (e != ncclSuccess) {
    delete c;
    return VSF_ERR_HIP;
  }
  (void)r.GetVersion(&c->version);
  *out = c;
  return VSF_OK;
}

void vsf_comm_destroy(vsf_comm* comm) {
  if (!comm) return;
  if (comm->comm) {
    (void)hipSetDevice(comm->device);
    (void)rccl().CommDestroy(comm->comm);
  }
  delete comm;
}

vsf_status vsf_comm_info(const vsf_comm* comm, int* rank, int* world, int* rccl_version) {
  if (!comm) return VSF_ERR_INVALID_ARG;
  if (rank) *rank = comm->rank;
  if (world) *world = comm->world;
  if (rccl_version) *rccl_version = comm->version;
  return VSF_OK;
}

vsf_status vsf_allgather_dev(vsf_ctx* ctx, vsf_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  if (!ctx || !comm || !d_send || !d_recv || bytes_per_rank == 0) return VSF_ERR_INVALID_ARG;
  if (hipSetDevice(comm->device) != hipSuccess) return VSF_ERR_HIP;
  const ncclResult_t e =
      rccl().AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, comm->comm, vsf_ctx_stream(ctx));
  comm->last_nccl = (int)e;
  return e == ncclSuccess ? VSF_OK : VSF_ERR_HIP;
}

vsf_status vsf_gather_payload_dev(vsf_ctx* ctx, vsf_comm* comm, const uint8_t* d_send, size_t bytes, uint8_t* d_recv,
                                  size_t recv_stride, int root) {
  if (!ctx || !comm || !d_send || bytes == 0 || root < 0 || root >= comm->world) return VSF_ERR_INVALID_ARG;
  if (comm->rank == root && (!d_recv || recv_stride < bytes)) return VSF_ERR_INVALID_ARG;
  if (hipSetDevice(comm->device) != hipSuccess) return VSF_ERR_HIP;
  Rccl& r = rccl();
  hipStream_t s = vsf_ctx_stream(ctx);
  // One group of point-to-point transfers: every rank sends `bytes` to the root, the root posts one receive per rank
  // (its own included).  On a fully connected xGMI node each peer -> root transfer rides its own link.
  ncclResult_t e = r.GroupStart();
  if (e == ncclSuccess && comm->rank == root)
    for (int p = 0; p < comm->world && e == ncclSuccess; p++)
      e = r.Recv(d_recv + (size_t)p * recv_stride, bytes, ncclUint8, p, comm->comm, s);
  if (e == ncclSuccess) e = r.Send(d_send, bytes, ncclUint8, root, comm->comm, s);
  const ncclResult_t e2 = r.GroupEnd();
  if (e == ncclSuccess) e = e2;
  comm->last_nccl = (int)e;
  return e == ncclSuccess ? VSF_OK : VSF_ERR_HIP;
}

}  // extern "C"
