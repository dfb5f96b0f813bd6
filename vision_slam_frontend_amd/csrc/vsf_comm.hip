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
#include <type_traits>

#include "vsf_internal.h"

// The handful of RCCL declarations this file needs, stated here (nccl.h's public, ABI-stable ones: RCCL keeps NCCL's API):
// the library is bound with dlopen so that a single-GPU user needs no RCCL at run time, and with these no RCCL development
// HEADERS are needed to build libvsf_hip.so either.  Where the header exists the block below checks them against it.
namespace vsfnccl {
typedef struct ncclComm* Comm_t;
struct UniqueId {
  char internal[128];
};
typedef int Result_t;                // ncclResult_t: 0 = ncclSuccess
constexpr Result_t Success = 0;
constexpr int Uint8 = 1;             // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1
typedef Result_t (*GetUniqueId_fn)(UniqueId*);
typedef Result_t (*CommInitRank_fn)(Comm_t*, int, UniqueId, int);
typedef Result_t (*CommDestroy_fn)(Comm_t);
typedef Result_t (*AllGather_fn)(const void*, void*, size_t, int, Comm_t, hipStream_t);
typedef Result_t (*Send_fn)(const void*, size_t, int, int, Comm_t, hipStream_t);
typedef Result_t (*Recv_fn)(void*, size_t, int, int, Comm_t, hipStream_t);
typedef Result_t (*Group_fn)();
typedef Result_t (*GetVersion_fn)(int*);
}  // namespace vsfnccl

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(sizeof(vsfnccl::UniqueId) == sizeof(ncclUniqueId) && NCCL_UNIQUE_ID_BYTES == 128, "ncclUniqueId");
static_assert((int)ncclSuccess == vsfnccl::Success && (int)ncclUint8 == vsfnccl::Uint8, "ncclResult_t / ncclDataType_t values");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int), "enums passed as int");
static_assert(std::is_same<decltype(&ncclAllGather), ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value, "ncclAllGather");
static_assert(std::is_same<decltype(&ncclSend), ncclResult_t (*)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclSend");
static_assert(std::is_same<decltype(&ncclRecv), ncclResult_t (*)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclRecv");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
#endif

namespace {

using namespace vsfnccl;

struct Rccl {
  void* handle = nullptr;
  GetUniqueId_fn GetUniqueId = nullptr;
  CommInitRank_fn CommInitRank = nullptr;
  CommDestroy_fn CommDestroy = nullptr;
  AllGather_fn AllGather = nullptr;
  Send_fn Send = nullptr;
  Recv_fn Recv = nullptr;
  Group_fn GroupStart = nullptr;
  Group_fn GroupEnd = nullptr;
  GetVersion_fn GetVersion = nullptr;
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
  vsfnccl::Comm_t comm = nullptr;
  int rank = 0, world = 1, device = 0, version = 0;
  int last_nccl = 0;
};

static_assert(VSF_COMM_ID_BYTES == sizeof(vsfnccl::UniqueId), "the id is RCCL's ncclUniqueId");

// A failed RCCL call is reported as VSF_ERR_HIP with vsf_last_hip_error(ctx) = VSF_RCCL_ERROR_BASE + the ncclResult_t
static vsf_status rccl_failed(vsf_ctx* ctx, vsf_comm* comm, int e) {
  if (comm) comm->last_nccl = e;
  vsf_ctx_set_last_error(ctx, VSF_RCCL_ERROR_BASE + e);
  return VSF_ERR_HIP;
}

extern "C" {

vsf_status vsf_comm_unique_id(uint8_t* id) {
  if (!id) return VSF_ERR_INVALID_ARG;
  Rccl& r = rccl();
  if (!r.ok) return VSF_ERR_UNSUPPORTED;
  vsfnccl::UniqueId u;
  if (r.GetUniqueId(&u) != vsfnccl::Success) return VSF_ERR_HIP;
  std::memcpy(id, u.internal, VSF_COMM_ID_BYTES);
  return VSF_OK;
}

vsf_status vsf_comm_create(vsf_ctx* ctx, const uint8_t* id, int rank, int world, vsf_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return VSF_ERR_INVALID_ARG;
  VsfErrorScope scope_(ctx);
  *out = nullptr;
  Rccl& r = rccl();
  if (!r.ok) return VSF_ERR_UNSUPPORTED;
  vsf_comm* c = new (std::nothrow) vsf_comm();
  if (!c) return VSF_ERR_INVALID_ARG;
  c->rank = rank;
  c->world = world;
  c->device = vsf_ctx_device(ctx);
  const hipError_t he = hipSetDevice(c->device);
  if (he != hipSuccess) {
    delete c;
    vsf_ctx_set_last_error(ctx, (int)he);
    return VSF_ERR_HIP;
  }
  vsfnccl::UniqueId u;
  std::memcpy(u.internal, id, VSF_COMM_ID_BYTES);
  const vsfnccl::Result_t e = r.CommInitRank(&c->comm, world, u, rank);  // (collective: every rank of the world calls it)
  if (e != vsfnccl::Success) {
    delete c;
    return rccl_failed(ctx, nullptr, e);
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
  VsfErrorScope scope_(ctx);
  const hipError_t he = hipSetDevice(comm->device);
  if (he != hipSuccess) {
    vsf_ctx_set_last_error(ctx, (int)he);
    return VSF_ERR_HIP;
  }
  const vsfnccl::Result_t e =
      rccl().AllGather(d_send, d_recv, bytes_per_rank, vsfnccl::Uint8, comm->comm, vsf_ctx_stream(ctx));
  comm->last_nccl = (int)e;
  return e == vsfnccl::Success ? VSF_OK : rccl_failed(ctx, comm, e);
}

vsf_status vsf_gather_payload_dev(vsf_ctx* ctx, vsf_comm* comm, const uint8_t* d_send, size_t bytes, uint8_t* d_recv,
                                  size_t recv_stride, int root) {
  if (!ctx || !comm || !d_send || bytes == 0 || root < 0 || root >= comm->world) return VSF_ERR_INVALID_ARG;
  if (comm->rank == root && (!d_recv || recv_stride < bytes)) return VSF_ERR_INVALID_ARG;
  VsfErrorScope scope_(ctx);
  const hipError_t he = hipSetDevice(comm->device);
  if (he != hipSuccess) {
    vsf_ctx_set_last_error(ctx, (int)he);
    return VSF_ERR_HIP;
  }
  Rccl& r = rccl();
  hipStream_t s = vsf_ctx_stream(ctx);
  // One group of point-to-point transfers: every rank sends `bytes` to the root, the root posts one receive per rank
  // (its own included).  On a fully connected xGMI node each peer -> root transfer rides its own link.
  vsfnccl::Result_t e = r.GroupStart();
  if (e == vsfnccl::Success && comm->rank == root)
    for (int p = 0; p < comm->world && e == vsfnccl::Success; p++)
      e = r.Recv(d_recv + (size_t)p * recv_stride, bytes, vsfnccl::Uint8, p, comm->comm, s);
  if (e == vsfnccl::Success) e = r.Send(d_send, bytes, vsfnccl::Uint8, root, comm->comm, s);
  const vsfnccl::Result_t e2 = r.GroupEnd();
  if (e == vsfnccl::Success) e = e2;
  comm->last_nccl = (int)e;
  return e == vsfnccl::Success ? VSF_OK : rccl_failed(ctx, comm, e);
}

}  // extern "C"
