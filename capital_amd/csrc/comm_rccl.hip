// comm_rccl.hip -- the reference's MPI collectives (C1-C10) as RCCL calls on the handle's HIP
// stream.  RCCL is bound at run time with dlopen so that (a) single-GPU use never loads it and (b) inside a
// torch.distributed process the SAME librccl.so.1 that torch already mapped is reused (one RCCL per process).
// One process per GPU; communicators of size 1 short-circuit every call -- unless CAPI_RCCL_FORCE is set, which sends
// even a 1-rank communicator through the real library (the only way to exercise these wrappers on a 1-GPU box).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "capi_internal.h"
#include "pair_paths.h"

struct capi_comm_s {
  ncclComm_t comm = nullptr;
  capi_handle_t h = nullptr;
  int rank = 0, size = 1;
};

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

char g_rccl_err[256] = {0};

int rccl_bind(const char* path) {
  if (g_rccl.lib) return CAPI_OK;
  const char* candidates[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
  void* lib = nullptr;
  for (int i = 0; i < 5 && !lib; ++i) {
    if (!candidates[i]) { if (i == 0) continue; else break; }
    // prefer a copy that is already mapped into this process (torch's), then load by name
    lib = dlopen(candidates[i], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!lib) lib = dlopen(candidates[i], RTLD_NOW | RTLD_GLOBAL);
  }
  if (!lib) { snprintf(g_rccl_err, sizeof(g_rccl_err), "cannot load librccl: %s", dlerror()); return CAPI_ECOMM; }
#define BIND(field, sym)                                                                       \
  *(void**)(&g_rccl.field) = dlsym(lib, sym);                                                  \
  if (!g_rccl.field) { snprintf(g_rccl_err, sizeof(g_rccl_err), "librccl lacks %s", sym); return CAPI_ECOMM; }
  BIND(GetUniqueId, "ncclGetUniqueId")
  BIND(CommInitRank, "ncclCommInitRank")
  BIND(CommSplit, "ncclCommSplit")
  BIND(CommDestroy, "ncclCommDestroy")
  BIND(CommUserRank, "ncclCommUserRank")
  BIND(CommCount, "ncclCommCount")
  BIND(Broadcast, "ncclBroadcast")
  BIND(AllReduce, "ncclAllReduce")
  BIND(Reduce, "ncclReduce")
  BIND(AllGather, "ncclAllGather")
  BIND(Send, "ncclSend")
  BIND(Recv, "ncclRecv")
  BIND(GroupStart, "ncclGroupStart")
  BIND(GroupEnd, "ncclGroupEnd")
  BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
  g_rccl.lib = lib;
  return CAPI_OK;
}

#define NCCL_CHECK(c, call)                                                                      \
  do {                                                                                           \
    ncclResult_t r__ = (call);                                                                   \
    if (r__ != ncclSuccess) {                                                                    \
      if ((c) && (c)->h) snprintf((c)->h->err, sizeof((c)->h->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r__)); \
      return CAPI_ECOMM;                                                                         \
    }                                                                                            \
  } while (0)

}  // namespace

extern "C" {

int capi_comm_load_rccl(const char* path) { return rccl_bind(path); }

int capi_comm_unique_id(void* id128) {
  if (!id128) return CAPI_EINVAL;
  int rc = rccl_bind(nullptr);
  if (rc != CAPI_OK) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return CAPI_ECOMM;
  memcpy(id128, &id, sizeof(id));
  return CAPI_OK;
}

int capi_comm_init_rank(capi_comm_t* out, capi_handle_t h, int nranks, const void* id128, int rank) {
  CAPI_REQUIRE(h, h && out && nranks >= 1 && rank >= 0 && rank < nranks, "args");
  *out = nullptr;
  capi_comm_s* c = new capi_comm_s();
  c->h = h;
  c->rank = rank;
  c->size = nranks;
  // every failure below releases the object: a caller that retries must not accumulate half-built communicators
  auto fail = [&](int rc) { delete c; return rc; };
  if (nranks > 1 || getenv("CAPI_RCCL_FORCE")) {
    if (!id128) { snprintf(h->err, sizeof(h->err), "capi_comm_init_rank: unique id missing"); return fail(CAPI_EINVAL); }
    int rc = rccl_bind(nullptr);
    if (rc != CAPI_OK) { snprintf(h->err, sizeof(h->err), "%s", g_rccl_err); return fail(rc); }
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) { snprintf(h->err, sizeof(h->err), "hipSetDevice -> %s", hipGetErrorString(e)); return fail(CAPI_EHIP); }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { snprintf(h->err, sizeof(h->err), "ncclCommInitRank -> %s", g_rccl.GetErrorString(r)); return fail(CAPI_ECOMM); }
  }
  *out = c;
  return CAPI_OK;
}

// MPI_Comm_split.  A negative colour is MPI_UNDEFINED / NCCL_SPLIT_NOCOLOR: the rank takes part in the call (it is
// collective over the parent) and receives no communicator -- *child = NULL with status OK.
int capi_comm_split(capi_comm_t parent, int color, int key, capi_comm_t* child) {
  if (!parent || !child) return CAPI_EINVAL;
  *child = nullptr;
  if (!parent->comm) {
    if (color < 0) return CAPI_OK;
    capi_comm_s* c = new capi_comm_s();
    c->h = parent->h;
    c->rank = 0;
    c->size = 1;
    *child = c;
    return CAPI_OK;
  }
  ncclComm_t sub = nullptr;
  NCCL_CHECK(parent, g_rccl.CommSplit(parent->comm, color < 0 ? NCCL_SPLIT_NOCOLOR : color, key, &sub, nullptr));
  if (color < 0 || !sub) return CAPI_OK;
  capi_comm_s* c = new capi_comm_s();
  c->h = parent->h;
  c->comm = sub;
  // rank/size of the child: RCCL orders by key then parent rank; recover them from the library
  if (g_rccl.CommUserRank(sub, &c->rank) != ncclSuccess || g_rccl.CommCount(sub, &c->size) != ncclSuccess) {
    snprintf(parent->h->err, sizeof(parent->h->err), "capi_comm_split: ncclCommUserRank/ncclCommCount failed on the child");
    g_rccl.CommDestroy(sub);
    delete c;
    return CAPI_ECOMM;
  }
  *child = c;
  return CAPI_OK;
}

int capi_comm_rank(capi_comm_t c, int* rank) { if (!c || !rank) return CAPI_EINVAL; *rank = c->rank; return CAPI_OK; }
int capi_comm_size(capi_comm_t c, int* size) { if (!c || !size) return CAPI_EINVAL; *size = c->size; return CAPI_OK; }

// what the LIBRARY reports for this communicator (a launcher prints it next to its own idea of rank and size);
// a communicator of one rank that never touched RCCL reports itself
int capi_comm_query(capi_comm_t c, int* rank, int* size) {
  if (!c || !rank || !size) return CAPI_EINVAL;
  *rank = c->rank; *size = c->size;
  if (c->comm) {
    NCCL_CHECK(c, g_rccl.CommUserRank(c->comm, rank));
    NCCL_CHECK(c, g_rccl.CommCount(c->comm, size));
  }
  return CAPI_OK;
}

int capi_comm_destroy(capi_comm_t c) {
  if (!c) return CAPI_EINVAL;
  if (c->comm) {
    // collectives of this communicator may be in flight on ANY of the handle's streams (the chunk pipeline issues them on
    // stream index 1): all of them are drained before the communicator goes away
    (void)capi_sync(c->h);
    g_rccl.CommDestroy(c->comm);
  }
  delete c;
  return CAPI_OK;
}

int capi_bcast(capi_comm_t c, double* buf, int64_t count, int root) {
  if (!c || count < 0 || root < 0 || root >= c->size) return CAPI_EINVAL;
  if (!c->comm || count == 0) return CAPI_OK;
  NCCL_CHECK(c, g_rccl.Broadcast(buf, buf, (size_t)count, ncclDouble, root, c->comm, c->h->stream));
  return CAPI_OK;
}

int capi_allreduce_sum(capi_comm_t c, double* buf, int64_t count) {
  if (!c || count < 0) return CAPI_EINVAL;
  if (!c->comm || count == 0) return CAPI_OK;
  NCCL_CHECK(c, g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->comm, c->h->stream));
  return CAPI_OK;
}

int capi_reduce_sum(capi_comm_t c, const double* send, double* recv, int64_t count, int root) {
  if (!c || count < 0 || root < 0 || root >= c->size) return CAPI_EINVAL;
  if (count == 0) return CAPI_OK;
  if (!c->comm) {
    if (send != recv) CAPI_HIP_CHECK(c->h, hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c->h->stream));
    return CAPI_OK;
  }
  NCCL_CHECK(c, g_rccl.Reduce(send, recv, (size_t)count, ncclDouble, ncclSum, root, c->comm, c->h->stream));
  return CAPI_OK;
}

int capi_allgather(capi_comm_t c, const double* send, double* recv, int64_t count) {
  if (!c || count < 0) return CAPI_EINVAL;
  if (count == 0) return CAPI_OK;
  if (!c->comm) {
    if (send != recv) CAPI_HIP_CHECK(c->h, hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c->h->stream));
    return CAPI_OK;
  }
  NCCL_CHECK(c, g_rccl.AllGather(send, recv, (size_t)count, ncclDouble, c->comm, c->h->stream));
  return CAPI_OK;
}

// A group that was opened is always closed: the first failing call inside it is remembered and reported after GroupEnd.
#define NCCL_IN_GROUP(c, first, call)                                                            \
  do {                                                                                           \
    ncclResult_t r__ = (call);                                                                   \
    if (r__ != ncclSuccess && (first) == ncclSuccess) (first) = r__;                             \
  } while (0)
static int group_result(capi_comm_t c, ncclResult_t first, const char* what) {
  ncclResult_t e = g_rccl.GroupEnd();
  if (first == ncclSuccess) first = e;
  if (first != ncclSuccess) {
    snprintf(c->h->err, sizeof(c->h->err), "%s -> %s", what, g_rccl.GetErrorString(first));
    return CAPI_ECOMM;
  }
  return CAPI_OK;
}

// MPI_Gather / MPI_Scatter of `count` doubles per rank (cholinv/policy.h:322-332,361-377): RCCL has neither, both are one
// group of point-to-point calls -- on the xGMI mesh every peer of the root is a link of its own.
int capi_gather(capi_comm_t c, const double* send, double* recv, int64_t count, int root) {
  if (!c || count < 0 || root < 0 || root >= c->size) return CAPI_EINVAL;
  if (count == 0) return CAPI_OK;
  if (!c->comm) {
    if (send != recv) CAPI_HIP_CHECK(c->h, hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c->h->stream));
    return CAPI_OK;
  }
  NCCL_CHECK(c, g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  NCCL_IN_GROUP(c, first, g_rccl.Send(send, (size_t)count, ncclDouble, root, c->comm, c->h->stream));
  if (c->rank == root)
    for (int r = 0; r < c->size; ++r) NCCL_IN_GROUP(c, first, g_rccl.Recv(recv + (int64_t)r * count, (size_t)count, ncclDouble, r, c->comm, c->h->stream));
  return group_result(c, first, "capi_gather");
}

int capi_scatter(capi_comm_t c, const double* send, double* recv, int64_t count, int root) {
  if (!c || count < 0 || root < 0 || root >= c->size) return CAPI_EINVAL;
  if (count == 0) return CAPI_OK;
  if (!c->comm) {
    if (send != recv) CAPI_HIP_CHECK(c->h, hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c->h->stream));
    return CAPI_OK;
  }
  NCCL_CHECK(c, g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  if (c->rank == root)
    for (int r = 0; r < c->size; ++r) NCCL_IN_GROUP(c, first, g_rccl.Send(send + (int64_t)r * count, (size_t)count, ncclDouble, r, c->comm, c->h->stream));
  NCCL_IN_GROUP(c, first, g_rccl.Recv(recv, (size_t)count, ncclDouble, root, c->comm, c->h->stream));
  return group_result(c, first, "capi_scatter");
}

int capi_sendrecv_replace(capi_comm_t c, double* buf, int64_t count, int peer, double* staging) {
  if (!c || count < 0 || peer < 0 || peer >= c->size) return CAPI_EINVAL;
  if (count == 0 || (peer == c->rank && !(c->comm && c->size == 1))) return CAPI_OK;   // (forced 1-rank communicator: a real self send/recv)
  if (!staging) return CAPI_EINVAL;
  NCCL_CHECK(c, g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  NCCL_IN_GROUP(c, first, g_rccl.Send(buf, (size_t)count, ncclDouble, peer, c->comm, c->h->stream));
  NCCL_IN_GROUP(c, first, g_rccl.Recv(staging, (size_t)count, ncclDouble, peer, c->comm, c->h->stream));
  int rc = group_result(c, first, "capi_sendrecv_replace");
  if (rc != CAPI_OK) return rc;
  CAPI_HIP_CHECK(c->h, hipMemcpyAsync(buf, staging, sizeof(double) * count, hipMemcpyDeviceToDevice, c->h->stream));
  return CAPI_OK;
}

// Multi-path pair transfers (pair_paths.h): every rank of `c` -- the WORLD communicator of the node -- calls this with the same `dst`
// array; rank r sends `count` doubles from its `send` to rank dst[r] (or nothing, dst[r] < 0), each rank receives from at most one.
// Messages of at least CAPITAL_MULTIPATH_MIN (default 2^20) doubles are cut into size() units that travel over all links of the mesh
// in two grouped rounds of ncclSend / ncclRecv, relayed through `scratch` (capi_pairs_scratch_count doubles) on the ranks in between;
// shorter ones go directly.  Stream-ordered on the handle's selected stream like every other collective here.
int64_t capi_pairs_scratch_count(int nranks, int64_t count) { return pair_paths::scratch_count(nranks, count); }

int capi_pairs_transfer(capi_comm_t c, const int* dst, const double* send, double* recv, int64_t count, double* scratch) {
  if (!c || !dst || count < 0) return CAPI_EINVAL;
  struct RcclPaths {
    capi_comm_t c;
    ncclResult_t first = ncclSuccess;
    int group_begin() { first = ncclSuccess; return g_rccl.GroupStart() == ncclSuccess ? 0 : CAPI_ECOMM; }
    void send(const double* p, int64_t n, int peer) { NCCL_IN_GROUP(c, first, g_rccl.Send(p, (size_t)n, ncclDouble, peer, c->comm, c->h->stream)); }
    void recv(double* p, int64_t n, int peer) { NCCL_IN_GROUP(c, first, g_rccl.Recv(p, (size_t)n, ncclDouble, peer, c->comm, c->h->stream)); }
    int group_end() { return group_result(c, first, "capi_pairs_transfer"); }
  };
  if (!c->comm) {                                                                  // a lone rank that never touched RCCL: nothing, or a local copy to itself
    if (c->size != 1 || dst[0] > 0) return CAPI_EINVAL;
    if (dst[0] == 0 && count > 0) {
      if (!send || !recv) return CAPI_EINVAL;
      if (send != recv) CAPI_HIP_CHECK(c->h, hipMemcpyAsync(recv, send, sizeof(double) * count, hipMemcpyDeviceToDevice, c->h->stream));
    }
    return CAPI_OK;
  }
  // (read per call: tests and A/B tools change it between cases of one process, and every rank of a transfer must cut its message the same way)
  const char* min_env = getenv("CAPITAL_MULTIPATH_MIN");
  const int64_t min_count = min_env ? atoll(min_env) : ((int64_t)1 << 20);
  RcclPaths x{c};
  const int rc = pair_paths::transfer(x, c->rank, c->size, dst, send, recv, count, scratch, min_count);
  if (rc < 0) { snprintf(c->h->err, sizeof(c->h->err), "capi_pairs_transfer: invalid transfer set, or scratch missing"); return CAPI_EINVAL; }
  return rc;
}

}  // extern "C"
