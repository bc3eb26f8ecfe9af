// rccl_loopback.cpp -- TEST INFRASTRUCTURE, never loaded by the product on its own: a stand-in for librccl.so with the same
// C entry points (the fifteen that capital_amd/csrc/comm_rccl.hip binds with dlsym), for rehearsing the N > 1 paths of the
// product -- topo::square splits, SUMMA broadcasts / depth all-reduces, base-case gathers and scatters, the partner exchange,
// the chunk pipeline's stream / event discipline -- with N ranks that all sit on ONE GPU.  RCCL proper refuses two ranks on
// one device, and the build pool's boxes have one; the kernels, the packed wire formats and the stream ordering of the
// product are the real ones here, only the transport differs: messages are files in /dev/shm, staged through host memory.
//
// Semantics kept: every call is stream-ordered as RCCL's are (the stream is drained before a send buffer is read and before
// a receive buffer is written, so work queued behind the call sees the data and work queued before it is not overtaken);
// ncclGroupStart/End batches point-to-point calls, sends first; calls are synchronous on the host, which is a legal
// execution of the asynchronous API for any program whose per-communicator call order agrees across ranks (the same
// condition RCCL itself imposes).  A wait that exceeds CAPI_LOOPBACK_TIMEOUT_S (default 120) returns ncclSystemError.
//
// CAPI_LOOPBACK_MODE=async selects the second transport (loopback_async.hip): the same fifteen entry points, but every call only
// ENQUEUES device work on the caller's stream -- copy kernels through IPC-mapped rings, cross-process ordering by stream write / wait
// values -- and returns, as RCCL's do.  The host-staged mode above stays the default and the fallback.
//
// Loaded through the product's own hook: CAPI_RCCL_LIB=<this .so> (capital_amd/driver.py -> capi_comm_load_rccl).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <sys/stat.h>
#include <unistd.h>
#include <map>

#include "loopback_async.h"

namespace {

bool async_mode() {
  static const bool on = [] { const char* e = getenv("CAPI_LOOPBACK_MODE"); return e && strcmp(e, "async") == 0; }();
  return on;
}

struct LComm {
  std::string dir, name;
  int rank = 0, size = 1;
  std::vector<uint64_t> sent, recvd;     // per peer message counters: matching order on both sides is the only protocol
  int splits = 0;
  lb_async::AComm* a = nullptr;          // asynchronous mode: this communicator's channels
};

double timeout_s() {
  const char* e = getenv("CAPI_LOOPBACK_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}

std::string msg_path(const LComm* c, int src, int dst, uint64_t seq) {
  return c->dir + "/" + c->name + "." + std::to_string(src) + "." + std::to_string(dst) + "." + std::to_string((unsigned long long)seq);
}

bool put(LComm* c, int dst, const void* data, size_t bytes) {
  const std::string p = msg_path(c, c->rank, dst, c->sent[dst]++);
  const std::string tmp = p + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok && rename(tmp.c_str(), p.c_str()) == 0;
}

bool get(LComm* c, int src, void* data, size_t bytes) {
  const std::string p = msg_path(c, src, c->rank, c->recvd[src]++);
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
  int spins = 0;
  while (access(p.c_str(), R_OK) != 0) {
    if (std::chrono::steady_clock::now() > t_end) { fprintf(stderr, "rccl_loopback: rank %d of %s waited too long for %s\n", c->rank, c->name.c_str(), p.c_str()); return false; }
    if (++spins < 200) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return false;
  const bool ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
  fclose(f);
  unlink(p.c_str());
  return ok;
}

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

// device <-> host with the stream drained first (see the header)
bool d2h(std::vector<char>& host, const void* dev, size_t bytes, hipStream_t s) {
  host.resize(bytes ? bytes : 1);
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  return bytes == 0 || hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost) == hipSuccess;
}
bool h2d(void* dev, const std::vector<char>& host, size_t bytes, hipStream_t s) {
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  return bytes == 0 || hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
}

struct P2P { bool send; void* buf; size_t bytes; int peer; LComm* c; hipStream_t s; };
thread_local int g_depth = 0;
thread_local std::vector<P2P> g_ops;

// Asynchronous mode: the i-th send and the i-th receive of every (communicator, peer) go out in rounds -- all i-th sends, then all
// i-th receives -- so that no send of a group ever stands in front of the receive its own ring slot is waiting for.
ncclResult_t run_ops_async(std::vector<P2P>& ops) {
  std::map<std::pair<LComm*, int>, int> nth_send, nth_recv;
  std::vector<int> round(ops.size());
  int rounds = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    auto& cnt = ops[i].send ? nth_send : nth_recv;
    round[i] = cnt[{ops[i].c, ops[i].peer}]++;
    rounds = std::max(rounds, round[i] + 1);
  }
  for (int r = 0; r < rounds; ++r) {
    for (int pass = 0; pass < 2; ++pass)
      for (size_t i = 0; i < ops.size(); ++i) {
        P2P& o = ops[i];
        if (round[i] != r || o.send != (pass == 0) || o.peer == o.c->rank) continue;
        const bool ok = o.send ? lb_async::send(o.c->a, o.peer, o.buf, o.bytes, o.s) : lb_async::recv(o.c->a, o.peer, o.buf, o.bytes, o.s, false);
        if (!ok) return ncclSystemError;
      }
    // a rank's messages to itself: the r-th send meets the r-th receive
    for (size_t i = 0; i < ops.size(); ++i) {
      if (round[i] != r || !ops[i].send || ops[i].peer != ops[i].c->rank) continue;
      for (size_t j = 0; j < ops.size(); ++j)
        if (round[j] == r && !ops[j].send && ops[j].c == ops[i].c && ops[j].peer == ops[j].c->rank) {
          if (ops[j].bytes != ops[i].bytes || !lb_async::local_copy(ops[j].buf, ops[i].buf, ops[i].bytes, ops[j].s, false)) return ncclSystemError;
          break;
        }
    }
  }
  return ncclSuccess;
}

ncclResult_t run_ops(std::vector<P2P>& ops) {
  if (!ops.empty() && ops[0].c->a) return run_ops_async(ops);
  std::vector<char> host;
  for (auto& o : ops)
    if (o.send) { if (!d2h(host, o.buf, o.bytes, o.s) || !put(o.c, o.peer, host.data(), o.bytes)) return ncclSystemError; }
  for (auto& o : ops)
    if (!o.send) { host.resize(o.bytes ? o.bytes : 1); if (!get(o.c, o.peer, host.data(), o.bytes) || !h2d(o.buf, host, o.bytes, o.s)) return ncclSystemError; }
  return ncclSuccess;
}

LComm* L(ncclComm_t c) { return reinterpret_cast<LComm*>(c); }

bool barrier(LComm* c) {
  char b = 1;
  if (c->rank == 0) {
    for (int r = 1; r < c->size; ++r) if (!get(c, r, &b, 1)) return false;
    for (int r = 1; r < c->size; ++r) if (!put(c, r, &b, 1)) return false;
    return true;
  }
  return put(c, 0, &b, 1) && get(c, 0, &b, 1);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  uint64_t t = 0;
  FILE* f = fopen("/dev/urandom", "rb");
  if (f) { if (fread(&t, 1, sizeof(t), f) != sizeof(t)) t = 0; fclose(f); }
  if (!t) t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32);
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/dev/shm/capi_loopback_%016llx", (unsigned long long)t);
  if (mkdir(id->internal, 0700) != 0) return ncclSystemError;
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  if (strncmp(id.internal, "/dev/shm/capi_loopback_", 23) != 0) return ncclInvalidArgument;
  LComm* c = new LComm();
  c->dir = id.internal;
  c->name = "w";
  c->rank = rank;
  c->size = nranks;
  c->sent.assign(nranks, 0);
  c->recvd.assign(nranks, 0);
  if (!barrier(c)) { delete c; return ncclSystemError; }
  if (async_mode() && !(c->a = lb_async::attach(c->dir, c->name, rank, nranks))) { delete c; return ncclSystemError; }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t*) {
  LComm* p = L(comm);
  if (!p || !newcomm) return ncclInvalidArgument;
  *newcomm = nullptr;
  const int split_no = p->splits++;
  struct CK { int color, key; } mine{color, key};
  std::vector<CK> all(p->size);
  all[p->rank] = mine;
  for (int r = 0; r < p->size; ++r) if (r != p->rank && !put(p, r, &mine, sizeof(mine))) return ncclSystemError;
  for (int r = 0; r < p->size; ++r) if (r != p->rank && !get(p, r, &all[r], sizeof(CK))) return ncclSystemError;
  if (color == NCCL_SPLIT_NOCOLOR) return ncclSuccess;
  std::vector<int> members;
  for (int r = 0; r < p->size; ++r) if (all[r].color == color) members.push_back(r);
  std::stable_sort(members.begin(), members.end(), [&](int a, int b) { return all[a].key < all[b].key; });
  LComm* c = new LComm();
  c->dir = p->dir;
  c->name = p->name + "_s" + std::to_string(split_no) + "c" + std::to_string(color);
  c->size = (int)members.size();
  c->rank = (int)(std::find(members.begin(), members.end(), p->rank) - members.begin());
  c->sent.assign(c->size, 0);
  c->recvd.assign(c->size, 0);
  if (p->a && !(c->a = lb_async::attach(c->dir, c->name, c->rank, c->size))) { delete c; return ncclSystemError; }
  *newcomm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  LComm* c = L(comm);
  if (!c) return ncclInvalidArgument;
  if (c->a) { lb_async::detach(c->a); c->a = nullptr; }
  if (c->name == "w") {
    // the world communicator goes last: once everybody is here the message directory can go (best effort)
    const bool all_here = barrier(c);
    if (all_here && c->rank == 0) rmdir(c->dir.c_str());
  }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) { if (!comm || !rank) return ncclInvalidArgument; *rank = L(comm)->rank; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { if (!comm || !count) return ncclInvalidArgument; *count = L(comm)->size; return ncclSuccess; }

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t s) {
  LComm* c = L(comm);
  const size_t bytes = count * type_bytes(t);
  if (!c || !type_bytes(t) || root < 0 || root >= c->size) return ncclInvalidArgument;
  if (c->a) {
    if (c->rank == root) {
      for (int r = 0; r < c->size; ++r) if (r != root && !lb_async::send(c->a, r, send, bytes, s)) return ncclSystemError;
      return lb_async::local_copy(recv, send, bytes, s, false) ? ncclSuccess : ncclSystemError;
    }
    return lb_async::recv(c->a, root, recv, bytes, s, false) ? ncclSuccess : ncclSystemError;
  }
  std::vector<char> host;
  if (c->rank == root) {
    if (!d2h(host, send, bytes, s)) return ncclSystemError;
    for (int r = 0; r < c->size; ++r) if (r != root && !put(c, r, host.data(), bytes)) return ncclSystemError;
    if (send != recv && !h2d(recv, host, bytes, s)) return ncclSystemError;
  } else {
    host.resize(bytes ? bytes : 1);
    if (!get(c, root, host.data(), bytes) || !h2d(recv, host, bytes, s)) return ncclSystemError;
  }
  return ncclSuccess;
}

// sums in rank order on one member, so every rank receives the same bits (as a ring all-reduce does)
static ncclResult_t reduce_to(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, int root, bool all, LComm* c, hipStream_t s) {
  if (!c || t != ncclFloat64 || op != ncclSum || root < 0 || root >= c->size) return ncclInvalidArgument;
  const size_t bytes = count * sizeof(double);
  if (c->a) {
    // the same sum, (((x0 + x1) + x2) + ...) on `root`, by receive-side accumulation in rank order; then the broadcast back
    if (c->rank == root) {
      const void* own = send;
      if (send == recv && root != 0) {                       // in place on a root that is not the first summand: its part is set aside first
        void* tmp = lb_async::scratch(c->a, bytes);
        if (!tmp || !lb_async::local_copy(tmp, send, bytes, s, false)) return ncclSystemError;
        own = tmp;
      }
      for (int r = 0; r < c->size; ++r) {
        const bool ok = r == root ? lb_async::local_copy(recv, own, bytes, s, r != 0) : lb_async::recv(c->a, r, recv, bytes, s, r != 0);
        if (!ok) return ncclSystemError;
      }
      if (all) for (int r = 0; r < c->size; ++r) if (r != root && !lb_async::send(c->a, r, recv, bytes, s)) return ncclSystemError;
    } else {
      if (!lb_async::send(c->a, root, send, bytes, s)) return ncclSystemError;
      if (all && !lb_async::recv(c->a, root, recv, bytes, s, false)) return ncclSystemError;
    }
    return ncclSuccess;
  }
  std::vector<char> host;
  if (!d2h(host, send, bytes, s)) return ncclSystemError;
  if (c->rank == root) {
    std::vector<std::vector<char>> parts(c->size);
    for (int r = 0; r < c->size; ++r) {
      if (r == root) continue;
      parts[r].resize(bytes ? bytes : 1);
      if (!get(c, r, parts[r].data(), bytes)) return ncclSystemError;
    }
    parts[root].swap(host);
    std::vector<char> acc(bytes ? bytes : 1, 0);
    double* a = reinterpret_cast<double*>(acc.data());
    for (int r = 0; r < c->size; ++r) {
      const double* p = reinterpret_cast<const double*>(parts[r].data());
      if (r == 0) memcpy(a, p, bytes); else for (size_t i = 0; i < count; ++i) a[i] += p[i];
    }
    if (all) for (int r = 0; r < c->size; ++r) if (r != root && !put(c, r, acc.data(), bytes)) return ncclSystemError;
    if (!h2d(recv, acc, bytes, s)) return ncclSystemError;
  } else {
    if (!put(c, root, host.data(), bytes)) return ncclSystemError;
    if (all) { if (!get(c, root, host.data(), bytes) || !h2d(recv, host, bytes, s)) return ncclSystemError; }
  }
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  return reduce_to(send, recv, count, t, op, 0, true, L(comm), s);
}
ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t s) {
  return reduce_to(send, recv, count, t, op, root, false, L(comm), s);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  LComm* c = L(comm);
  const size_t bytes = count * type_bytes(t);
  if (!c || !type_bytes(t)) return ncclInvalidArgument;
  if (c->a) {
    for (int r = 0; r < c->size; ++r) if (r != c->rank && !lb_async::send(c->a, r, send, bytes, s)) return ncclSystemError;
    for (int r = 0; r < c->size; ++r) {
      char* dst = static_cast<char*>(recv) + (size_t)r * bytes;
      const bool ok = r == c->rank ? lb_async::local_copy(dst, send, bytes, s, false) : lb_async::recv(c->a, r, dst, bytes, s, false);
      if (!ok) return ncclSystemError;
    }
    return ncclSuccess;
  }
  std::vector<char> mine, host;
  if (!d2h(mine, send, bytes, s)) return ncclSystemError;
  for (int r = 0; r < c->size; ++r) if (r != c->rank && !put(c, r, mine.data(), bytes)) return ncclSystemError;
  for (int r = 0; r < c->size; ++r) {
    char* dst = static_cast<char*>(recv) + (size_t)r * bytes;
    if (r == c->rank) { if (!h2d(dst, mine, bytes, s)) return ncclSystemError; continue; }
    host.resize(bytes ? bytes : 1);
    if (!get(c, r, host.data(), bytes) || !h2d(dst, host, bytes, s)) return ncclSystemError;
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<P2P> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  LComm* c = L(comm);
  if (!c || !type_bytes(t) || peer < 0 || peer >= c->size) return ncclInvalidArgument;
  g_ops.push_back({true, const_cast<void*>(buf), count * type_bytes(t), peer, c, s});
  if (g_depth > 0) return ncclSuccess;
  std::vector<P2P> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  LComm* c = L(comm);
  if (!c || !type_bytes(t) || peer < 0 || peer >= c->size) return ncclInvalidArgument;
  g_ops.push_back({false, buf, count * type_bytes(t), peer, c, s});
  if (g_depth > 0) return ncclSuccess;
  std::vector<P2P> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success (loopback)";
    case ncclSystemError: return "loopback transport failed or timed out";
    case ncclInvalidArgument: return "invalid argument (loopback)";
    case ncclInvalidUsage: return "invalid usage (loopback)";
    default: return "error (loopback)";
  }
}

}  // extern "C"
