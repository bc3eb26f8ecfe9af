// loopback_async.hip -- TEST INFRASTRUCTURE (never loaded by the product on its own): the ASYNCHRONOUS mode of the loopback transport.
//
// The host-staged mode (rccl_loopback.cpp) drains the caller's stream around every message, so every call has completed when it returns:
// a legal execution of RCCL's API, and one on which a missing consumer-side wait in the product's chunk pipelines cannot fail.  This
// mode behaves like RCCL does: a call ENQUEUES device work on the caller's stream and returns; nothing here calls hipStreamSynchronize,
// waits for an event on the host, or looks at data.  What the reference gets from MPI_Ibcast / MPI_Iallreduce + MPI_Wait
// (summa.hpp:195-215,238-249) -- transfers that progress beside the local BLAS call -- is then really asynchronous in the rehearsal too.
//
// All ranks sit on ONE GPU in separate processes (ipc_probe.cpp established what the pool's driver offers two such processes):
//   * every ordered pair (src, dst) of a communicator has a CHANNEL: a ring of device memory that src allocates and exports
//     (hipIpcGetMemHandle) and dst maps (hipIpcOpenMemHandle), plus two counters in a host page both registered with HIP;
//   * send  = [wait until the ring region is free again: hipStreamWaitValue32 on `consumed`] -> copy kernel user buffer -> ring ->
//             hipStreamWriteValue32(posted = seq), all on the caller's stream;
//   * recv  = hipStreamWaitValue32(posted >= seq) -> copy (or add) kernel ring -> user buffer -> hipStreamWriteValue32(consumed = seq);
//   * the copy kernels run with a configurable number of workgroups (CAPI_LOOPBACK_COPY_WGS, default 16) -- like RCCL's point-to-point
//     kernels they occupy CUs beside the caller's compute kernels -- and an optional delay in front of every receive-side copy
//     (CAPI_LOOPBACK_DELAY_US) widens the window in which a consumer that did not wait reads stale bytes.
// Both sides of a channel place messages in the ring with the same deterministic rule (sizes match on both sides, as NCCL demands), so
// no descriptor travels; the host only meets its peer when a ring is first created or has to grow (the sender publishes a new
// generation, the receiver maps it).  A ring holds at least four messages of the largest size seen, so a sender only ever waits for
// the consumption of messages at least three back: inside one ncclGroup at most three messages per (communicator, peer, direction).
// A watchdog thread ends the process with a diagnosis when device-side waits stop making progress (CAPI_LOOPBACK_TIMEOUT_S).
// The transport checks ITSELF: every message ends in a trailer (checksum of what the sender put into the ring, length, sequence number, channel identity) that the
// receiver compares with what it finds before it reports the message consumed (CAPI_LOOPBACK_VERIFY=0 turns that off), and every ring ends in a label the receiver reads
// back through its mapping when it opens the handle (CAPI_LOOPBACK_LABEL: 0 skips the label, -1 also its 4096 bytes -- diagnostics).  Why: DESIGN.md section 6 (iv).
// CAPI_LOOPBACK_MIN_RING_MB sets the smallest ring (default 1).
#include "loopback_async.h"
#include "ring_place.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace lb_async {
namespace {

constexpr int MAX_GEN = 24;
struct alignas(256) Channel {
  uint32_t posted;            // device-written: messages whose bytes are in the ring
  uint32_t consumed;          // device-written: messages the receiver has copied out
  uint32_t gen;               // host-written by the sender (release): generation of the ring it has published
  uint32_t pad_;
  uint64_t capacity;          // bytes of that generation
  // device-written by the receiver's check kernel: the first message whose bytes, as the receiver finds them in the ring, do not give the sender's checksum
  uint32_t bad_seq, bad_pad_;
  uint64_t bad_record[8];     // what it saw (lb_check_kernel)
  hipIpcMemHandle_t handle[MAX_GEN];   // one per generation (a ring at least doubles when it grows: 1 MiB .. 2^(20 + MAX_GEN) bytes), so a receiver
                                       // that lags its sender by several growths still maps the generation its message sits in
};
struct alignas(256) Header {
  uint32_t magic, size, bar_count, bar_gen;
};
constexpr uint32_t MAGIC = 0x4c424153u;

double timeout_s() {
  const char* e = getenv("CAPI_LOOPBACK_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}
int copy_wgs() {
  static const int v = [] { const char* e = getenv("CAPI_LOOPBACK_COPY_WGS"); int w = e ? atoi(e) : 16; return w < 1 ? 1 : (w > 1024 ? 1024 : w); }();
  return v;
}
bool verify() {
  static const bool v = [] { const char* e = getenv("CAPI_LOOPBACK_VERIFY"); return !e || atoi(e) != 0; }();      // on unless CAPI_LOOPBACK_VERIFY=0
  return v;
}
long delay_us() {
  static const long v = [] { const char* e = getenv("CAPI_LOOPBACK_DELAY_US"); long d = e ? atol(e) : 0; return d < 0 ? 0 : (d > 1000000 ? 1000000 : d); }();
  return v;
}

// `coherent`: the source is a ring that another rank's kernel has just written -- read it past the caches (system-scope loads).  A peer in another PROCESS is read
// through an IPC mapping; a peer that is a THREAD of this process (tests/thread_ranks) through the very pointer its kernel wrote, whose lines this XCD's L2 may still hold
// from the slot's previous lap: nothing the HIP runtime knows of orders the two kernels (the ordering is the stream wait-value), so no cache maintenance is implied.
__global__ void lb_copy_kernel(char* __restrict__ dst, const char* __restrict__ src, size_t bytes, int accumulate, int coherent) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
  // coherent > 0 (receive side): drop whatever this XCD's caches hold of the ring BEFORE reading it (system-scope acquire: buffer_inv sc0 sc1);
  // coherent < 0 (send side): push the ring's new contents out of this XCD's L2 when done (system-scope release: buffer_wbl2), see the end of the kernel
  if (coherent > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (coherent > 0 && ((((uintptr_t)dst) | ((uintptr_t)src) | bytes) & 7) == 0) {
    unsigned long long* d = (unsigned long long*)dst;
    const unsigned long long* s = (const unsigned long long*)src;
    for (size_t i = tid; i < bytes / 8; i += nthr) {
      const unsigned long long v = __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (accumulate) ((double*)d)[i] = ((double*)d)[i] + __longlong_as_double((long long)v); else d[i] = v;
    }
    return;
  }
  struct Release { int on; __device__ ~Release() { if (on) __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); } } release_at_end{coherent < 0};
  const bool al16 = ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0;
  if (accumulate) {                                   // doubles (the reductions' only type here)
    double* d = (double*)dst;
    const double* s = (const double*)src;
    for (size_t i = tid; i < bytes / 8; i += nthr) d[i] = d[i] + s[i];
    return;
  }
  size_t done = 0;
  if (al16) {
    const size_t n16 = bytes / 16;
    uint4* d = (uint4*)dst;
    const uint4* s = (const uint4*)src;
    for (size_t i = tid; i < n16; i += nthr) d[i] = s[i];
    done = n16 * 16;
  }
  for (size_t i = done + tid; i < bytes; i += nthr) dst[i] = coherent > 0 ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : src[i];
}
// Every message carries a checksum: the sender sums (xor of the 8-byte words, the odd bytes folded in) what it has put into the ring and leaves the sum in the message's
// trailer; the receiver sums what IT finds in the ring, after its copy and before it reports the message consumed, and records a mismatch in the channel's control block
// (the watchdog reports it and ends the process).  A message overwritten too early, read too early or read stale is then an error of the TRANSPORT with a name --
// communicator, pair, sequence number -- instead of a wrong factor somewhere downstream.  One workgroup; these are test messages.
__device__ unsigned long long lb_sum_of(const char* p, size_t bytes, bool coherent) {
  __shared__ unsigned long long part[256];
  unsigned long long x = 0;
  const unsigned long long* w = (const unsigned long long*)p;           // (ring slots are 256-byte aligned)
  for (size_t i = threadIdx.x; i < bytes / 8; i += blockDim.x) x ^= coherent ? __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : w[i];
  for (size_t i = (bytes & ~(size_t)7) + threadIdx.x; i < bytes; i += blockDim.x) x ^= (unsigned long long)(unsigned char)p[i] << (8 * (i & 7));
  part[threadIdx.x] = x;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) { if ((int)threadIdx.x < s_) part[threadIdx.x] ^= part[threadIdx.x + s_]; __syncthreads(); }
  return part[0];
}
// trailer of a message: [0] checksum, [1] length, [2] sequence number on its channel, [3] identity of the channel (hash of the communicator's name, source, destination)
__global__ __launch_bounds__(256) void lb_sum_kernel(const char* slot, size_t bytes, unsigned long long* trailer, unsigned long long seq, unsigned long long ident) {
  const unsigned long long x = lb_sum_of(slot, bytes, false);
  if (threadIdx.x == 0) { trailer[0] = x; trailer[1] = (unsigned long long)bytes; trailer[2] = seq; trailer[3] = ident; __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
}
// On a mismatch the receiver looks a SECOND time, two milliseconds later, and records both views: bytes that have become right were read too early (an ordering
// failure); a trailer that names a later message of this channel was overwritten too early (flow control); one that names another channel, or nothing, is another
// block of memory (the mapping).  record[0..7] = trailer as first seen (4 words), the sum first found, then sum / trailer[0] / trailer[2] of the second look.
__global__ __launch_bounds__(256) void lb_check_kernel(const char* slot, size_t bytes, const unsigned long long* trailer, uint32_t seq, unsigned long long ident, uint32_t* bad_seq,
                                                       unsigned long long* record) {
  __shared__ int mismatch;
  __shared__ unsigned long long seen[4];
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const unsigned long long x = lb_sum_of(slot, bytes, true);
  if (threadIdx.x == 0) {
    for (int i = 0; i < 4; ++i) seen[i] = __hip_atomic_load(trailer + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    mismatch = (seen[0] != x || seen[1] != (unsigned long long)bytes || seen[2] != seq || seen[3] != ident) ? 1 : 0;
  }
  __syncthreads();
  if (!mismatch) return;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 200000) __builtin_amdgcn_s_sleep(64);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
  const unsigned long long x2 = lb_sum_of(slot, bytes, true);
  if (threadIdx.x == 0 && atomicCAS(bad_seq, 0u, seq) == 0u) {
    for (int i = 0; i < 4; ++i) record[i] = seen[i];
    record[4] = x;
    record[5] = x2;
    record[6] = __hip_atomic_load(trailer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    record[7] = __hip_atomic_load(trailer + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// a bounded wait (the wall clock runs at 100 MHz): leaves on its own, whatever happens elsewhere
__global__ void lb_delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

struct Registry;
Registry& registry();
uint64_t ident_of(const std::string& path, int src, int dst) {
  uint64_t h = 1469598103934665603ull;
  for (char ch : path) { h ^= (unsigned char)ch; h *= 1099511628211ull; }
  return (h << 16) ^ ((uint64_t)(src & 0xff) << 8) ^ (uint64_t)(dst & 0xff) ^ 0x4c42000000000000ull;
}

}  // namespace

struct AComm {
  std::string path;
  int rank = 0, size = 1;
  size_t map_bytes = 0;
  char* map = nullptr;          // host view of the control segment
  char* map_dev = nullptr;      // its device alias (hipHostGetDevicePointer)
  bool registered = false;
  std::vector<Ring> tx, rx;     // per peer
  std::vector<hipStream_t> streams;   // every stream this rank has used with the communicator (detach drains THESE, never the device)
  std::vector<void*> old_scr;
  void saw(hipStream_t s) { for (hipStream_t t : streams) if (t == s) return; streams.push_back(s); }
  void* scr = nullptr;
  size_t scr_bytes = 0;
  std::mutex mu;
  Header* hdr() const { return (Header*)map; }
  Channel* ch(int src, int dst) const { return (Channel*)(map + sizeof(Header)) + (size_t)src * size + dst; }
  template <typename T> T* dev(T* host_ptr) const { return (T*)(map_dev + ((char*)host_ptr - map)); }
};

namespace {

struct Registry {
  std::mutex mu;
  std::vector<AComm*> comms;
  // ranks of one communicator that live in THIS process (the ranks-as-threads rehearsal): the same device memory must not be IPC-opened
  // by the process that exported it, so a receiver takes the sender's pointer instead
  AComm* local_peer(const std::string& path, int rank) {
    std::lock_guard<std::mutex> g(mu);
    for (AComm* c : comms) if (c->path == path && c->rank == rank) return c;
    return nullptr;
  }
  bool started = false;
  void watch() {
    uint64_t last = ~0ull;
    auto since = std::chrono::steady_clock::now();
    for (;;) {
      std::this_thread::sleep_for(std::chrono::milliseconds(500));
      uint64_t sum = 0;
      bool outstanding = false;
      std::string what;
      {
        std::lock_guard<std::mutex> g(mu);
        for (AComm* c : comms) {
          std::lock_guard<std::mutex> gc(c->mu);
          for (int p = 0; p < c->size; ++p) {
            if (p == c->rank) continue;
            const uint32_t posted = __atomic_load_n(&c->ch(p, c->rank)->posted, __ATOMIC_RELAXED), mine = __atomic_load_n(&c->ch(c->rank, p)->posted, __ATOMIC_RELAXED);
            const uint32_t consumed = __atomic_load_n(&c->ch(c->rank, p)->consumed, __ATOMIC_RELAXED), eaten = __atomic_load_n(&c->ch(p, c->rank)->consumed, __ATOMIC_RELAXED);
            sum += (uint64_t)posted + mine + consumed + eaten;
            const uint32_t bad = __atomic_load_n(&c->ch(p, c->rank)->bad_seq, __ATOMIC_RELAXED);
            if (bad) {
              const Channel* ch_ = c->ch(p, c->rank);
              const uint64_t* w = ch_->bad_record;
              fprintf(stderr, "rccl_loopback (async): DATA CHECK FAILED on %s: message %u from rank %d to rank %d (channel identity %016llx; receives issued %u, posted %u, copied out %u; the sender is %s)\n"
                      "    the trailer in the ring read: checksum %016llx, length %llu, sequence %llu, identity %016llx; the receiver's sum over the slot: %016llx\n"
                      "    two milliseconds later: sum %016llx, trailer checksum %016llx, sequence %llu  =>  %s\n", c->path.c_str(), bad, p, c->rank,
                      (unsigned long long)ident_of(c->path, p, c->rank), c->rx[p].seq, posted, eaten, c->rx[p].base_local ? "a thread of this process" : "another process",
                      (unsigned long long)w[0], (unsigned long long)w[1], (unsigned long long)w[2], (unsigned long long)w[3], (unsigned long long)w[4], (unsigned long long)w[5],
                      (unsigned long long)w[6], (unsigned long long)w[7],
                      (w[5] == w[6] && w[7] == bad) ? "the bytes ARRIVED LATER: the receiver read before the sender had written (ordering)"
                      : (w[3] == ident_of(c->path, p, c->rank) && w[2] > bad) ? "a LATER message of this channel lies there: overwritten before it was consumed (flow control)"
                      : (w[3] == ident_of(c->path, p, c->rank)) ? "this channel's trailer, other bytes"
                      : "not a trailer of this channel: the receiver looks at other memory than the sender wrote (the mapping), or the slot was never written");
              fflush(stderr);
              _exit(87);
            }
            if (c->rx[p].seq > eaten || c->tx[p].seq > mine) {        // every channel with work of THIS rank still queued on the device
              char line[384];
              snprintf(line, sizeof(line), "\n    %s rank %d <-> %d: receives issued %u, peer posted %u, copied out %u; sends issued %u, posted %u, peer consumed %u",
                       c->path.c_str(), c->rank, p, c->rx[p].seq, posted, eaten, c->tx[p].seq, mine, consumed);
              if (what.size() < 6000) what += line;
              outstanding = true;
            }
          }
        }
      }
      const auto now = std::chrono::steady_clock::now();
      if (!outstanding || sum != last) { last = sum; since = now; continue; }
      if (std::chrono::duration<double>(now - since).count() > timeout_s()) {
        fprintf(stderr, "rccl_loopback (async): no device-side progress for %.0f s; channels with queued work:%s\n", timeout_s(), what.c_str());
        fflush(stderr);
        _exit(86);
      }
    }
  }
};
Registry& registry() { static Registry* r = new Registry(); return *r; }

bool host_barrier(AComm* c) {
  Header* h = c->hdr();
  const uint32_t g = __atomic_load_n(&h->bar_gen, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&h->bar_count, 1, __ATOMIC_ACQ_REL) == (uint32_t)c->size) {
    __atomic_store_n(&h->bar_count, 0, __ATOMIC_RELEASE);
    __atomic_add_fetch(&h->bar_gen, 1, __ATOMIC_ACQ_REL);
    return true;
  }
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
  while (__atomic_load_n(&h->bar_gen, __ATOMIC_ACQUIRE) == g) {
    if (std::chrono::steady_clock::now() > t_end) { fprintf(stderr, "rccl_loopback (async): rank %d of %s waited too long at a barrier\n", c->rank, c->path.c_str()); return false; }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return true;
}

#define LB_HIP(call)                                                                                                             \
  do {                                                                                                                           \
    hipError_t e__ = (call);                                                                                                     \
    if (e__ != hipSuccess) { fprintf(stderr, "rccl_loopback (async): %s:%d %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e__)); return false; } \
  } while (0)

// Every ring ends in a LABEL the sender writes before it publishes the ring (which channel, which generation): a receiver that has opened the handle reads the label back
// through its mapping, so a mapping that does not show the sender's memory is found when it is made, not by a wrong factor later.
struct Label { uint64_t magic, ident, gen, capacity; };
constexpr uint64_t LABEL_MAGIC = 0x4c424c4142454c21ull;
constexpr size_t LABEL_BYTES = 4096;
// diagnostics (profiles/r4_eight_thread_ranks_2x2x2.txt): CAPI_LOOPBACK_LABEL=0 keeps the ring's size (capacity + 4096) and skips the label; =-1 also drops the 4096 bytes
int label_mode() { static const int v = [] { const char* e = getenv("CAPI_LOOPBACK_LABEL"); return e ? atoi(e) : 1; }(); return v; }

// export and import of ring memory, one call at a time per process (ranks as threads of one process would otherwise run them concurrently)
std::mutex& ipc_mu() { static std::mutex* m = new std::mutex(); return *m; }

bool launch_copy(void* dst, const void* src, size_t bytes, hipStream_t s, bool accumulate, int coherent = 0) {
  if (bytes == 0) return true;
  hipLaunchKernelGGL(lb_copy_kernel, dim3((unsigned)copy_wgs()), dim3(256), 0, s, (char*)dst, (const char*)src, bytes, accumulate ? 1 : 0, coherent);
  LB_HIP(hipGetLastError());
  return true;
}

}  // namespace

AComm* attach(const std::string& dir, const std::string& name, int rank, int size) {
  AComm* c = new AComm();
  c->path = dir + "/" + name + ".ctl";
  c->rank = rank;
  c->size = size;
  c->map_bytes = sizeof(Header) + sizeof(Channel) * (size_t)size * size;
  c->tx.resize(size);
  c->rx.resize(size);
  auto fail = [&](const char* what) { fprintf(stderr, "rccl_loopback (async): %s: %s\n", c->path.c_str(), what); delete c; return (AComm*)nullptr; };
  int fd = -1;
  if (rank == 0) {
    const std::string tmp = c->path + ".tmp";
    fd = open(tmp.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return fail("cannot create the control segment");
    Header h0{MAGIC, (uint32_t)size, 0, 0};
    if (pwrite(fd, &h0, sizeof(h0), 0) != (ssize_t)sizeof(h0) || rename(tmp.c_str(), c->path.c_str()) != 0) { close(fd); return fail("cannot publish the control segment"); }
  } else {
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
    while ((fd = open(c->path.c_str(), O_RDWR)) < 0) {
      if (std::chrono::steady_clock::now() > t_end) return fail("the control segment never appeared");
      std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
  }
  c->map = (char*)mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->map == MAP_FAILED) { c->map = nullptr; return fail("mmap failed"); }
  if (c->hdr()->magic != MAGIC || c->hdr()->size != (uint32_t)size) return fail("control segment of another communicator");
  if (hipHostRegister(c->map, c->map_bytes, hipHostRegisterMapped) != hipSuccess || hipHostGetDevicePointer((void**)&c->map_dev, c->map, 0) != hipSuccess)
    return fail("hipHostRegister of the control segment failed");
  c->registered = true;
  if (!host_barrier(c)) return fail("peers missing");            // everybody has the page registered before the first counter is written
  if (rank == 0) unlink(c->path.c_str());                           // the mappings keep it alive; nothing is left behind in /dev/shm
  Registry& R = registry();
  std::lock_guard<std::mutex> g(R.mu);
  R.comms.push_back(c);
  if (!R.started) { R.started = true; std::thread([&R] { R.watch(); }).detach(); }
  return c;
}

void detach(AComm* c) {
  if (!c) return;
  {
    Registry& R = registry();
    std::lock_guard<std::mutex> g(R.mu);
    for (size_t i = 0; i < R.comms.size(); ++i) if (R.comms[i] == c) { R.comms.erase(R.comms.begin() + i); break; }
  }
  // (ncclCommDestroy is a synchronising call in RCCL too.)  The streams this rank used with the communicator are drained -- NOT the device: with
  // several ranks as threads of one process (tests/thread_ranks) a device-wide wait would also wait for a sibling rank's stream, which may sit in a
  // device-side wait for a message this rank has not enqueued yet.  For the same reason nothing is hipFree'd there (hipFree drains the device):
  // CAPI_LOOPBACK_NO_FREE leaves the rings to the end of the process.
  for (hipStream_t s_ : c->streams) (void)hipStreamSynchronize(s_);
  static const bool no_free = getenv("CAPI_LOOPBACK_NO_FREE") != nullptr;
  const bool met = host_barrier(c);                                  // every rank's transfers have completed
  for (Ring& r : c->rx) {
    if (r.base && !r.base_local) (void)hipIpcCloseMemHandle(r.base);
    for (char* p : r.retired) (void)hipIpcCloseMemHandle(p);
  }
  if (met) (void)host_barrier(c);                                    // every mapping is closed before its owner frees it
  if (!no_free) {
    for (Ring& r : c->tx) {
      if (r.base) (void)hipFree(r.base);
      for (char* p : r.retired) (void)hipFree(p);
    }
    if (c->scr) (void)hipFree(c->scr);
    for (void* p : c->old_scr) (void)hipFree(p);
  }
  if (c->registered) (void)hipHostUnregister(c->map);
  if (c->map) munmap(c->map, c->map_bytes);
  delete c;
}

void* scratch(AComm* c, size_t bytes) {
  if (bytes > c->scr_bytes) {
    // the old block may still be read by work in flight: it is released with the communicator (rare: sizes repeat)
    if (c->scr) { c->old_scr.push_back(c->scr); c->scr = nullptr; c->scr_bytes = 0; }
    size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    if (hipMalloc(&c->scr, want) != hipSuccess) return nullptr;
    c->scr_bytes = want;
  }
  return c->scr;
}

bool local_copy(void* dst, const void* src, size_t bytes, hipStream_t s, bool accumulate) {
  if (dst == src && !accumulate) return true;
  return launch_copy(dst, src, bytes, s, accumulate);
}

bool send(AComm* c, int peer, const void* buf, size_t bytes, hipStream_t s) {
  std::lock_guard<std::mutex> g(c->mu);
  c->saw(s);
  Ring& r = c->tx[peer];
  Channel* ch = c->ch(c->rank, peer);
  uint32_t wait_seq = 0;
  bool grew = false;
  const size_t off = ring_place(r, bytes, wait_seq, grew);
  if (grew) {
    if (r.base) r.retired.push_back(r.base);
    r.base = nullptr;
    if (r.gen >= (uint32_t)MAX_GEN) { fprintf(stderr, "rccl_loopback (async): ring generations exhausted\n"); return false; }
    {
      std::lock_guard<std::mutex> gi(ipc_mu());
      LB_HIP(hipMalloc((void**)&r.base, r.capacity + (label_mode() >= 0 ? LABEL_BYTES : 0)));
      const Label lab{LABEL_MAGIC, ident_of(c->path, c->rank, peer), r.gen, r.capacity};
      if (label_mode() > 0) LB_HIP(hipMemcpy(r.base + r.capacity, &lab, sizeof(lab), hipMemcpyHostToDevice));          // (synchronous: in place before the handle exists)
      LB_HIP(hipIpcGetMemHandle(&ch->handle[r.gen], r.base));
    }
    r.by_gen.resize(r.gen + 1, nullptr);
    r.by_gen[r.gen] = r.base;
    ch->capacity = r.capacity;
    __atomic_store_n(&ch->gen, r.gen, __ATOMIC_RELEASE);
  }
  if (wait_seq) LB_HIP(hipStreamWaitValue32(s, c->dev(&ch->consumed), wait_seq, hipStreamWaitValueGte, 0xffffffffu));
  if (!launch_copy(r.base + off, buf, bytes, s, false, /*coherent: release at the end*/ -1)) return false;
  if (verify()) {
    hipLaunchKernelGGL(lb_sum_kernel, dim3(1), dim3(256), 0, s, r.base + off, bytes, (unsigned long long*)(r.base + r.live.back().end - 256), (unsigned long long)r.seq,
                       (unsigned long long)ident_of(c->path, c->rank, peer));
    LB_HIP(hipGetLastError());
  }
  LB_HIP(hipStreamWriteValue32(s, c->dev(&ch->posted), r.seq, 0));
  return true;
}

bool recv(AComm* c, int peer, void* buf, size_t bytes, hipStream_t s, bool accumulate) {
  std::unique_lock<std::mutex> g(c->mu);
  c->saw(s);
  Ring& r = c->rx[peer];
  Channel* ch = c->ch(peer, c->rank);
  uint32_t wait_seq = 0;
  bool grew = false;
  const size_t off = ring_place(r, bytes, wait_seq, grew);
  const size_t trailer_off = r.live.back().end - 256;
  const uint32_t seq = r.seq, gen = r.gen;
  if (grew) {
    // the one place where a host meets its peer's HOST: the sender has to have published this generation of the ring
    g.unlock();
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
    while (__atomic_load_n(&ch->gen, __ATOMIC_ACQUIRE) < gen) {
      if (std::chrono::steady_clock::now() > t_end) { fprintf(stderr, "rccl_loopback (async): rank %d of %s: peer %d never published ring generation %u\n", c->rank, c->path.c_str(), peer, gen); return false; }
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    char* local_base = nullptr;
    if (AComm* pc = registry().local_peer(c->path, peer)) {          // the sender is a thread of this process
      std::lock_guard<std::mutex> gp(pc->mu);
      const Ring& t = pc->tx[c->rank];
      if (gen < t.by_gen.size()) local_base = t.by_gen[gen];
      if (!local_base) { fprintf(stderr, "rccl_loopback (async): same-process peer %d has no ring generation %u\n", peer, gen); return false; }
    }
    g.lock();
    if (r.base && !r.base_local) r.retired.push_back(r.base);
    r.base = nullptr;
    if (gen >= (uint32_t)MAX_GEN) return false;
    if (local_base) { r.base = local_base; r.base_local = true; }
    else {
      std::lock_guard<std::mutex> gi(ipc_mu());
      const Label want{LABEL_MAGIC, ident_of(c->path, peer, c->rank), gen, r.capacity};
      for (int attempt = 1;; ++attempt) {
        LB_HIP(hipIpcOpenMemHandle((void**)&r.base, ch->handle[gen], hipIpcMemLazyEnablePeerAccess));
        if (label_mode() <= 0) break;
        Label got{};
        LB_HIP(hipMemcpy(&got, r.base + r.capacity, sizeof(got), hipMemcpyDeviceToHost));
        if (got.magic == want.magic && got.ident == want.ident && got.gen == want.gen && got.capacity == want.capacity) break;
        fprintf(stderr, "rccl_loopback (async): %s: rank %d opened ring generation %u of peer %d (%zu bytes) and finds the label {%016llx %016llx gen %llu capacity %llu} where the sender wrote "
                "{%016llx %016llx gen %llu capacity %llu}: the mapping at %p does not show the sender's memory (attempt %d)\n", c->path.c_str(), c->rank, gen, peer, r.capacity,
                (unsigned long long)got.magic, (unsigned long long)got.ident, (unsigned long long)got.gen, (unsigned long long)got.capacity, (unsigned long long)want.magic,
                (unsigned long long)want.ident, (unsigned long long)want.gen, (unsigned long long)want.capacity, (void*)r.base, attempt);
        fflush(stderr);
        (void)hipIpcCloseMemHandle(r.base);
        r.base = nullptr;
        if (attempt == 5) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
      r.base_local = false;
    }
  }
  LB_HIP(hipStreamWaitValue32(s, c->dev(&ch->posted), seq, hipStreamWaitValueGte, 0xffffffffu));
  if (delay_us() > 0) { hipLaunchKernelGGL(lb_delay_kernel, dim3(1), dim3(1), 0, s, (long long)delay_us() * 100); LB_HIP(hipGetLastError()); }
  if (!launch_copy(buf, r.base + off, bytes, s, accumulate, /*coherent: acquire + system-scope loads*/ 1)) return false;
  if (verify()) {
    hipLaunchKernelGGL(lb_check_kernel, dim3(1), dim3(256), 0, s, r.base + off, bytes, (const unsigned long long*)(r.base + trailer_off), seq,
                       (unsigned long long)ident_of(c->path, peer, c->rank), c->dev(&ch->bad_seq), (unsigned long long*)c->dev(&ch->bad_record[0]));
    LB_HIP(hipGetLastError());
  }
  LB_HIP(hipStreamWriteValue32(s, c->dev(&ch->consumed), seq, 0));
  return true;
}

}  // namespace lb_async
