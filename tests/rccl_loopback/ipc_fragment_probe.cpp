// ipc_fragment_probe.cpp -- TEST INFRASTRUCTURE: does an IPC mapping of a SMALL hipMalloc block show the exporter's bytes?
// The asynchronous loopback transport first cut its rings at 1 MiB.  Blocks of that size are not allocations of their own: the runtime carves them out of
// larger blocks, several to a block, and the IPC handle of such a piece goes through another path than the handle of a whole allocation.  This probe exports
// COUNT blocks of BYTES each from process A (each filled with its own value), opens them in process B -- from two threads at once, as the ranks-as-threads
// rehearsal does -- and has B read every block through its mapping with a kernel; then B closes the even mappings and reads the odd ones again, A allocates
// and exports a second batch while B still holds the first, and B reads that.  Every read that does not return the block's value is reported.
//   usage: ipc_fragment_probe BYTES COUNT        (the parent forks both children before any HIP call)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>

constexpr int MAXB = 256;
struct Shared {
  std::atomic<int> stage[2];
  hipIpcMemHandle_t h[2][MAXB];
  unsigned long long addr[2][MAXB], base[2][MAXB], range[2][MAXB];
};
static Shared* S;
static int me;

#define CK(call)                                                                                     \
  do {                                                                                               \
    hipError_t e__ = (call);                                                                         \
    if (e__ != hipSuccess) { printf("[%d] %s:%d %s -> %s\n", me, __FILE__, __LINE__, #call, hipGetErrorString(e__)); fflush(stdout); _exit(3); } \
  } while (0)

static void arrive(int v) { S->stage[me].store(v); }
static void await_peer(int v) {
  auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(60);
  while (S->stage[1 - me].load() < v) {
    if (std::chrono::steady_clock::now() > t_end) { printf("[%d] peer never reached stage %d\n", me, v); fflush(stdout); _exit(4); }
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
}

__global__ void fill_k(unsigned long long* p, size_t n, unsigned long long v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// counts the words of p that are not v (system-scope loads: nothing orders this kernel after the writer but the host)
__global__ void count_k(const unsigned long long* p, size_t n, unsigned long long v, unsigned long long* wrong, unsigned long long* first_seen) {
  unsigned long long bad = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long w = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (w != v) { ++bad; if (i == 0) *first_seen = w; }
  }
  if (bad) atomicAdd(wrong, bad);
}

static int bad_blocks = 0;
static void check(const char* what, int batch, int i, void* mapped, size_t bytes, hipStream_t s, unsigned long long* d_out) {
  CK(hipMemsetAsync(d_out, 0, 16, s));
  count_k<<<16, 256, 0, s>>>((const unsigned long long*)mapped, bytes / 8, 1000ull * (batch + 1) + i, d_out, d_out + 1);
  unsigned long long out[2];
  CK(hipMemcpyAsync(out, d_out, 16, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  if (out[0]) {
    ++bad_blocks;
    printf("[1] %s: batch %d block %d (exporter's address %#llx, allocation base %#llx + %llu of %llu) mapped at %p: %llu of %zu words WRONG, word 0 reads %llu (want %llu)\n", what, batch, i,
           S->addr[batch][i], S->base[batch][i], S->addr[batch][i] - S->base[batch][i], S->range[batch][i], mapped, out[0], bytes / 8, out[1], 1000ull * (batch + 1) + i);
    fflush(stdout);
  }
}

static int child(size_t bytes, int count) {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (me == 0) {
    std::vector<void*> mine;
    for (int batch = 0; batch < 2; ++batch) {
      if (batch == 1) await_peer(2);
      int shared_base = 0;
      for (int i = 0; i < count; ++i) {
        void* p = nullptr;
        CK(hipMalloc(&p, bytes));
        mine.push_back(p);
        fill_k<<<16, 256, 0, s>>>((unsigned long long*)p, bytes / 8, 1000ull * (batch + 1) + i);
        void* b = nullptr;
        size_t r = 0;
        CK(hipMemGetAddressRange(&b, &r, p));
        S->addr[batch][i] = (unsigned long long)p; S->base[batch][i] = (unsigned long long)b; S->range[batch][i] = r;
        if (b != p) ++shared_base;
      }
      CK(hipStreamSynchronize(s));
      for (int i = 0; i < count; ++i) CK(hipIpcGetMemHandle(&S->h[batch][i], mine[(size_t)batch * count + i]));
      unsigned long long lo = ~0ull, hi = 0;
      for (int i = 0; i < count; ++i) { lo = S->addr[batch][i] < lo ? S->addr[batch][i] : lo; hi = S->addr[batch][i] > hi ? S->addr[batch][i] : hi; }
      printf("[0] batch %d: %d blocks of %zu bytes exported; addresses span %#llx .. %#llx (%.1f MiB for %.1f MiB of blocks); %d of them are not the base of their allocation\n", batch, count,
             bytes, lo, hi, (hi - lo + bytes) / 1048576.0, count * (double)bytes / 1048576.0, shared_base);
      fflush(stdout);
      arrive(batch == 0 ? 1 : 3);
    }
    await_peer(4);
    for (void* p : mine) (void)hipFree(p);
    printf("[0] done\n");
    return 0;
  }
  unsigned long long* d_out = nullptr;
  CK(hipMalloc((void**)&d_out, 4096));
  std::vector<void*> map0(count, nullptr), map1(count, nullptr);
  await_peer(1);
  // two threads open the handles at once (even / odd), as two ranks of one process do
  auto open_half = [&](int parity) {
    for (int i = parity; i < count; i += 2) {
      hipError_t e = hipIpcOpenMemHandle(&map0[i], S->h[0][i], hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { printf("[1] open of batch 0 block %d -> %s\n", i, hipGetErrorString(e)); fflush(stdout); _exit(5); }
    }
  };
  std::thread t0(open_half, 0), t1(open_half, 1);
  t0.join(); t1.join();
  int same = 0;
  for (int i = 0; i < count; ++i) for (int j = 0; j < i; ++j) if (map0[i] == map0[j]) ++same;
  for (int i = 0; i < count; ++i) check("first read", 0, i, map0[i], bytes, s, d_out);
  printf("[1] batch 0 opened from two threads: %d wrong blocks so far; %d pairs of handles came back at the SAME address\n", bad_blocks, same);
  for (int i = 0; i < count; i += 2) CK(hipIpcCloseMemHandle(map0[i]));
  for (int i = 1; i < count; i += 2) check("after closing the even mappings", 0, i, map0[i], bytes, s, d_out);
  printf("[1] after closing the even mappings: %d wrong blocks so far\n", bad_blocks);
  fflush(stdout);
  arrive(2);
  await_peer(3);
  for (int i = 0; i < count; ++i) CK(hipIpcOpenMemHandle(&map1[i], S->h[1][i], hipIpcMemLazyEnablePeerAccess));
  for (int i = 0; i < count; ++i) check("second batch", 1, i, map1[i], bytes, s, d_out);
  for (int i = 1; i < count; i += 2) check("first batch again", 0, i, map0[i], bytes, s, d_out);
  printf("[1] second batch opened while the first is held: %d wrong blocks in all\n", bad_blocks);
  for (int i = 1; i < count; i += 2) (void)hipIpcCloseMemHandle(map0[i]);
  for (int i = 0; i < count; ++i) (void)hipIpcCloseMemHandle(map1[i]);
  fflush(stdout);
  arrive(4);
  return bad_blocks ? 9 : 0;
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 0) : (size_t)1 << 20;
  const int count = argc > 2 ? atoi(argv[2]) : 16;
  if (count < 2 || count > MAXB || bytes < 4096 || (bytes & 7)) { fprintf(stderr, "usage: ipc_fragment_probe BYTES COUNT(2..%d)\n", MAXB); return 2; }
  S = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (S == MAP_FAILED) return 2;
  new (S) Shared();
  pid_t pids[2];
  for (int i = 0; i < 2; ++i) {
    pids[i] = fork();
    if (pids[i] == 0) { me = i; _exit(child(bytes, count)); }
  }
  int rc = 0;
  for (int i = 0; i < 2; ++i) {
    int st = 0;
    waitpid(pids[i], &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
    if (code) { printf("child %d left with %d\n", i, code); rc = code; }
  }
  printf("ipc_fragment_probe %zu x %d: rc=%d\n", bytes, count, rc);
  return rc;
}
