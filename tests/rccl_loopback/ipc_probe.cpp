// ipc_probe.cpp -- TEST INFRASTRUCTURE: which cross-process, device-side mechanisms does this pool's driver offer two processes
// that share ONE GPU?  Decides what the asynchronous mode of rccl_loopback is built on.  The parent forks two children before any
// HIP call and never touches the GPU itself; every device-side wait is polled on the host with a limit, and a child that runs
// into its limit says so and leaves (its queues die with it).
//   T1  hipIpcGetMemHandle / hipIpcOpenMemHandle of a hipMalloc block (base pointer, and a pointer inside the block)
//   T2  interprocess events: record behind a slow kernel in A, hipStreamWaitEvent + copy in B
//   T3  hipStreamWriteValue32 / hipStreamWaitValue32 on a host page both processes registered
//   T4  the same with the flag word in IPC-mapped device memory
//   T5  a copy kernel of B reading A's block through the IPC mapping
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>

struct Shared {
  std::atomic<int> stage[2];
  hipIpcMemHandle_t mem, mem_off, flagmem;
  hipIpcEventHandle_t ev;
  int mem_off_rc, ev_rc;
  alignas(4096) uint32_t flags[1024];      // the page both children register (T3)
};

static Shared* S;
static int me;

#define CK(call)                                                                                     \
  do {                                                                                               \
    hipError_t e__ = (call);                                                                         \
    if (e__ != hipSuccess) { printf("[%d] %s:%d %s -> %s\n", me, __FILE__, __LINE__, #call, hipGetErrorString(e__)); fflush(stdout); _exit(3); } \
  } while (0)

static void arrive(int v) { S->stage[me].store(v); }
static bool await_peer(int v, double limit_s = 60) {
  auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(limit_s);
  while (S->stage[1 - me].load() < v) {
    if (std::chrono::steady_clock::now() > t_end) { printf("[%d] peer never reached stage %d\n", me, v); fflush(stdout); _exit(4); }
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  return true;
}
static bool drain(hipStream_t s, const char* what, double limit_s = 20) {
  auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(limit_s);
  for (;;) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) { printf("[%d] %s: stream query -> %s\n", me, what, hipGetErrorString(e)); fflush(stdout); _exit(5); }
    if (std::chrono::steady_clock::now() > t_end) { printf("[%d] %s: DEVICE WAIT NEVER RELEASED (limit %.0f s)\n", me, what, limit_s); fflush(stdout); _exit(6); }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

__global__ void slow_fill(double* p, size_t n, double v, long long ticks) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  }
  __syncthreads();
  if (blockIdx.x == 0) for (size_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = v;
}
__global__ void copy_k(double* __restrict__ d, const double* __restrict__ s, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

static double first_of(const double* dev, hipStream_t s) {
  double h = -1;
  CK(hipMemcpyAsync(&h, dev, 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  return h;
}

static int child() {
  const size_t N = 1 << 20;                 // doubles
  const long long TICKS = 20000000;         // 0.2 s at the 100 MHz wall clock
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int can_wait = -1;
  (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
  double *mine = nullptr, *peer = nullptr, *peer_off = nullptr, *local = nullptr;
  uint32_t* flagdev = nullptr;
  hipEvent_t ev = nullptr;
  CK(hipMalloc((void**)&local, N * 8));
  CK(hipMemsetAsync(local, 0, N * 8, s));

  // ---- T1
  if (me == 0) {
    CK(hipMalloc((void**)&mine, N * 8));
    CK(hipMalloc((void**)&flagdev, 4096));
    CK(hipMemset(flagdev, 0, 4096));
    slow_fill<<<1, 256, 0, s>>>(mine, N, 1.0, 0);
    CK(hipStreamSynchronize(s));
    CK(hipIpcGetMemHandle(&S->mem, mine));
    S->mem_off_rc = (int)hipIpcGetMemHandle(&S->mem_off, mine + 4096);
    CK(hipIpcGetMemHandle(&S->flagmem, flagdev));
    CK(hipEventCreateWithFlags(&ev, hipEventInterprocess | hipEventDisableTiming));
    S->ev_rc = (int)hipIpcGetEventHandle(&S->ev, ev);
    printf("[0] T1 export ok; handle of an inner pointer rc=%d; T2 event export rc=%d (%s); CanUseStreamWaitValue=%d\n", S->mem_off_rc, S->ev_rc,
           hipGetErrorString((hipError_t)S->ev_rc), can_wait);
    fflush(stdout);
    arrive(1);
    await_peer(1);
  } else {
    await_peer(1);
    CK(hipIpcOpenMemHandle((void**)&peer, S->mem, hipIpcMemLazyEnablePeerAccess));
    CK(hipMemcpyAsync(local, peer, N * 8, hipMemcpyDeviceToDevice, s));
    printf("[1] T1 open + D2D copy: first=%g (want 1)\n", first_of(local, s));
    if (S->mem_off_rc == 0) {
      hipError_t e = hipIpcOpenMemHandle((void**)&peer_off, S->mem_off, hipIpcMemLazyEnablePeerAccess);
      printf("[1] T1 inner-pointer handle opens: %s; maps to base+%lld bytes\n", hipGetErrorString(e), e == hipSuccess ? (long long)((char*)peer_off - (char*)peer) : -1LL);
    }
    CK(hipIpcOpenMemHandle((void**)&flagdev, S->flagmem, hipIpcMemLazyEnablePeerAccess));
    if (S->ev_rc == 0) {
      hipError_t e = hipIpcOpenEventHandle(&ev, S->ev);
      printf("[1] T2 event handle opens: %s\n", hipGetErrorString(e));
      if (e != hipSuccess) ev = nullptr;
    }
    fflush(stdout);
    arrive(1);
  }

  // ---- T2: interprocess event
  if (me == 0) {
    if (S->ev_rc == 0) {
      slow_fill<<<1, 256, 0, s>>>(mine, N, 2.0, TICKS);
      CK(hipEventRecord(ev, s));
    }
    arrive(2);
    await_peer(2);
    CK(hipStreamSynchronize(s));
  } else {
    await_peer(2);                                  // the record has been ISSUED (not completed)
    if (ev) {
      CK(hipStreamWaitEvent(s, ev, 0));
      CK(hipMemcpyAsync(local, peer, N * 8, hipMemcpyDeviceToDevice, s));
      drain(s, "T2");
      printf("[1] T2 interprocess event: copy behind hipStreamWaitEvent saw %g (2 = waited, 1 = did not wait)\n", first_of(local, s));
    } else printf("[1] T2 skipped\n");
    fflush(stdout);
    arrive(2);
  }

  // ---- T3: write/wait value on a registered host page
  uint32_t* flaghost = nullptr;
  hipError_t reg = hipHostRegister(S->flags, sizeof(S->flags), hipHostRegisterMapped);
  if (reg == hipSuccess) reg = hipHostGetDevicePointer((void**)&flaghost, S->flags, 0);
  printf("[%d] T3 hipHostRegister of the shared page: %s\n", me, hipGetErrorString(reg));
  fflush(stdout);
  if (me == 0) {
    await_peer(3);                                  // B has enqueued its wait first: the wait really waits
    if (reg == hipSuccess) {
      slow_fill<<<1, 256, 0, s>>>(mine, N, 3.0, TICKS);
      hipError_t e = hipStreamWriteValue32(s, flaghost, 7, 0);
      printf("[0] T3 hipStreamWriteValue32: %s\n", hipGetErrorString(e));
      CK(hipStreamSynchronize(s));
      printf("[0] T3 host sees flag %u\n", S->flags[0]);
      if (e != hipSuccess) S->flags[0] = 7;         // release the peer anyway
    } else S->flags[0] = 7;
    fflush(stdout);
    arrive(3);
  } else {
    if (reg == hipSuccess) {
      hipError_t e = hipStreamWaitValue32(s, flaghost, 7, hipStreamWaitValueGte, 0xffffffffu);
      printf("[1] T3 hipStreamWaitValue32: %s\n", hipGetErrorString(e));
      CK(hipMemcpyAsync(local, peer, N * 8, hipMemcpyDeviceToDevice, s));
      arrive(3);
      drain(s, "T3");
      printf("[1] T3 host-page flag: copy behind the wait saw %g (3 = waited)\n", first_of(local, s));
    } else arrive(3);
    fflush(stdout);
    await_peer(3);
  }

  // ---- T4: flag word in IPC-mapped device memory
  if (me == 0) {
    await_peer(4);
    slow_fill<<<1, 256, 0, s>>>(mine, N, 4.0, TICKS);
    hipError_t e = hipStreamWriteValue32(s, flagdev, 9, 0);
    printf("[0] T4 hipStreamWriteValue32 (device word): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) CK(hipMemsetD32Async((hipDeviceptr_t)flagdev, 9, 1, s));
    CK(hipStreamSynchronize(s));
    fflush(stdout);
    arrive(4);
  } else {
    hipError_t e = hipStreamWaitValue32(s, flagdev, 9, hipStreamWaitValueGte, 0xffffffffu);
    printf("[1] T4 hipStreamWaitValue32 (device word): %s\n", hipGetErrorString(e));
    CK(hipMemcpyAsync(local, peer, N * 8, hipMemcpyDeviceToDevice, s));
    arrive(4);
    if (e == hipSuccess) {
      drain(s, "T4");
      printf("[1] T4 device-word flag: copy behind the wait saw %g (4 = waited)\n", first_of(local, s));
    }
    fflush(stdout);
    await_peer(4);
    CK(hipStreamSynchronize(s));
  }

  // ---- T5: kernel reads through the mapping
  if (me == 1) {
    copy_k<<<16, 256, 0, s>>>(local, peer, N);
    drain(s, "T5");
    printf("[1] T5 copy kernel through the IPC mapping saw %g (want 4)\n", first_of(local, s));
    fflush(stdout);
    arrive(5);
    if (peer_off) (void)hipIpcCloseMemHandle(peer_off);
    CK(hipIpcCloseMemHandle(peer));
    CK(hipIpcCloseMemHandle(flagdev));
    arrive(6);
  } else {
    await_peer(6);                                  // the exporter frees after the importer closed
    CK(hipFree(mine));
    CK(hipFree(flagdev));
  }
  if (reg == hipSuccess) (void)hipHostUnregister(S->flags);
  printf("[%d] done\n", me);
  fflush(stdout);
  return 0;
}

int main() {
  S = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (S == MAP_FAILED) return 1;
  memset((void*)S, 0, sizeof(Shared));
  pid_t pid[2];
  for (int r = 0; r < 2; ++r) {
    pid[r] = fork();
    if (pid[r] == 0) { me = r; _exit(child()); }
  }
  int bad = 0;
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(150);
  for (int done = 0; done < 2;) {
    int st = 0;
    pid_t p = waitpid(-1, &st, WNOHANG);
    if (p > 0) { ++done; if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { bad = 1; printf("child %d ended with status 0x%x\n", (int)p, st); } continue; }
    if (std::chrono::steady_clock::now() > t_end) { for (int r = 0; r < 2; ++r) kill(pid[r], SIGKILL); printf("probe limit reached\n"); return 2; }
    usleep(20000);
  }
  return bad;
}
