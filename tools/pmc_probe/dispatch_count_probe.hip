// dispatch_count_probe.hip -- diagnostics for the host SIGSEGV of `rocprofv3 --pmc` in long bench processes (profiles/README.md).
// NOT the product: a program of its own that does nothing but dispatch trivial kernels, N of them, from two kernel symbols and two
// streams, printing its progress.  If THIS crashes under `rocprofv3 --pmc FETCH_SIZE` after a comparable number of dispatches, the
// fault needs nothing of the product (its kernels, its argument structs, its workspaces) -- only the profiler and a dispatch count.
//   usage: dispatch_count_probe N [dynamic_lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Args { double* p; long n; double a[24]; };          // a by-value struct of the tile kernels' size (232 bytes)
__global__ void k_one(Args a) { extern __shared__ double l[]; if (threadIdx.x == 0 && a.n < 0) a.p[0] = l[0]; }
template <bool X> __global__ void k_two(Args a) { extern __shared__ double l[]; if (threadIdx.x == 0 && a.n < 0) a.p[1] = X ? l[1] : a.a[3]; }

#define CK(c) do { hipError_t e = (c); if (e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #c, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 100000;
  const int lds = argc > 2 ? atoi(argv[2]) : 69632;
  double* d;
  CK(hipMalloc((void**)&d, 4096));
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)k_one, hipFuncAttributeMaxDynamicSharedMemorySize, 139264));
  CK(hipFuncSetAttribute((const void*)k_two<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 139264));
  CK(hipFuncSetAttribute((const void*)k_two<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 139264));
  Args a{d, 1, {0}};
  for (long i = 0; i < N; ++i) {
    switch (i % 3) {
      case 0: hipLaunchKernelGGL(k_one, dim3(4), dim3(256), lds, s[i & 1], a); break;
      case 1: hipLaunchKernelGGL(k_two<true>, dim3(4, 2), dim3(256), lds, s[i & 1], a); break;
      default: hipLaunchKernelGGL(k_two<false>, dim3(1), dim3(64), 0, s[i & 1], a); break;
    }
    if ((i + 1) % 5000 == 0) {
      CK(hipGetLastError());
      CK(hipStreamSynchronize(s[0]));
      CK(hipStreamSynchronize(s[1]));
      printf("dispatched %ld\n", i + 1);
      fflush(stdout);
    }
  }
  CK(hipDeviceSynchronize());
  printf("done: %ld dispatches, no fault\n", N);
  return 0;
}
