// which XCD does block b land on, and when?  (diagnostic for the tile -> XCD mapping of gemm_f64.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(int* xcc, unsigned long long* t0, int spin) {
  extern __shared__ double lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  unsigned long long t = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { xcc[blockIdx.x] = (int)(id & 0xf); t0[blockIdx.x] = t; }
  // burn time so that later blocks queue behind earlier ones
  double a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;
  lds[threadIdx.x] = a;
  __syncthreads();
  if (a == 12345.678) xcc[0] = -1;
}
int main() {
  const int nblk = 4096;
  int* dx; unsigned long long* dt;
  hipMalloc(&dx, nblk * 4); hipMalloc(&dt, nblk * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 73728, 0, dx, dt, 20000);
  hipDeviceSynchronize();
  std::vector<int> x(nblk); std::vector<unsigned long long> t(nblk);
  hipMemcpy(x.data(), dx, nblk * 4, hipMemcpyDeviceToHost); hipMemcpy(t.data(), dt, nblk * 8, hipMemcpyDeviceToHost);
  printf("first 32 blocks xcc:"); for (int i = 0; i < 32; ++i) printf(" %d", x[i]); printf("\n");
  int rr = 0; for (int i = 0; i < nblk; ++i) if (x[i] == x[i % 8]) ++rr;
  printf("blocks with xcc[b]==xcc[b%%8]: %d of %d\n", rr, nblk);
  // how many distinct start-time "rounds": count blocks started within 20us of block 0
  unsigned long long tmin = t[0]; for (auto v : t) if (v < tmin) tmin = v;
  int first = 0; for (auto v : t) if (v - tmin < 2000) ++first;   // 100 MHz ticks: 2000 = 20 us
  printf("blocks started in the first 20 us: %d\n", first);
  // per xcc count in first wave
  int cnt[16] = {0}; for (int i = 0; i < nblk; ++i) if (t[i] - tmin < 2000) cnt[x[i]]++;
  printf("first-wave per xcc:"); for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]); printf("\n");
  // order of start among blocks of xcc 0: list first 20 block ids by start time
  std::vector<std::pair<unsigned long long,int>> v; for (int i = 0; i < nblk; ++i) if (x[i] == x[0]) v.push_back({t[i], i});
  std::sort(v.begin(), v.end());
  printf("xcc(x[0]) start order (block ids):"); for (int i = 0; i < 80 && i < (int)v.size(); ++i) printf(" %d", v[i].second); printf("\n");
  return 0;
}
