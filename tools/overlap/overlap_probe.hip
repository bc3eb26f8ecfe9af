// overlap_probe.hip -- what does a communication kernel cost, and what does it get, beside the MFMA tile kernel?  
//
// One GPU, one process, the product's C-ABI: the chunked trailing update of the h = 16384 level of config 4 (per GPU: C(16384^2, upper)
// -= A^T A with K = 16384, by column chunks as summa::syrk's pipeline issues them -- capi_dgemm for the rectangle above a chunk's
// diagonal block, capi_dgemmt for the block -- launches in resident rounds, as on grids) on the compute stream, while the
// communication stream, behind each chunk's event, runs a STAND-IN for RCCL's point-to-point kernels: W persistent workgroups that
// copy the chunk's bytes (rows 0..c1 of its columns: what crosses the depth fibre) device-to-device, paced to a link rate.  RCCL's own
// kernels are not available with one GPU; what the stand-in shares with them is what matters here: they are KERNELS, they need wave
// slots on CUs that the tile kernel fills completely (2 workgroups x 256 VGPRs per SIMD lane), and they hold them for the transfer's
// whole duration.
// Measured per configuration: the update alone, the copies alone, both together; for every chunk the lag between the end of its
// compute and the start of its copy (the copy kernel stamps the wall clock itself), and the copy's duration.
//   usage: overlap_probe [h=16384] [chunks=4] [W=16] [GBps=60] [reserve=0] [reps=2]
//   environment: CAPI_COMM_PRIO_NORMAL=1 -> communication stream at default priority (round 3's setting)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "capital_hip.h"

#define CK(c) do { int rc__ = (c); if (rc__ != 0) { fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #c, rc__, capi_last_error(h)); exit(1); } } while (0)
#define HK(c) do { hipError_t e__ = (c); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #c, hipGetErrorString(e__)); exit(1); } } while (0)

__global__ void fill_kernel(double* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = (double)(x & 0xffff) * (1.0 / 65536.0) - 0.5;
  }
}
__global__ void stamp_kernel(unsigned long long* out) { *out = (unsigned long long)wall_clock64(); }

// W persistent workgroups copy `rows` x `cols` doubles (leading dimensions lds / ldd) in 64 KiB pieces; each workgroup keeps to its share of
// `ticks_per_piece` (pacing to a link rate); stamps: [0] = min start, [1] = min ~end
__global__ __launch_bounds__(256) void paced_copy_kernel(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src, int64_t lds_, int64_t rows,
                                                         int64_t cols, long long ticks_per_piece, unsigned long long* stamps) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) atomicMin(stamps, (unsigned long long)t0);
  const int64_t rp = (rows + 8191) / 8192;                    // pieces of 8192 rows (64 KiB) per column
  const int64_t pieces = rp * cols;
  long long done = 0;
  for (int64_t pc = blockIdx.x; pc < pieces; pc += gridDim.x, ++done) {
    const int64_t c = pc / rp, r0 = (pc % rp) * 8192, r1 = r0 + 8192 < rows ? r0 + 8192 : rows;
    const double2* s = (const double2*)(src + r0 + c * lds_);
    double2* d = (double2*)(dst + r0 + c * ldd);
    for (int64_t i = threadIdx.x; i < (r1 - r0) / 2; i += blockDim.x) d[i] = s[i];
    if (ticks_per_piece > 0) while (wall_clock64() - t0 < (done + 1) * ticks_per_piece) __builtin_amdgcn_s_sleep(32);
  }
  __syncthreads();
  if (threadIdx.x == 0) atomicMin(stamps + 1, ~(unsigned long long)wall_clock64());
}

int main(int argc, char** argv) {
  const int64_t H = argc > 1 ? atoll(argv[1]) : 16384;
  const int nch = argc > 2 ? atoi(argv[2]) : 4;
  const int W = argc > 3 ? atoi(argv[3]) : 16;
  const double gbps = argc > 4 ? atof(argv[4]) : 60.0;
  const int reserve = argc > 5 ? atoi(argv[5]) : 0;
  const int reps = argc > 6 ? atoi(argv[6]) : 2;
  capi_handle_t h = nullptr;
  if (capi_create(&h, 0) != 0) { fprintf(stderr, "capi_create failed\n"); return 1; }
  CK(capi_set_launch_rounds(h, 1, nullptr));
  if (reserve > 0) CK(capi_reserve_cus(h, reserve));
  double *A, *C, *Land;
  CK(capi_malloc(h, (void**)&A, sizeof(double) * H * H));
  CK(capi_malloc(h, (void**)&C, sizeof(double) * H * H));
  CK(capi_malloc(h, (void**)&Land, sizeof(double) * H * ((H + nch - 1) / nch + 2)));
  CK(capi_stream_select(h, 0));
  hipStream_t s0 = (hipStream_t)capi_get_stream(h);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, s0, A, (size_t)H * H, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, s0, C, (size_t)H * H, 2u);
  CK(capi_stream_select(h, 1));
  hipStream_t s1 = (hipStream_t)capi_get_stream(h);
  CK(capi_stream_select(h, 0));
  CK(capi_sync(h));
  int wall_khz = 100000;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  unsigned long long* st;          // per chunk: [0] compute end, [1..2] copy start / ~end
  HK(hipHostMalloc((void**)&st, sizeof(unsigned long long) * 4 * 64, hipHostMallocDefault));
  hipEvent_t e0, e1;
  HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
  auto chunk = [&](int j, int64_t& c0, int64_t& c1) { const int64_t per = ((H + nch - 1) / nch + 1) & ~(int64_t)1; c0 = std::min<int64_t>(H, per * j); c1 = std::min<int64_t>(H, c0 + per); };
  auto compute_chunk = [&](int j) {
    int64_t c0, c1; chunk(j, c0, c1);
    if (c0 > 0) CK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, c0, c1 - c0, H, -1.0, A, H, A + c0 * H, H, 1.0, C + c0 * H, H));
    CK(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, c1 - c0, H, -1.0, A + c0 * H, H, A + c0 * H, H, 1.0, C + c0 + c0 * H, H));
  };
  auto copy_chunk = [&](int j) {
    int64_t c0, c1; chunk(j, c0, c1);
    const double bytes = 8.0 * (double)c1 * (double)(c1 - c0);
    const int64_t pieces = ((c1 + 8191) / 8192) * (c1 - c0);
    // the whole chunk at `gbps`: every workgroup moves pieces / W pieces, each in (bytes / gbps) / (pieces / W) seconds
    const long long tpp = gbps > 0 ? (long long)((bytes / (gbps * 1e9)) / ((double)pieces / W) * wall_khz * 1e3) : 0;
    hipLaunchKernelGGL(paced_copy_kernel, dim3(W), dim3(256), 0, s1, Land, c1, C + c0 * H, H, c1, c1 - c0, tpp, st + 4 * j + 1);
  };
  auto ms_of = [&](unsigned long long a, unsigned long long b) { return (double)(b - a) / (double)wall_khz; };
  double best[3] = {1e30, 1e30, 1e30};
  std::vector<double> lag(nch), dur(nch), dur_alone(nch);
  for (int rep = 0; rep < reps + 1; ++rep) {
    float ms;
    // (1) the update alone
    HK(hipEventRecord(e0, s0));
    for (int j = 0; j < nch; ++j) compute_chunk(j);
    HK(hipEventRecord(e1, s0)); HK(hipEventSynchronize(e1)); HK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0) best[0] = std::min(best[0], (double)ms);
    // (2) the copies alone
    for (int j = 0; j < nch; ++j) { st[4 * j + 1] = ~0ull; st[4 * j + 2] = ~0ull; }
    HK(hipEventRecord(e0, s1));
    for (int j = 0; j < nch; ++j) copy_chunk(j);
    HK(hipEventRecord(e1, s1)); HK(hipEventSynchronize(e1)); HK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0) { best[1] = std::min(best[1], (double)ms); for (int j = 0; j < nch; ++j) dur_alone[j] = ms_of(st[4 * j + 1], ~st[4 * j + 2]); }
    // (3) the pipeline: copy of chunk j behind the event of chunk j's compute, beside the compute of chunk j + 1
    for (int j = 0; j < nch; ++j) { st[4 * j] = 0; st[4 * j + 1] = ~0ull; st[4 * j + 2] = ~0ull; }
    HK(hipEventRecord(e0, s0));
    for (int j = 0; j < nch; ++j) {
      compute_chunk(j);
      hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s0, st + 4 * j);
      CK(capi_event_record(h, 900 + j));
      CK(capi_stream_select(h, 1));
      CK(capi_event_wait(h, 900 + j));
      copy_chunk(j);
      if (j == nch - 1) CK(capi_event_record(h, 990));
      CK(capi_stream_select(h, 0));
    }
    CK(capi_event_wait(h, 990));
    HK(hipEventRecord(e1, s0)); HK(hipEventSynchronize(e1)); HK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best[2]) {
      best[2] = ms;
      for (int j = 0; j < nch; ++j) { lag[j] = ms_of(st[4 * j], st[4 * j + 1]); dur[j] = ms_of(st[4 * j + 1], ~st[4 * j + 2]); }
    }
  }
  printf("h=%lld chunks=%d W=%d link=%.0f GB/s reserve=%d CUs comm-priority=%s | update alone %.2f ms, copies alone %.2f ms, pipelined %.2f ms "
         "(sum %.2f; ideal max+last %.2f) | per chunk: lag of the copy's start behind its compute's end / copy ms (alone):",
         (long long)H, nch, W, gbps, reserve, getenv("CAPI_COMM_PRIO_NORMAL") ? "normal" : "high", best[0], best[1], best[2], best[0] + best[1],
         best[0] + dur_alone[nch - 1]);
  for (int j = 0; j < nch; ++j) printf("  [%d] %.3f / %.2f (%.2f)", j, lag[j], dur[j], dur_alone[j]);
  printf("\n");
  capi_free(h, A); capi_free(h, C); capi_free(h, Land);
  capi_destroy(h);
  return 0;
}
