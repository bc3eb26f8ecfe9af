// capi_internal.h -- private state of libcapital_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "capital_hip.h"

struct capi_handle_s {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // streams[0] = compute (== the handle's primary stream), streams[1] = communication / packing, streams[2..] = bulk
  // trailing updates that run beside the factorisation's latency-bound chain (lowest priority); created on first select
  static constexpr int NSTREAMS = 4;
  hipStream_t streams[NSTREAMS] = {nullptr, nullptr, nullptr, nullptr};
  int cur = 0;                    // index of the selected stream
  int cu_of[NSTREAMS] = {0, 0, 0, 0};   // CUs a stream may use when it carries a CU mask (0 = all): launch heuristics count rounds with it
  hipEvent_t* events = nullptr;   // 1024 lazily created slots
  // private workspace (in-place trmm staging, split-K slabs, potrf panels); grows on demand.  One per stream index:
  // reuse is stream-ordered, so work on different streams must not share a block
  void* ws[NSTREAMS] = {nullptr, nullptr, nullptr, nullptr};
  size_t ws_bytes[NSTREAMS] = {0, 0, 0, 0};
  // second, independent scratch block (diagonal-block inverses, recursion temporaries)
  void* ws2[NSTREAMS] = {nullptr, nullptr, nullptr, nullptr};
  size_t ws2_bytes[NSTREAMS] = {0, 0, 0, 0};
  // third block: clean copies of small triangular operands (tall right-TRMM); never aliases what callers keep in ws / ws2
  void* ws3[NSTREAMS] = {nullptr, nullptr, nullptr, nullptr};
  size_t ws3_bytes[NSTREAMS] = {0, 0, 0, 0};
  // fourth block: the tall QR paths' panels (CholeskyQR2 + Householder reconstruction keeps two m x n images beside the BLAS /
  // LAPACK calls it makes, which use ws / ws2 / ws3 themselves).  hipMalloc of tens of GiB takes hundreds of ms: kept between calls
  void* ws4[NSTREAMS] = {nullptr, nullptr, nullptr, nullptr};
  size_t ws4_bytes[NSTREAMS] = {0, 0, 0, 0};
  // device-side LAPACK info word (0 ok, >0 first bad pivot, 1-based) + pinned host mirror
  int* d_info = nullptr;
  int* h_info = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cu = 256;
  // per-launch HIP-event bracketing of the MFMA tile kernel (capi_prof_*): events live in a pool, results are summed on collect
  bool prof_on = false;
  struct prof_rec { hipEvent_t e0, e1; double flops; int variant; int m, n, k, kind; };   // kind: 0 plain, 1 triangular output, 2 TRMM
  prof_rec* prof = nullptr;
  int prof_n = 0, prof_cap = 0;
  // per-launch execution intervals written by the tile kernels themselves (gemm_f64.hip: stamp_begin / stamp_end), record i at [2 i, 2 i + 1]
  unsigned long long* d_stamps = nullptr;
  int stamps_cap = 0;
  int wall_khz = 100000;          // rate of wall_clock64() (hipDeviceAttributeWallClockRate)
  // CAPI_DEFER_FREE: blocks whose release is put off until capi_destroy.  hipFree drains the WHOLE device; with several handles of one process on
  // one GPU whose streams wait for each other's messages on the device (ranks as threads over the asynchronous loopback transport) that wait can
  // never end.  Off by default: one rank per process frees at once.
  void** deferred = nullptr;
  int deferred_n = 0, deferred_cap = 0;
  // kernels whose dynamic-LDS limit has been raised on THIS handle's device (hipFuncSetAttribute is per device; a
  // process-wide flag would leave a second device's copy of the kernel at the default limit)
  uint32_t lds_attr_done = 0;
  // captured launch sequences of the blocked diagonal-block routine, one per (block, order): factor_f64.hip
  struct graph_ent { int64_t n, lda, ldx; double *A, *X; void* w; hipStream_t s; hipGraphExec_t exec; int seen; };
  graph_ent* graphs = nullptr;
  int graphs_n = 0, graphs_cap = 0;
  // large launches one resident round (2 x CUs workgroups) at a time -- capi_set_launch_rounds; the environment gives the defaults
  // (CAPI_ROUNDS: bit 0 plain products, bit 1 triangular outputs; CAPI_TRMM_PAIR: 0 never / 1 up to four whole rounds / 2 every whole-round
  // launch in tile pairs; CAPI_TRMM_PAIR_ROUNDS: pairs one round per launch, from K = CAPI_TRMM_PAIR_ROUNDS_MIN up)
  int rounds_mode = 0, pair_mode = 1, pair_rounds = 0, pair_rounds_min = 0;
  int rounds_env[4] = {0, 1, 0, 0};
  char err[512] = {0};
};

enum { CAPI_ATTR_LEAF = 0, CAPI_ATTR_TRMM_TS32 = 1, CAPI_ATTR_TRMM_TS16 = 2, CAPI_ATTR_GRAM_TS = 3, CAPI_ATTR_SMALL0 = 4 /* ..7 */,
       CAPI_ATTR_GRAM_WIDE = 8, CAPI_ATTR_TRMM_WIDE = 9, CAPI_ATTR_PAIR0 = 10 /* ..13 */ };
#define CAPI_RAISE_LDS_LIMIT(h, bit, fn, bytes)                                                                    \
  do {                                                                                                             \
    if (!((h)->lds_attr_done & (1u << (bit)))) {                                                                   \
      CAPI_HIP_CHECK(h, hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      (h)->lds_attr_done |= 1u << (bit);                                                                           \
    }                                                                                                              \
  } while (0)

#define CAPI_HIP_CHECK(h, call)                                                              \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      if (h) snprintf((h)->err, sizeof((h)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return CAPI_EHIP;                                                                      \
    }                                                                                        \
  } while (0)

#define CAPI_REQUIRE(h, cond, msg)                                                           \
  do {                                                                                       \
    if (!(cond)) {                                                                           \
      if (h) snprintf((h)->err, sizeof((h)->err), "%s:%d invalid argument: %s", __FILE__, __LINE__, msg); \
      return CAPI_EINVAL;                                                                    \
    }                                                                                        \
  } while (0)

// returns a workspace pointer of at least `bytes` (stream-ordered reuse: callers run on h->stream)
int capi_ws_get(capi_handle_t h, size_t bytes, void** p);
int capi_ws2_get(capi_handle_t h, size_t bytes, void** p);
int capi_ws3_get(capi_handle_t h, size_t bytes, void** p);
int capi_ws4_get(capi_handle_t h, size_t bytes, void** p);

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFree, or (CAPI_DEFER_FREE) remember the block until capi_destroy
hipError_t capi_release(capi_handle_t h, void* p);
