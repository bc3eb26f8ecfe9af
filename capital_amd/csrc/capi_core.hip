// capi_core.hip -- handle lifecycle, device memory, workspace, timers.
// Replaces the host new[]/memcpy/memset inside matrix<> (reference src/matrix/structure.hpp:4-26).
#include <dlfcn.h>
#include <stdlib.h>
#include "capi_internal.h"

static void graphs_invalidate(capi_handle_t h);

hipError_t capi_release(capi_handle_t h, void* p) {
  static const bool defer = getenv("CAPI_DEFER_FREE") != nullptr;
  if (!p) return hipSuccess;
  if (!defer || !h) return hipFree(p);
  if (h->deferred_n == h->deferred_cap) {
    const int ncap = h->deferred_cap ? 2 * h->deferred_cap : 64;
    void** np_ = (void**)realloc(h->deferred, sizeof(void*) * ncap);
    if (!np_) return hipFree(p);
    h->deferred = np_; h->deferred_cap = ncap;
  }
  h->deferred[h->deferred_n++] = p;
  return hipSuccess;
}
extern "C" int capi_internal_copy2d(capi_handle_t h, int64_t m, int64_t n, const double* A, int64_t lda, double* B, int64_t ldb);   // movement.hip

extern "C" {

int capi_version(void) { return 100; }

int capi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int capi_create_fill(capi_handle_s* h, int device, void* stream, bool own) {
  CAPI_HIP_CHECK(h, hipSetDevice(device));
  if (own) {
    // the handle's own compute stream carries the factorisation's latency-bound chain: highest priority, so that its
    // workgroups are preferred over the bulk streams' wherever the dispatcher arbitrates
    int least = 0, greatest = 0;
    CAPI_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    CAPI_HIP_CHECK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, getenv("CAPI_MAIN_PRIO_NORMAL") ? 0 : greatest));
    h->owns_stream = true;
  } else {
    h->stream = (hipStream_t)stream;
  }
  CAPI_HIP_CHECK(h, hipMalloc((void**)&h->d_info, sizeof(int)));
  CAPI_HIP_CHECK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  CAPI_HIP_CHECK(h, hipHostMalloc((void**)&h->h_info, sizeof(int), hipHostMallocDefault));
  CAPI_HIP_CHECK(h, hipEventCreate(&h->ev0));
  CAPI_HIP_CHECK(h, hipEventCreate(&h->ev1));
  hipDeviceProp_t prop;
  CAPI_HIP_CHECK(h, hipGetDeviceProperties(&prop, device));
  h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return CAPI_OK;
}

static int capi_create_common(capi_handle_t* out, int device, void* stream, bool own) {
  if (!out) return CAPI_EINVAL;
  *out = nullptr;
  capi_handle_s* h = new capi_handle_s();
  h->device = device;
  int rc = capi_create_fill(h, device, stream, own);
  if (rc != CAPI_OK) {          // whatever was created so far goes away with the half-built handle
    fprintf(stderr, "capi_create: %s\n", h->err);
    capi_destroy(h);
    return rc;
  }
  *out = h;
  return CAPI_OK;
}

static void rounds_defaults(capi_handle_t h) {
  const char* e;
  h->rounds_env[0] = (e = getenv("CAPI_ROUNDS")) ? atoi(e) : 0;
  h->rounds_env[1] = (e = getenv("CAPI_TRMM_PAIR")) ? atoi(e) : 1;
  h->rounds_env[2] = (e = getenv("CAPI_TRMM_PAIR_ROUNDS")) ? atoi(e) : 0;
  h->rounds_env[3] = (e = getenv("CAPI_TRMM_PAIR_ROUNDS_MIN")) ? atoi(e) : 0;
  h->rounds_mode = h->rounds_env[0]; h->pair_mode = h->rounds_env[1]; h->pair_rounds = h->rounds_env[2]; h->pair_rounds_min = h->rounds_env[3];
}
int capi_create(capi_handle_t* h, int device) { int rc = capi_create_common(h, device, nullptr, true); if (rc == CAPI_OK) rounds_defaults(*h); return rc; }
int capi_create_on_stream(capi_handle_t* h, int device, void* s) { int rc = capi_create_common(h, device, s, false); if (rc == CAPI_OK) rounds_defaults(*h); return rc; }

// on = 1: every large launch of this handle goes out one resident round at a time (plain and triangular outputs, TRMMs as equal-work tile
// pairs); on = 0: back to the handle's defaults (environment).  For callers whose products all run on ONE stream (grids; the TRSM mode):
// same time, about half the L2-to-fabric traffic.  Returns the previous setting through *was (may be NULL).
int capi_set_launch_rounds(capi_handle_t h, int on, int* was) {
  CAPI_REQUIRE(h, h && (on == 0 || on == 1), "args");
  if (was) *was = (h->rounds_mode == 3 && h->pair_mode == 2 && h->pair_rounds == 1 && h->pair_rounds_min == 0) ? 1 : 0;
  if (on) { h->rounds_mode = 3; h->pair_mode = 2; h->pair_rounds = 1; h->pair_rounds_min = 0; }
  else { h->rounds_mode = h->rounds_env[0]; h->pair_mode = h->rounds_env[1]; h->pair_rounds = h->rounds_env[2]; h->pair_rounds_min = h->rounds_env[3]; }
  return CAPI_OK;
}

int capi_destroy(capi_handle_t h) {
  if (!h) return CAPI_EINVAL;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (int i = 1; i < capi_handle_s::NSTREAMS; ++i)
    if (h->streams[i]) { (void)hipStreamSynchronize(h->streams[i]); (void)hipStreamDestroy(h->streams[i]); }
  if (h->streams[0]) h->stream = h->streams[0];
  if (h->events) { for (int i = 0; i < 1024; ++i) if (h->events[i]) (void)hipEventDestroy(h->events[i]); free(h->events); }
  for (int i = 0; i < capi_handle_s::NSTREAMS; ++i) {
    if (h->ws[i]) (void)hipFree(h->ws[i]);
    if (h->ws2[i]) (void)hipFree(h->ws2[i]);
    if (h->ws3[i]) (void)hipFree(h->ws3[i]);
    if (h->ws4[i]) (void)hipFree(h->ws4[i]);
  }
  if (h->d_info) (void)hipFree(h->d_info);
  if (h->h_info) (void)hipHostFree(h->h_info);
  for (int i = 0; i < h->prof_cap; ++i) if (h->prof[i].e0) { (void)hipEventDestroy(h->prof[i].e0); (void)hipEventDestroy(h->prof[i].e1); }
  free(h->prof);
  if (h->d_stamps) (void)hipFree(h->d_stamps);
  for (int i = 0; i < h->deferred_n; ++i) (void)hipFree(h->deferred[i]);
  free(h->deferred);
  for (int i = 0; i < h->graphs_n; ++i) if (h->graphs[i].exec) (void)hipGraphExecDestroy(h->graphs[i].exec);
  free(h->graphs);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->owns_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return CAPI_OK;
}

void* capi_get_stream(capi_handle_t h) { return h ? (void*)h->stream : nullptr; }
const char* capi_last_error(capi_handle_t h) { return h ? h->err : "null handle"; }

int capi_malloc(capi_handle_t h, void** p, size_t bytes) {
  CAPI_REQUIRE(h, h && p, "null");
  hipError_t e = hipMalloc(p, bytes ? bytes : 8);
  if (e == hipErrorOutOfMemory) { snprintf(h->err, sizeof(h->err), "hipMalloc(%zu) out of memory", bytes); return CAPI_ENOMEM; }
  CAPI_HIP_CHECK(h, e);
  return CAPI_OK;
}
int capi_free(capi_handle_t h, void* p) {
  CAPI_REQUIRE(h, h, "null handle");
  if (p) { CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream)); CAPI_HIP_CHECK(h, capi_release(h, p)); }
  return CAPI_OK;
}
int capi_memset_async(capi_handle_t h, void* p, int v, size_t bytes) {
  CAPI_REQUIRE(h, h && (p || !bytes), "null");
  if (bytes) CAPI_HIP_CHECK(h, hipMemsetAsync(p, v, bytes, h->stream));
  return CAPI_OK;
}
int capi_memcpy_h2d(capi_handle_t h, void* d, const void* s, size_t bytes) {
  CAPI_REQUIRE(h, h, "null handle");
  if (!bytes) return CAPI_OK;
  CAPI_HIP_CHECK(h, hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, h->stream));
  CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return CAPI_OK;
}
int capi_memcpy_d2h(capi_handle_t h, void* d, const void* s, size_t bytes) {
  CAPI_REQUIRE(h, h, "null handle");
  if (!bytes) return CAPI_OK;
  CAPI_HIP_CHECK(h, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return CAPI_OK;
}
int capi_memcpy_d2d_async(capi_handle_t h, void* d, const void* s, size_t bytes) {
  CAPI_REQUIRE(h, h, "null handle");
  if (!bytes) return CAPI_OK;
  // whole doubles on 8-byte boundaries (every caller in the host layer): the product's own copy kernel (movement.hip), see there
  if ((bytes & 7) == 0 && ((((uintptr_t)d) | ((uintptr_t)s)) & 7) == 0) return capi_internal_copy2d(h, (int64_t)(bytes / 8), 1, (const double*)s, (int64_t)(bytes / 8), (double*)d, (int64_t)(bytes / 8));
  CAPI_HIP_CHECK(h, hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, h->stream));
  return CAPI_OK;
}
int capi_sync(capi_handle_t h) {
  CAPI_REQUIRE(h, h, "null handle");
  for (int i = capi_handle_s::NSTREAMS - 1; i >= 0; --i)
    if (h->streams[i]) CAPI_HIP_CHECK(h, hipStreamSynchronize(h->streams[i]));
  CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return CAPI_OK;
}

int capi_reserve_workspace(capi_handle_t h, size_t bytes) {
  void* p;
  return capi_ws_get(h, bytes, &p);
}

int capi_get_info(capi_handle_t h, int* info) {
  CAPI_REQUIRE(h, h && info, "null");
  CAPI_HIP_CHECK(h, hipMemcpyAsync(h->h_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  *info = *h->h_info;
  return CAPI_OK;
}
int capi_reset_info(capi_handle_t h) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_HIP_CHECK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  return CAPI_OK;
}

int capi_stream_select(capi_handle_t h, int which) {
  CAPI_REQUIRE(h, h && which >= 0 && which < capi_handle_s::NSTREAMS, "stream index");
  if (!h->streams[0]) h->streams[0] = h->stream;
  if (which >= 1 && !h->streams[which]) {
    CAPI_HIP_CHECK(h, hipSetDevice(h->device));
    if (which == 1) {
      // The communication stream carries RCCL's kernels (and packing copies): few, short-lived or few-workgroup launches that a chunk
      // pipeline is waiting for.  Highest priority, like the compute stream: whenever the dispatcher has a free slot and both queues hold
      // work, theirs goes first -- a tile launch of thousands of workgroups behind it loses nothing (CAPI_COMM_PRIO_NORMAL: default priority,
      // A/B; measured in profiles/r4_overlap_*: what decides is whether a slot is free at all, see capi_reserve_cus).
      int least = 0, greatest = 0;
      CAPI_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
      CAPI_HIP_CHECK(h, hipStreamCreateWithPriority(&h->streams[1], hipStreamNonBlocking, getenv("CAPI_COMM_PRIO_NORMAL") ? 0 : greatest));
    } else {
      // Bulk streams run long MFMA tile kernels beside the compute stream's latency-bound chain: lowest priority.
      // A tile workgroup owns a whole CU (512 threads x 256 VGPRs) and the workgroups of one launch finish in rounds, so
      // the chain's kernels mostly start when a round ends.  CAPITAL_BULK_CUS = n (< number of CUs) instead keeps the
      // bulk streams off the last CUs with a CU mask (mask bit b = CU b/8 of XCD b%8); measured at n = 32768: the chain
      // then runs freely, but the bulk launches lose their fit to 256 CUs (8 rounds become 9.1 -> 10) and the step time is
      // the same, so the default is no mask.
      const char* e = getenv("CAPITAL_BULK_CUS");
      int bulk = e ? atoi(e) : 0;
      if (bulk > 0 && bulk < h->num_cu) {
        uint32_t mask[16] = {0};
        for (int b = 0; b < bulk && b < 512; ++b) mask[b >> 5] |= 1u << (b & 31);
        CAPI_HIP_CHECK(h, hipExtStreamCreateWithCUMask(&h->streams[which], (uint32_t)((h->num_cu + 31) / 32), mask));
        h->cu_of[which] = bulk;
      } else {
        int least = 0, greatest = 0;
        CAPI_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
        CAPI_HIP_CHECK(h, hipStreamCreateWithPriority(&h->streams[which], hipStreamNonBlocking, least));
      }
    }
  }
  h->stream = h->streams[which];
  h->cur = which;
  return CAPI_OK;
}
// Keep the handle's COMPUTE stream off `reserve` CUs (0: all CUs again).  A 128-tile workgroup owns half a CU's registers for its whole
// life and a launch refills every slot the moment it frees, so kernels of another stream -- RCCL's send/recv kernels on the communication
// stream of a grid run -- find no slot before a resident round ends (stream priorities only order the dispatcher's choices, they free
// nothing).  With a CU mask on the compute stream those kernels start at once, on CUs the tile kernel never touches; the tile launches
// count their rounds on the remaining CUs (cu_of).  Mask bit b is CU b / 8 of XCD b % 8: the reserve is spread evenly over the XCDs.
// Only for handles that own their compute stream; drains the handle's streams.  NOT applied by default: measured on the chunked trailing update of
// config 4's top level, a kernel of the communication stream starts one resident round (3.7 ms at K = 16384) after its event without a reserve and
// 10 us after it with one -- but 32 reserved CUs cost the update 12.5 % (81.0 against 73.1 ms pipelined): the lag is cheaper than the cure.
int capi_reserve_cus(capi_handle_t h, int reserve) {
  // (multiples of 32 only: one CU per shader engine per XCD.  The dispatcher deals a launch's workgroups evenly over the four shader engines of
  //  an XCD; with unequal engines the short one takes a second pass while the others idle -- measured: 8 reserved CUs cost 65 % of the tile
  //  kernel's rate, 16 cost 40 %, 32 the proportional 12.5 %; profiles/r4_overlap_contention.txt)
  CAPI_REQUIRE(h, h && reserve >= 0 && reserve < h->num_cu && reserve % 32 == 0, "reserve: a multiple of 32 below the CU count");
  CAPI_REQUIRE(h, h->owns_stream, "the compute stream belongs to the caller");
  CAPI_HIP_CHECK(h, hipSetDevice(h->device));
  int rc = capi_sync(h);
  if (rc != CAPI_OK) return rc;
  if ((h->cu_of[0] ? h->num_cu - h->cu_of[0] : 0) == reserve) return CAPI_OK;
  graphs_invalidate(h);
  hipStream_t fresh = nullptr;
  int least = 0, greatest = 0;
  CAPI_HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
  if (reserve == 0) {
    CAPI_HIP_CHECK(h, hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, greatest));
  } else {
    const int keep = h->num_cu - reserve;
    uint32_t mask[16] = {0};
    for (int b = 0; b < keep && b < 512; ++b) mask[b >> 5] |= 1u << (b & 31);
    CAPI_HIP_CHECK(h, hipExtStreamCreateWithCUMask(&fresh, (uint32_t)((h->num_cu + 31) / 32), mask));
  }
  hipStream_t old = h->streams[0] ? h->streams[0] : h->stream;
  const bool selected = h->cur == 0;
  if (h->streams[0]) h->streams[0] = fresh;
  if (selected) h->stream = fresh;
  h->cu_of[0] = reserve ? h->num_cu - reserve : 0;
  CAPI_HIP_CHECK(h, hipStreamDestroy(old));
  return CAPI_OK;
}

static int event_slot(capi_handle_t h, int slot, hipEvent_t** ev) {
  CAPI_REQUIRE(h, h && slot >= 0 && slot < 1024, "event slot");
  if (!h->events) {
    h->events = (hipEvent_t*)calloc(1024, sizeof(hipEvent_t));
    if (!h->events) return CAPI_ENOMEM;
  }
  if (!h->events[slot]) CAPI_HIP_CHECK(h, hipEventCreateWithFlags(&h->events[slot], hipEventDisableTiming));
  *ev = &h->events[slot];
  return CAPI_OK;
}
int capi_event_record(capi_handle_t h, int slot) {
  hipEvent_t* ev;
  int rc = event_slot(h, slot, &ev);
  if (rc != CAPI_OK) return rc;
  CAPI_HIP_CHECK(h, hipEventRecord(*ev, h->stream));
  return CAPI_OK;
}
int capi_event_wait(capi_handle_t h, int slot) {
  hipEvent_t* ev;
  int rc = event_slot(h, slot, &ev);
  if (rc != CAPI_OK) return rc;
  CAPI_HIP_CHECK(h, hipStreamWaitEvent(h->stream, *ev, 0));
  return CAPI_OK;
}

int capi_timer_start(capi_handle_t h) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_HIP_CHECK(h, hipEventRecord(h->ev0, h->stream));
  return CAPI_OK;
}
int capi_timer_stop_ms(capi_handle_t h, float* ms) {
  CAPI_REQUIRE(h, h && ms, "null");
  CAPI_HIP_CHECK(h, hipEventRecord(h->ev1, h->stream));
  CAPI_HIP_CHECK(h, hipEventSynchronize(h->ev1));
  CAPI_HIP_CHECK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return CAPI_OK;
}


// ---- phase markers: the reference's CRITTER_START/STOP regions (src/util/shared.h:26-35) as roctx ranges, so that
// `rocprofv3 --marker-trace` attributes a step to CI::factor_diag / CI::trsm / CI::tmu / CQR::gram / CQR::formR.  The roctx
// library is bound on first use (rocprofiler-sdk's, which rocprofv3 intercepts; roctracer's as a fallback); without
// either, or with CAPI_NO_MARKERS set, the calls cost one predictable branch.
namespace {
struct RoctxApi {
  bool tried = false;
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
} g_roctx;
void roctx_bind() {
  g_roctx.tried = true;
  if (getenv("CAPI_NO_MARKERS")) return;
  const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                         "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so.4"};
  for (const char* nm : names) {
    void* lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) continue;
    *(void**)(&g_roctx.push) = dlsym(lib, "roctxRangePushA");
    *(void**)(&g_roctx.pop) = dlsym(lib, "roctxRangePop");
    if (g_roctx.push && g_roctx.pop) return;
    g_roctx.push = nullptr; g_roctx.pop = nullptr;
  }
}
}  // namespace

int capi_range_push(const char* name) {
  if (!g_roctx.tried) roctx_bind();
  if (g_roctx.push && name) g_roctx.push(name);
  return CAPI_OK;
}
int capi_range_pop(void) {
  if (!g_roctx.tried) roctx_bind();
  if (g_roctx.pop) g_roctx.pop();
  return CAPI_OK;
}

}  // extern "C"

// A captured launch chain (factor_f64.hip: CAPI_GRAPH) has the workspace pointers of its launches baked in -- the primary block's
// split-K slabs, ws2 / ws3 scratch -- and the profile / kernel-choice state of the day it was captured: when any workspace block
// moves or is released, every captured chain of the handle is dropped and re-captured on its next use.
static void graphs_invalidate(capi_handle_t h) {
  for (int i = 0; i < h->graphs_n; ++i) if (h->graphs[i].exec) (void)hipGraphExecDestroy(h->graphs[i].exec);
  h->graphs_n = 0;
}

static int ws_grow(capi_handle_t h, void** slot, size_t* cap, size_t bytes, void** p) {
  if (!h || !p) return CAPI_EINVAL;
  if (bytes > *cap) {
    // stream-ordered users of the old block must finish before it is released -- and a captured chain that was replayed just before may still
    // be in flight on any of the handle's streams: everything drains first, only then are the captured chains dropped
    for (int i = capi_handle_s::NSTREAMS - 1; i >= 0; --i)
      if (h->streams[i] && h->streams[i] != h->stream) CAPI_HIP_CHECK(h, hipStreamSynchronize(h->streams[i]));
    CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    graphs_invalidate(h);
    if (*slot) CAPI_HIP_CHECK(h, capi_release(h, *slot));
    *slot = nullptr;
    *cap = 0;
    size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    // CAPI_WS_CAP_MB (diagnostics / tests): requests above it fail as an out-of-memory hipMalloc would -- the fallback paths behind
    // CAPI_ENOMEM can then be exercised without exhausting 288 GB
    const char* cap_mb = getenv("CAPI_WS_CAP_MB");
    hipError_t e = (cap_mb && want > ((size_t)atoll(cap_mb) << 20)) ? hipErrorOutOfMemory : hipMalloc(slot, want);
    if (e == hipErrorOutOfMemory) {
      // callers with a slower path that needs no workspace (qr_f64.hip: the Householder panels behind the tall-panel routines) go on after
      // this status: HIP's sticky last-error must not be what their next hipGetLastError() check sees
      (void)hipGetLastError();
      snprintf(h->err, sizeof(h->err), "workspace hipMalloc(%zu) out of memory", want);
      return CAPI_ENOMEM;
    }
    CAPI_HIP_CHECK(h, e);
    *cap = want;
  }
  *p = *slot;
  return CAPI_OK;
}
int capi_ws_get(capi_handle_t h, size_t bytes, void** p) { return h ? ws_grow(h, &h->ws[h->cur], &h->ws_bytes[h->cur], bytes, p) : CAPI_EINVAL; }
int capi_ws2_get(capi_handle_t h, size_t bytes, void** p) { return h ? ws_grow(h, &h->ws2[h->cur], &h->ws2_bytes[h->cur], bytes, p) : CAPI_EINVAL; }
int capi_ws3_get(capi_handle_t h, size_t bytes, void** p) { return h ? ws_grow(h, &h->ws3[h->cur], &h->ws3_bytes[h->cur], bytes, p) : CAPI_EINVAL; }
int capi_ws4_get(capi_handle_t h, size_t bytes, void** p) { return h ? ws_grow(h, &h->ws4[h->cur], &h->ws4_bytes[h->cur], bytes, p) : CAPI_EINVAL; }

// release every private workspace block of the handle (they grow on demand and are otherwise kept until capi_destroy)
extern "C" int capi_trim_workspaces(capi_handle_t h) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_HIP_CHECK(h, hipSetDevice(h->device));
  int rc = capi_sync(h);
  if (rc != CAPI_OK) return rc;
  graphs_invalidate(h);
  for (int i = 0; i < capi_handle_s::NSTREAMS; ++i) {
    void** blocks[4] = {&h->ws[i], &h->ws2[i], &h->ws3[i], &h->ws4[i]};
    size_t* sizes[4] = {&h->ws_bytes[i], &h->ws2_bytes[i], &h->ws3_bytes[i], &h->ws4_bytes[i]};
    for (int b = 0; b < 4; ++b)
      if (*blocks[b]) { CAPI_HIP_CHECK(h, capi_release(h, *blocks[b])); *blocks[b] = nullptr; *sizes[b] = 0; }
  }
  return CAPI_OK;
}
