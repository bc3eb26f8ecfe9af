// shared.h -- process-wide context of the host-side C++ layer (one process per GPU).
//
// Takes the place of the reference's src/util/shared.h (which pulls in <mpi.h> and "mkl.h"): here the two
// external dependencies are the C-ABI of libcapital_hip.so (device BLAS/LAPACK/data movement) and its RCCL-backed
// communicators.  MPI_COMM_WORLD becomes capital::world(); MPI_Init becomes capital::init().
#ifndef CAPITAL_SHARED_H_
#define CAPITAL_SHARED_H_

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "capital_hip.h"

// Region markers of the reference's external profiler (shared.h:26-35: CRITTER_START(sym) / CRITTER_STOP(sym) around the
// algorithmic phases CI::factor_diag, CI::trsm, CI::tmu, CQR::gram, CQR::formR) become roctx ranges with the same names:
// `rocprofv3 --marker-trace` then attributes a step to its phases.  -DCAPITAL_NO_MARKERS compiles them out.
#ifndef CRITTER_START
#ifdef CAPITAL_NO_MARKERS
#define CRITTER_START(ARG)
#define CRITTER_STOP(ARG)
#else
#define CRITTER_START(ARG) capi_range_push(#ARG)
#define CRITTER_STOP(ARG) capi_range_pop()
#endif
#endif

// One rank per PROCESS is the product's model (one process per GPU).  -DCAPITAL_THREAD_RANKS makes the layer's three pieces of per-rank state
// thread-local instead (this context, summa's scratch arena and event-slot counter), so that a rehearsal can run several ranks as threads of
// one process -- tests/thread_ranks: the 2 x 2 x 2 grid's eight ranks on ONE GPU inside the pool's limit of six GPU processes.
#ifdef CAPITAL_THREAD_RANKS
#define CAPITAL_RANK_LOCAL thread_local
#else
#define CAPITAL_RANK_LOCAL
#endif

namespace capital {

struct context {
  capi_handle_t handle = nullptr;
  capi_comm_t world = nullptr;
  int rank = 0, size = 1, device = 0;
  bool owns_handle = false;
};

inline context& ctx() {
  static CAPITAL_RANK_LOCAL context c;
  return c;
}

inline void check(int rc, const char* what) {
  if (rc != CAPI_OK) {
    const char* msg = ctx().handle ? capi_last_error(ctx().handle) : "";
    char buf[768];
    snprintf(buf, sizeof(buf), "capital: %s failed (status %d) %s", what, rc, msg);
    throw std::runtime_error(buf);
  }
}
#define CAPITAL_CHECK(call) ::capital::check((call), #call)

// Bind the layer to an existing handle (e.g. one living on torch's stream) and, for size > 1, to a world
// communicator built from a 128-byte RCCL unique id that the launcher distributed.
inline void init_with_handle(capi_handle_t h, int rank, int size, const void* unique_id) {
  context& c = ctx();
  c.handle = h;
  c.rank = rank;
  c.size = size;
  if (c.world) { capi_comm_destroy(c.world); c.world = nullptr; }
  CAPITAL_CHECK(capi_comm_init_rank(&c.world, h, size, unique_id, rank));
}
inline void init(int device, int rank, int size, const void* unique_id) {
  capi_handle_t h;
  int rc = capi_create(&h, device);
  if (rc != CAPI_OK) throw std::runtime_error("capital: capi_create failed (is a GPU visible and libcapital_hip.so built?)");
  ctx().owns_handle = true;
  ctx().device = device;
  init_with_handle(h, rank, size, unique_id);
}
inline void finalize() {
  context& c = ctx();
  if (c.world) { capi_comm_destroy(c.world); c.world = nullptr; }
  if (c.handle && c.owns_handle) capi_destroy(c.handle);
  c.handle = nullptr;
  c.owns_handle = false;
}
inline capi_handle_t handle() {
  if (!ctx().handle) throw std::runtime_error("capital: call capital::init() first");
  return ctx().handle;
}
inline capi_comm_t world() { return ctx().world; }
inline void sync() { CAPITAL_CHECK(capi_sync(handle())); }

// Device buffer with value semantics of a raw pointer + ownership flag (the reference's new[]/delete[]).
inline double* dev_alloc(int64_t count) {
  void* p = nullptr;
  CAPITAL_CHECK(capi_malloc(handle(), &p, sizeof(double) * (size_t)std::max<int64_t>(count, 1)));
  return (double*)p;
}
inline void dev_free(double* p) {
  if (p && ctx().handle) capi_free(ctx().handle, p);
}
inline void dev_zero(double* p, int64_t count) { CAPITAL_CHECK(capi_memset_async(handle(), p, 0, sizeof(double) * (size_t)count)); }
inline void dev_copy(double* dst, const double* src, int64_t count) {
  CAPITAL_CHECK(capi_memcpy_d2d_async(handle(), dst, src, sizeof(double) * (size_t)count));
}

// Scope in which every large launch of the handle goes out one resident round at a time (capi_set_launch_rounds): for schedules whose
// products all run on ONE stream -- grids (the lookahead is a single-GPU device), the TRSM mode.  CAPITAL_NO_LAUNCH_ROUNDS keeps the defaults.
struct launch_rounds_scope {
  int was = 0;
  bool active = false;
  explicit launch_rounds_scope(bool on) {
    if (on && !getenv("CAPITAL_NO_LAUNCH_ROUNDS")) { check(capi_set_launch_rounds(handle(), 1, &was), "capi_set_launch_rounds"); active = true; }
  }
  launch_rounds_scope(const launch_rounds_scope&) = delete;
  launch_rounds_scope& operator=(const launch_rounds_scope&) = delete;
  ~launch_rounds_scope() { if (active && !was && ctx().handle) (void)capi_set_launch_rounds(ctx().handle, 0, nullptr); }
};

// How many ranks of `comm` hold a non-zero status word (collective over comm; one 8-byte all-reduce and one 8-byte read).
// An error that only some ranks can see -- the base-case policies that factor on layer 0 or on the slice root alone
// (policy.h:226-514) -- must unwind every rank or none: a rank that throws while its peers go on leaves them hanging in
// their next collective.
inline int ranks_with_nonzero(capi_comm_t comm, int word) {
  int size = 1;
  if (!comm || capi_comm_size(comm, &size) != CAPI_OK || size <= 1) return word != 0 ? 1 : 0;
  double flag = word != 0 ? 1.0 : 0.0;
  double* d = dev_alloc(1);
  int rc = capi_memcpy_h2d(handle(), d, &flag, sizeof(double));
  if (rc == CAPI_OK) rc = capi_allreduce_sum(comm, d, 1);
  if (rc == CAPI_OK) rc = capi_memcpy_d2h(handle(), &flag, d, sizeof(double));
  dev_free(d);
  check(rc, "ranks_with_nonzero");
  return (int)(flag + 0.5);
}

}  // namespace capital

#endif  // CAPITAL_SHARED_H_
