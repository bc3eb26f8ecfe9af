// bench/launch.h -- process bootstrap shared by the two bench mains: what MPI_Init_thread + MPI_COMM_WORLD did in the
// reference benches.  One process per GPU.  Single process: nothing to configure.  Several processes on one node:
// the launcher sets RANK, WORLD_SIZE and LOCAL_RANK (torchrun / mpiexec style) and CAPITAL_UID_FILE to a path on a
// shared filesystem; rank 0 writes the 128-byte RCCL unique id there and the others pick it up.
#ifndef CAPITAL_BENCH_LAUNCH_H_
#define CAPITAL_BENCH_LAUNCH_H_

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../src/util/shared.h"

namespace capital_bench {

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

inline void init(int& rank, int& size) {
  rank = env_int("RANK", env_int("PMI_RANK", 0));
  size = env_int("WORLD_SIZE", env_int("PMI_SIZE", 1));
  const int device = env_int("LOCAL_RANK", rank);
  unsigned char uid[128] = {0};
  if (size > 1) {
    const char* path = getenv("CAPITAL_UID_FILE");
    if (!path) throw std::runtime_error("WORLD_SIZE > 1 needs CAPITAL_UID_FILE (shared path for the RCCL unique id)");
    std::string done = std::string(path) + ".ready";
    if (rank == 0) {
      if (capi_comm_unique_id(uid) != CAPI_OK) throw std::runtime_error("capi_comm_unique_id failed");
      FILE* f = fopen(path, "wb");
      if (!f || fwrite(uid, 1, 128, f) != 128) throw std::runtime_error("cannot write CAPITAL_UID_FILE");
      fclose(f);
      f = fopen(done.c_str(), "wb");
      if (f) fclose(f);
    } else {
      for (int tries = 0; tries < 6000; ++tries) {
        FILE* r = fopen(done.c_str(), "rb");
        if (r) { fclose(r); break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
      FILE* f = fopen(path, "rb");
      if (!f || fread(uid, 1, 128, f) != 128) throw std::runtime_error("cannot read CAPITAL_UID_FILE");
      fclose(f);
    }
  }
  capital::init(device, rank, size, size > 1 ? uid : nullptr);
}

// MPI_Barrier + MPI_Wtime of the reference benches: device sync, then a tiny all-reduce as the barrier
inline void barrier() {
  capital::sync();
  if (capital::ctx().size > 1) {
    double one = 1.0;
    double* d = capital::dev_alloc(1);
    CAPITAL_CHECK(capi_memcpy_h2d(capital::handle(), d, &one, sizeof(double)));
    CAPITAL_CHECK(capi_allreduce_sum(capital::world(), d, 1));
    capital::sync();
    capital::dev_free(d);
  }
}
inline double wtime() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline double max_over_ranks(double v) {
  if (capital::ctx().size == 1) return v;
  // max via sum of one-hot slots keeps to the allreduce(sum) the C-ABI exposes
  const int n = capital::ctx().size;
  std::vector<double> slots(n, 0.0);
  slots[capital::ctx().rank] = v;
  double* d = capital::dev_alloc(n);
  CAPITAL_CHECK(capi_memcpy_h2d(capital::handle(), d, slots.data(), sizeof(double) * n));
  CAPITAL_CHECK(capi_allreduce_sum(capital::world(), d, n));
  CAPITAL_CHECK(capi_memcpy_d2h(capital::handle(), slots.data(), d, sizeof(double) * n));
  capital::dev_free(d);
  double m = slots[0];
  for (double s : slots) m = std::max(m, s);
  return m;
}

}  // namespace capital_bench

#endif  // CAPITAL_BENCH_LAUNCH_H_
