// bench/launch.h -- process bootstrap shared by the two bench mains: what MPI_Init_thread + MPI_COMM_WORLD did in the
// reference benches.  One process per GPU.  Single process: nothing to configure.  Several processes on one node:
// the launcher sets RANK, WORLD_SIZE and LOCAL_RANK (torchrun / mpiexec style) and CAPITAL_UID_FILE to a path on a
// shared filesystem; rank 0 writes the 128-byte RCCL unique id there and the others pick it up.
#ifndef CAPITAL_BENCH_LAUNCH_H_
#define CAPITAL_BENCH_LAUNCH_H_

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../src/util/shared.h"

namespace capital_bench {

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// ---- file rendezvous of the 128-byte RCCL unique id ------------------------------------------------------------------
// The id must be THIS launch's: ranks that pick up a file left by an earlier (crashed, or merely previous) run hand
// ncclCommInitRank mismatched ids and hang on the GPUs.  Hence a handshake in which every file is validated by a token that
// only the current launch can know, every file appears atomically (temp + rename), every wait is bounded and ends in an
// exception (non-zero exit), and rank 0 removes the files once the communicator is up:
//   rank r > 0 : draws a random token, publishes  <path>.<run>.hello.<r>  = token   (re-publishes it if it disappears)
//   rank 0     : consumes the hello files; once it holds a token per rank it publishes  <path>.<run>  = id + tokens
//   rank r > 0 : accepts the id file only if ITS token is in it, then publishes  <path>.<run>.ack.<r> = token + head of the id
//   rank 0     : proceeds when every ack carries the token it published for that rank AND the head of THIS launch's id (a
//                hello/ack pair left by one earlier launch shares a token, but cannot know the new id); a newer hello (it
//                had consumed a stale one) replaces the token and the id file is rewritten.
// <run> = CAPITAL_RUN_ID, else TORCHELASTIC_RUN_ID, else MASTER_PORT, else "0" -- a launcher-provided nonce keeps
// concurrent launches that share a path apart; correctness against stale files does not depend on it.
namespace detail {
inline bool read_file(const std::string& p, std::vector<unsigned char>& out) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return false;
  unsigned char buf[4096];
  out.clear();
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
  fclose(f);
  return true;
}
inline void write_atomic(const std::string& p, const void* data, size_t bytes) {
  const std::string tmp = p + ".tmp." + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f || fwrite(data, 1, bytes, f) != bytes) { if (f) fclose(f); throw std::runtime_error("rendezvous: cannot write " + tmp); }
  fclose(f);
  if (rename(tmp.c_str(), p.c_str()) != 0) throw std::runtime_error("rendezvous: cannot rename to " + p);
}
inline uint64_t random_token() {
  uint64_t t = 0;
  FILE* f = fopen("/dev/urandom", "rb");
  if (f) { if (fread(&t, 1, sizeof(t), f) != sizeof(t)) t = 0; fclose(f); }
  if (!t) t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull ^ (uint64_t)getpid();
  return t ? t : 1;
}
}  // namespace detail

inline void rendezvous_uid(const std::string& path, int rank, int size, unsigned char (&uid)[128], double timeout_s = 120.0) {
  using clock = std::chrono::steady_clock;
  const char* run = getenv("CAPITAL_RUN_ID");
  if (!run) run = getenv("TORCHELASTIC_RUN_ID");
  if (!run) run = getenv("MASTER_PORT");
  const std::string base = path + "." + (run ? run : "0");
  auto hello = [&](int r) { return base + ".hello." + std::to_string(r); };
  auto ack = [&](int r) { return base + ".ack." + std::to_string(r); };
  const auto t_end = clock::now() + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(timeout_s));
  auto expired = [&] { return clock::now() > t_end; };
  auto nap = [] { std::this_thread::sleep_for(std::chrono::milliseconds(5)); };
  std::vector<unsigned char> buf;
  if (rank == 0) {
    if (capi_comm_unique_id(uid) != CAPI_OK) throw std::runtime_error("capi_comm_unique_id failed");
    std::vector<uint64_t> tok(size, 0);
    std::vector<bool> acked(size, false);
    bool dirty = true;
    for (;;) {
      for (int r = 1; r < size; ++r) {
        if (detail::read_file(hello(r), buf) && buf.size() == 8) {
          uint64_t t; memcpy(&t, buf.data(), 8);
          remove(hello(r).c_str());
          if (t != tok[r]) { tok[r] = t; acked[r] = false; dirty = true; }
        }
      }
      bool all = true;
      for (int r = 1; r < size; ++r) all = all && tok[r] != 0;
      if (all && dirty) {
        std::vector<unsigned char> img(128 + 8 * (size_t)size, 0);
        memcpy(img.data(), uid, 128);
        memcpy(img.data() + 128, tok.data(), 8 * (size_t)size);
        detail::write_atomic(base, img.data(), img.size());
        dirty = false;
      }
      bool done = all && !dirty;
      for (int r = 1; r < size && done; ++r) {
        if (!acked[r] && detail::read_file(ack(r), buf) && buf.size() == 16) {
          // token AND the head of the id it accepted: an ack left by an earlier launch carries that launch's id
          acked[r] = memcmp(buf.data(), &tok[r], 8) == 0 && memcmp(buf.data() + 8, uid, 8) == 0;
        }
        done = acked[r];
      }
      if (done) break;
      if (expired()) throw std::runtime_error("rendezvous: rank 0 timed out waiting for the other ranks at " + base);
      nap();
    }
  } else {
    const uint64_t mine = detail::random_token();
    remove(ack(rank).c_str());
    detail::write_atomic(hello(rank), &mine, 8);
    for (;;) {
      if (detail::read_file(base, buf) && buf.size() == 128 + 8 * (size_t)size) {
        uint64_t t; memcpy(&t, buf.data() + 128 + 8 * (size_t)rank, 8);
        if (t == mine) { memcpy(uid, buf.data(), 128); break; }
      }
      FILE* f = fopen(hello(rank).c_str(), "rb");          // consumed by rank 0 without our token in the id file (yet): publish again
      if (f) fclose(f); else detail::write_atomic(hello(rank), &mine, 8);
      if (expired()) throw std::runtime_error("rendezvous: rank " + std::to_string(rank) + " timed out waiting for this launch's id at " + base);
      nap();
    }
    unsigned char a16[16];
    memcpy(a16, &mine, 8);
    memcpy(a16 + 8, uid, 8);
    detail::write_atomic(ack(rank), a16, 16);
  }
}

// rank 0, once the communicator is up (every rank has read the id by then): leave nothing behind for a later launch
inline void rendezvous_cleanup(const std::string& path, int size) {
  const char* run = getenv("CAPITAL_RUN_ID");
  if (!run) run = getenv("TORCHELASTIC_RUN_ID");
  if (!run) run = getenv("MASTER_PORT");
  const std::string base = path + "." + (run ? run : "0");
  remove(base.c_str());
  for (int r = 1; r < size; ++r) { remove((base + ".hello." + std::to_string(r)).c_str()); remove((base + ".ack." + std::to_string(r)).c_str()); }
}

inline void init(int& rank, int& size) {
  rank = env_int("RANK", env_int("PMI_RANK", 0));
  size = env_int("WORLD_SIZE", env_int("PMI_SIZE", 1));
  const int device = env_int("LOCAL_RANK", rank);
  unsigned char uid[128] = {0};
  const char* path = getenv("CAPITAL_UID_FILE");
  if (size > 1) {
    if (!path) throw std::runtime_error("WORLD_SIZE > 1 needs CAPITAL_UID_FILE (shared path for the RCCL unique id)");
    rendezvous_uid(path, rank, size, uid, (double)env_int("CAPITAL_RENDEZVOUS_TIMEOUT_S", 120));
  }
  capital::init(device, rank, size, size > 1 ? uid : nullptr);
  if (size > 1 && rank == 0 && !getenv("CAPITAL_KEEP_UID_FILES")) rendezvous_cleanup(path, size);
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
