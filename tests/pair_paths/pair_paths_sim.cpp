// pair_paths_sim.cpp -- TEST PROGRAM (tests/test_pair_paths.py): the product's multi-path transfer algorithm (capital_amd/csrc/pair_paths.h)
// run by N rank threads over an in-memory transport with RCCL's point-to-point semantics (messages of one ordered pair match in posting
// order; a group's sends are posted before its receives are awaited).  Checks, for transfer sets of the kinds the grid issues (pair
// broadcasts, exchanges, depth halves) and for random ones: every destination receives exactly its source's data; and the property the
// design rests on -- in each of the two phases every DIRECTED LINK of the mesh carries at most one message, of at most one unit.
// usage: pair_paths_sim <nranks> <count> <min_multipath_count> <set>   set: row | column | depth | transpose | random<seed>
// prints one JSON line.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../capital_amd/csrc/pair_paths.h"

namespace {
struct Mail {
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<std::vector<double>>> box;      // (src, dst) -> FIFO
  std::map<std::pair<int, int>, std::vector<int64_t>> log[2];               // phase -> (src, dst) -> message lengths
} g;

struct Transport {
  int me;
  int phase = -1;
  struct Op { const double* s; double* r; int64_t n; int peer; };
  std::vector<Op> ops;
  int group_begin() { ops.clear(); ++phase; return 0; }
  void send(const double* p, int64_t n, int peer) { ops.push_back({p, nullptr, n, peer}); }
  void recv(double* p, int64_t n, int peer) { ops.push_back({nullptr, p, n, peer}); }
  int group_end() {
    {
      std::lock_guard<std::mutex> l(g.m);
      for (auto& o : ops)
        if (o.s) {
          g.box[{me, o.peer}].emplace_back(o.s, o.s + o.n);
          g.log[phase > 1 ? 1 : phase][{me, o.peer}].push_back(o.n);
        }
    }
    g.cv.notify_all();
    for (auto& o : ops)
      if (o.r) {
        std::unique_lock<std::mutex> l(g.m);
        auto& q = g.box[{o.peer, me}];
        if (!g.cv.wait_for(l, std::chrono::seconds(20), [&] { return !q.empty(); })) return 99;
        if ((int64_t)q.front().size() != o.n) return 98;                    // a length mismatch is a protocol error
        memcpy(o.r, q.front().data(), sizeof(double) * (size_t)o.n);
        q.pop_front();
      }
    return 0;
  }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int n = atoi(argv[1]);
  const int64_t count = atoll(argv[2]), minc = atoll(argv[3]);
  const std::string set = argv[4];
  std::vector<int> dst((size_t)n, -1);
  // a 2 x 2 x c grid in layout 0 (rank = z + c x + c d y), as topo::square numbers it; c = n / 4 (n = 4: c = 1, n = 8: c = 2)
  const int d = n >= 4 ? 2 : 1, c = n / (d * d);
  auto rank_of = [&](int x, int y, int z) { return z + c * x + c * d * y; };
  if (set == "row" || set == "column") {
    for (int y = 0; y < d; ++y) for (int x = 0; x < d; ++x) for (int z = 0; z < c; ++z) {
      const int q = z % d;
      if ((set == "row" ? x : y) == q) dst[(size_t)rank_of(x, y, z)] = set == "row" ? rank_of(1 - x, y, z) : rank_of(x, 1 - y, z);
    }
  } else if (set == "depth") {
    if (c != 2) return 2;
    for (int y = 0; y < d; ++y) for (int x = 0; x < d; ++x) for (int z = 0; z < c; ++z) dst[(size_t)rank_of(x, y, z)] = rank_of(x, y, 1 - z);
  } else if (set == "transpose") {
    for (int y = 0; y < d; ++y) for (int x = 0; x < d; ++x) for (int z = 0; z < c; ++z) if (x != y) dst[(size_t)rank_of(x, y, z)] = rank_of(y, x, z);
  } else {
    std::mt19937 rng((unsigned)atoi(set.c_str() + 6));
    std::vector<int> perm((size_t)n);
    for (int i = 0; i < n; ++i) perm[(size_t)i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int i = 0; i < n; ++i) if (perm[(size_t)i] != i && (rng() & 3)) dst[(size_t)i] = perm[(size_t)i];      // a partial permutation without fixed points
  }
  std::vector<std::vector<double>> sendb((size_t)n), recvb((size_t)n), scratch((size_t)n);
  for (int r = 0; r < n; ++r) {
    sendb[(size_t)r].resize((size_t)count);
    for (int64_t i = 0; i < count; ++i) sendb[(size_t)r][(size_t)i] = 1000.0 * r + (double)i;
    recvb[(size_t)r].assign((size_t)count, -1.0);
    scratch[(size_t)r].assign((size_t)pair_paths::scratch_count(n, count) + 2, 0.0);
  }
  std::vector<int> rc((size_t)n, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < n; ++r)
    th.emplace_back([&, r] {
      Transport x{r};
      rc[(size_t)r] = pair_paths::transfer(x, r, n, dst.data(), sendb[(size_t)r].data(), recvb[(size_t)r].data(), count, scratch[(size_t)r].data(), minc);
    });
  for (auto& t : th) t.join();
  int bad_rc = 0, wrong = 0, transfers = 0;
  for (int r = 0; r < n; ++r) bad_rc += rc[(size_t)r] != 0;
  for (int a = 0; a < n; ++a) {
    const int b = dst[(size_t)a];
    if (b < 0) continue;
    ++transfers;
    for (int64_t i = 0; i < count; ++i) wrong += recvb[(size_t)b][(size_t)i] != sendb[(size_t)a][(size_t)i];
  }
  int64_t leftover = 0;
  for (auto& kv : g.box) leftover += (int64_t)kv.second.size();
  const int64_t unit = pair_paths::unit_len(count, n);
  int max_msgs_per_link = 0; int64_t max_len = 0, links_used[2] = {0, 0};
  for (int ph = 0; ph < 2; ++ph)
    for (auto& kv : g.log[ph]) {
      ++links_used[ph];
      if ((int)kv.second.size() > max_msgs_per_link) max_msgs_per_link = (int)kv.second.size();
      for (int64_t l : kv.second) if (l > max_len) max_len = l;
    }
  printf("{\"nranks\": %d, \"count\": %lld, \"transfers\": %d, \"bad_rc\": %d, \"wrong\": %d, \"leftover\": %lld, \"unit\": %lld, \"max_msgs_per_link_per_phase\": %d, "
         "\"max_message\": %lld, \"links_phase1\": %lld, \"links_phase2\": %lld}\n",
         n, (long long)count, transfers, bad_rc, wrong, (long long)leftover, (long long)unit, max_msgs_per_link, (long long)max_len,
         (long long)links_used[0], (long long)links_used[1]);
  return (bad_rc || wrong || leftover) ? 1 : 0;
}
