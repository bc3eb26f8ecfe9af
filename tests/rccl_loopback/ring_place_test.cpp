// ring_place_test.cpp -- CPU property test of the asynchronous loopback transport's ring bookkeeping (ring_place.h): a sender places messages of random
// sizes while a receiver consumes them in order with a random lag; before a message's region is written the sender waits until `wait_seq` has been
// consumed.  Property: at that moment no unconsumed message of the same ring generation overlaps the region.  (The first version of ring_place
// stopped scanning at the first live message that was clear of the region and failed this after a wrap: tests/test_loopback_ring.py.)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "ring_place.h"

int main(int argc, char** argv) {
  const unsigned seed0 = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
  long checked = 0, waits = 0;
  for (unsigned seed = seed0; seed < seed0 + 200; ++seed) {
    std::mt19937_64 rng(seed);
    lb_async::Ring s, r;                       // sender's and receiver's view: they must place identically
    struct Rec { uint32_t seq, gen; size_t off, end; };
    std::vector<Rec> sent;                     // every message, in order
    uint32_t consumed = 0;                     // the counter the device would hold
    const size_t big = (size_t)1 << (18 + seed % 6);
    for (int k = 0; k < 4000; ++k) {
      // sizes: mostly small with occasional large ones (what the schedules send: pieces, panels, whole blocks)
      size_t bytes = (rng() % 8 == 0) ? (size_t)(rng() % big) + 1 : (size_t)(rng() % (big / 16 + 1)) + 1;
      uint32_t ws = 0, wr = 0; bool gs = false, gr = false;
      const size_t off = lb_async::ring_place(s, bytes, ws, gs), off_r = lb_async::ring_place(r, bytes, wr, gr);
      if (off != off_r || gs != gr || s.gen != r.gen) { printf("FAIL: sender and receiver disagree at message %d (seed %u)\n", k, seed); return 1; }
      if (ws > consumed) { consumed = ws; ++waits; }                         // the sender's device-side wait
      const size_t end = s.live.back().end;
      if (end > s.capacity) { printf("FAIL: message %d leaves the ring (seed %u)\n", k, seed); return 1; }
      for (const Rec& m : sent)
        if (m.seq > consumed && m.gen == s.gen && m.off < end && m.end > off) {
          printf("FAIL: message %u [%zu, %zu) would be written over unconsumed message %u [%zu, %zu) (seed %u, consumed %u)\n", s.seq, off, end, m.seq, m.off, m.end, seed, consumed);
          return 1;
        }
      sent.push_back({s.seq, s.gen, off, end});
      ++checked;
      // the receiver consumes in order and, in phases, more slowly than the sender sends: the lag grows until the ring's flow control (the wait
      // above) is what paces the sender -- the regime in which a lost wait corrupts data
      const bool slow = ((k / 500) & 1) != 0;
      const uint32_t step = slow ? (uint32_t)(rng() % 4 == 0) : (uint32_t)(rng() % 3);
      consumed = consumed + step < s.seq ? consumed + step : s.seq;
      size_t drop = 0;
      while (drop < sent.size() && sent[drop].seq <= consumed) ++drop;
      sent.erase(sent.begin(), sent.begin() + drop);
    }
  }
  printf("ok: %ld placements checked, %ld of them waited\n", checked, waits);
  return 0;
}
