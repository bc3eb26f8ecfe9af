// ring_place.h -- TEST INFRASTRUCTURE: the ring bookkeeping of the asynchronous loopback transport (loopback_async.hip), host-only C++ so that
// tests/test_loopback_ring.py can check its flow-control rule on the CPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <vector>

namespace lb_async {

struct Msg { uint32_t seq; size_t off, end; };
struct Ring {                  // one direction of one channel, as THIS side sees it
  uint32_t gen = 0, seq = 0;
  size_t capacity = 0, head = 0;
  char* base = nullptr;        // sender: its own allocation; receiver: the mapping (or, for a peer in this very process, the sender's pointer)
  bool base_local = false;     // receiver: `base` is a same-process pointer, nothing to close
  std::deque<Msg> live;
  std::vector<char*> retired;  // earlier generations: released at detach (messages may still be in flight in them)
  std::vector<char*> by_gen;   // sender: ring base of every generation (a receiver thread of the same process asks for it by generation)
};

// The placement rule both sides run.  Returns the offset of the message and, through wait_seq, the newest earlier message whose region it
// overlaps (0: none) -- the sender must not write before that one has been consumed.  grew: a new generation starts with this message.
inline size_t ring_place(Ring& r, size_t bytes, uint32_t& wait_seq, bool& grew) {
  const size_t need = ((bytes + 255) & ~(size_t)255) + 256;      // the message and a 256-byte trailer (its checksum, written by the sender: loopback_async.hip lb_sum_kernel)
  grew = false;
  wait_seq = 0;
  if (4 * need > r.capacity || r.gen == 0) {
    // (CAPI_LOOPBACK_MIN_RING_MB: diagnostics -- rings that never have to grow)
    static const size_t floor_ = [] { const char* e = getenv("CAPI_LOOPBACK_MIN_RING_MB"); return e ? (size_t)atol(e) << 20 : (size_t)1 << 20; }();
    size_t cap = floor_ ? floor_ : (size_t)1 << 20;
    while (cap < 4 * need) cap <<= 1;
    if (cap < r.capacity) cap = r.capacity;
    r.capacity = cap;
    r.head = 0;
    r.live.clear();
    ++r.gen;
    grew = true;
  }
  if (r.head + need > r.capacity) r.head = 0;
  const size_t off = r.head, end = off + need;
  // Every live message that lies in the way, wherever it stands in the list: after a wrap the messages left over at the ring's far end are OLDER than
  // the ones at its start and do not overlap the new region -- stopping at the first message that is clear of it (as this loop first did) let a
  // wrapped message overwrite unconsumed ones behind such a leftover.  Found by the eight-rank rehearsal: an intermittent wrong factor in the
  // second case of a process, never in a fresh one.  Consumption is in order, so waiting for the newest one in the way covers all before it.
  for (const Msg& m : r.live)
    if (m.off < end && m.end > off && m.seq > wait_seq) wait_seq = m.seq;
  while (!r.live.empty() && r.live.front().seq <= wait_seq) r.live.pop_front();
  ++r.seq;
  r.live.push_back({r.seq, off, end});
  r.head = end;
  return off;
}

}  // namespace lb_async
