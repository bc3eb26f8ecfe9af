// pair_paths.h -- grid-wide sets of point-to-point transfers that use EVERY xGMI link of the node, not one per pair.
//
// The 3-D SUMMA of the reference moves its operands between PAIRS of ranks on a 2 x 2 x 2 grid: MPI_Bcast over a row or a column
// of two (summa.hpp:185,193), MPI_Allreduce over a depth fibre of two (summa.hpp:236), MPI_Sendrecv_replace with the transpose
// partner (util.hpp:240) -- 1 to 2 GiB each at n = 65536.  On MI355X the 8 GPUs of a node are a full mesh of point-to-point xGMI
// links (7 per GPU); a transfer between two GPUs as one RCCL call on a 2-rank communicator travels over ONE of them while the
// sender's other six stand idle.  Here every rank of the WORLD communicator takes part in every transfer: a message a -> b is cut
// into n units (n = ranks of the node); two travel directly, one in each of two phases, and each of the other n - 2 ranks relays
// one unit (a -> k in phase 1, k -> b in phase 2).  All transfers of a SUMMA step (four pair broadcasts, or eight halves of four
// pair exchanges) run together, and every directed link of the mesh carries at most one unit per phase:
//     time = 2 units = 2/n of the message per link, against the whole message on one link  (n = 8: 4 x, n = 4: 2 x).
//
// The algorithm is written once, over a transport with RCCL's point-to-point vocabulary (group_begin / send / recv / group_end,
// messages between one ordered pair matched in posting order): capital_amd/csrc/comm_rccl.hip instantiates it with ncclSend /
// ncclRecv on the handle's stream, tests/cpu_shim with its gloo-backed callback -- the CPU rehearsals execute THIS code.
#ifndef CAPITAL_PAIR_PATHS_H_
#define CAPITAL_PAIR_PATHS_H_

#include <cstdint>
#include <vector>

namespace pair_paths {

// unit length in doubles: n units cover `count`, each a multiple of 2 doubles (16-byte alignment of every piece)
inline int64_t unit_len(int64_t count, int n) {
  int64_t u = (count + n - 1) / n;
  return (u + 1) & ~(int64_t)1;
}
// relay space a rank needs for one call: one unit per transfer that it neither sends nor receives
inline int64_t scratch_count(int n, int64_t count) { return n <= 2 ? 0 : (int64_t)n * unit_len(count, n); }

// dst[r], r = 0 .. n-1 (the same array on every rank): the rank that r sends its `count` doubles to, or -1.  No rank is the
// destination of two transfers and nobody sends to itself (except the one rank of a 1-rank communicator).  Returns 0, or -1 for an
// invalid transfer set.
template <class X>
int transfer(X& x, int me, int n, const int* dst, const double* send, double* recv, int64_t count, double* scratch, int64_t multipath_min_count) {
  if (n < 1 || me < 0 || me >= n || count < 0) return -1;
  std::vector<int> src((size_t)n, -1);
  for (int r = 0; r < n; ++r) {
    const int b = dst[r];
    if (b < 0) continue;
    if (b >= n || (b == r && n > 1) || src[(size_t)b] >= 0) return -1;      // (a lone rank may send to itself: one send + one receive through the transport)
    src[(size_t)b] = r;
  }
  if (count == 0) return 0;
  const int to = dst[me], from = src[(size_t)me];
  if (to >= 0 && !send) return -1;
  if (from >= 0 && !recv) return -1;
  // small messages (latency-bound) and 2-rank nodes: one direct message per transfer
  if (n <= 2 || count < multipath_min_count) {
    if (to < 0 && from < 0) return 0;
    int rc = x.group_begin();
    if (rc) return rc;
    if (to >= 0) x.send(send, count, to);
    if (from >= 0) x.recv(recv, count, from);
    return x.group_end();
  }
  if (!scratch) return -1;
  const int64_t u = unit_len(count, n);
  auto off = [&](int j) { const int64_t o = (int64_t)j * u; return o < count ? o : count; };
  auto len = [&](int j) { return off(j + 1) - off(j); };
  // unit of relay k in transfer a -> b: 2 + (position of k among the ranks other than a and b, ascending)
  auto relay_unit = [&](int a, int b, int k) { int pos = k; if (k > a) --pos; if (k > b) --pos; return 2 + pos; };
  // phase 1: sources fan their units out (unit 0 straight to the destination), everyone else takes in what it will relay
  int rc = x.group_begin();
  if (rc) return rc;
  if (to >= 0) {
    if (len(0) > 0) x.send(send + off(0), len(0), to);
    for (int k = 0; k < n; ++k)
      if (k != me && k != to) { const int j = relay_unit(me, to, k); if (len(j) > 0) x.send(send + off(j), len(j), k); }
  }
  if (from >= 0 && len(0) > 0) x.recv(recv + off(0), len(0), from);
  int slot = 0;
  for (int a = 0; a < n; ++a) {
    const int b = dst[a];
    if (b < 0 || a == me || b == me) continue;
    const int j = relay_unit(a, b, me);
    if (len(j) > 0) x.recv(scratch + (int64_t)slot * u, len(j), a);
    ++slot;
  }
  rc = x.group_end();
  if (rc) return rc;
  // phase 2: the second direct unit, and the relays hand their units on
  rc = x.group_begin();
  if (rc) return rc;
  if (to >= 0 && len(1) > 0) x.send(send + off(1), len(1), to);
  if (from >= 0) {
    if (len(1) > 0) x.recv(recv + off(1), len(1), from);
    for (int k = 0; k < n; ++k)
      if (k != me && k != from) { const int j = relay_unit(from, me, k); if (len(j) > 0) x.recv(recv + off(j), len(j), k); }
  }
  slot = 0;
  for (int a = 0; a < n; ++a) {
    const int b = dst[a];
    if (b < 0 || a == me || b == me) continue;
    const int j = relay_unit(a, b, me);
    if (len(j) > 0) x.send(scratch + (int64_t)slot * u, len(j), b);
    ++slot;
  }
  return x.group_end();
}

}  // namespace pair_paths

#endif  // CAPITAL_PAIR_PATHS_H_
