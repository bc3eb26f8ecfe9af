// alg/cholesky/cholinv/cholinv.h -- recursive Cholesky with triangular inverse on MI355X
// (reference src/alg/cholesky/cholinv/cholinv.h:11-75, cholinv.hpp:6-183).
//
// Same call surface: cholinv<SerializePolicy,IntermediatesPolicy,BaseCasePolicy>::factor(A, args, topo),
// construct_R / construct_Rinv, and the info<T,U> pack (complete_inv, split, bc_mult_dim, dir; R, Rinv).
// Same schedule per recursion level (cholinv.hpp:107-155):
//   1. recurse on the leading block                         -> R11, R11^-1
//   2. CI::trsm   R12 = R11^-T * A12                        (trmm Left/Upper/Trans SUMMA)
//   3. CI::tmu    A22 <- A22 - R12^T R12                    (triangular-output SUMMA)
//   4. recurse on the trailing block                        -> R22, R22^-1
//   5. CI::tmu    R^-1_12 = -R11^-1 R12 R22^-1              (two trmm SUMMAs; skipped at the top level when !complete_inv)
// What is different underneath: blocks are views into the device-resident R / R^-1 (no serialize copies into per-level
// tables), every multiply runs on the fp64 MFMA tile kernel, step 3 computes only the upper triangle, and the base case
// is one fused potrf+trtri launch sequence on the aggregated block.
#ifndef CAPITAL_CHOLESKY_CHOLINV_H_
#define CAPITAL_CHOLESKY_CHOLINV_H_

#include "./../../alg.h"
#include "./../../matmult/summa/summa.h"
#include "./policy.h"

namespace cholesky {

template <class SerializePolicy = policy::cholinv::Serialize, class IntermediatesPolicy = policy::cholinv::SaveIntermediates,
          class BaseCasePolicy = policy::cholinv::NoReplication>
class cholinv : public SerializePolicy, public IntermediatesPolicy, public BaseCasePolicy {
public:
  using SP = SerializePolicy;
  using IP = IntermediatesPolicy;
  using BP = BaseCasePolicy;

  template <typename ScalarT, typename DimensionT>
  class info {
  public:
    using ScalarType = ScalarT;
    using DimensionType = DimensionT;
    using alg_type = cholinv<SerializePolicy, IntermediatesPolicy, BaseCasePolicy>;
    using SP = SerializePolicy;
    using IP = IntermediatesPolicy;
    using BP = BaseCasePolicy;
    info(const info& p) : complete_inv(p.complete_inv), split(p.split), bc_mult_dim(p.bc_mult_dim), dir(p.dir) {}
    info(info&& p) : complete_inv(p.complete_inv), split(p.split), bc_mult_dim(p.bc_mult_dim), dir(p.dir) {}
    info(DimensionType complete_inv, DimensionType split, DimensionType bc_mult_dim, char dir)
        : complete_inv(complete_inv), split(split), bc_mult_dim(bc_mult_dim), dir(dir) {}
    // user input (cholinv.h:50-53)
    const DimensionType complete_inv, split, bc_mult_dim;
    const char dir;
    // factors (cholinv.h:55-56)
    matrix<ScalarType, DimensionType, typename SerializePolicy::structure> R, Rinv;
    // full-storage working images when the returned structure is packed
    matrix<ScalarType, DimensionType, rect> Rfull, Rinvfull;
    matmult::arena work;
    DimensionType localDimension = 0, globalDimension = 0, trueLocalDimension = 0, trueGlobalDimension = 0, bcDimension = 0;
    // bookkeeping exposed for tests / benches
    int64_t num_base_cases = 0, num_levels = 0;
    // TRSM mode (not in the reference, which has no TRSM anywhere; BASELINE north_star names the kernel): factor() runs the
    // textbook right-looking recursion -- potrf on the diagonal block, R12 = R11^-T A12 by a block TRSM, trailing SYRK -- and
    // forms NO inverse: n^3/3 executed flops instead of 5 n^3/12.  R is the same factor; Rinv is not formed (construct_Rinv
    // throws).  One GPU, or a d x d x c grid (potrf_rec_grid).  Default off: the reference-exact schedule.
    bool solve_with_trsm = false;
    // LAPACK info of the factorisation (the reference drops it, lapack/interface.hpp:39,54): 0, or the 1-based position,
    // inside the first diagonal block that failed, of the first non-positive pivot.  factor() throws std::domain_error then.
    int potrf_info = 0;
    bool zeroed = false;
    // top-level overlap state: the input's right part is copied, and the finished left part packed, on the second stream
    const double* input = nullptr;
    DimensionType early_split = 0;
    DimensionType input_lead = 0;       // > 0: only this leading block of the input's left half was copied on the compute stream
    // trailing updates currently running beside the recursion on the bulk streams (single-GPU lookahead)
    int la_depth = 0;
  };

  template <typename MatrixType, typename ArgType, typename CommType>
  static void factor(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    CRITTER_START(CI::factor);
    using U = typename ArgType::DimensionType;
    static_assert(std::is_same<typename MatrixType::StructureType, rect>::value, "cholinv::factor takes a rect-structured input block");
    if (!(args.split > 0) || args.dir != 'U') throw std::invalid_argument("cholinv: split > 0 and dir == 'U' required (cholinv.hpp:9)");
    if (CommInfo.d > 1 && CommInfo.d % CommInfo.c) throw std::invalid_argument("cholinv: c must divide d (or d == 1)");
    if (args.solve_with_trsm) { factor_trsm(A, args, CommInfo); CRITTER_STOP(CI::factor); return; }
    const U localDimension = A.num_rows_local(), globalDimension = A.num_rows_global();
    CAPITAL_CHECK(capi_stream_select(capital::handle(), 0));    // (a call that threw mid-way may have left another stream selected)
    CAPITAL_CHECK(capi_reset_info(capital::handle()));
    args.potrf_info = 0;
    args.R._register_(A.num_columns_global(), A.num_rows_global(), CommInfo.d, CommInfo.d);
    args.Rinv._register_(A.num_columns_global(), A.num_rows_global(), CommInfo.d, CommInfo.d);
    constexpr bool packed = !std::is_same<typename SP::structure, rect>::value;
    if (packed) {
      args.Rfull._register_(A.num_columns_global(), A.num_rows_global(), CommInfo.d, CommInfo.d);
      args.Rinvfull._register_(A.num_columns_global(), A.num_rows_global(), CommInfo.d, CommInfo.d);
    }
    double* R = packed ? args.Rfull.data() : (double*)args.R.data();
    double* Ri = packed ? args.Rinvfull.data() : (double*)args.Rinv.data();
    const U ld = localDimension;
    // cholinv.hpp:13: the upper triangle of the input becomes the working R; the rest of R and all of R^-1 are zero
    // Freshly registered blocks are zero (matrix::allocate) and nothing below ever stores a non-zero where the result
    // must be zero (strictly-lower parts; the skipped R^-1_12 at the top level), so later calls do not re-zero 2 n_loc^2.
    if (!args.zeroed) {
      capital::dev_zero(R, ld * ld);
      capital::dev_zero(Ri, ld * ld);
      args.zeroed = true;
    }
    // base-case size rule, cholinv.hpp:15-18
    U bcDimLocal = (U)(CommInfo.c * CommInfo.d);
    U bcMult = args.bc_mult_dim;
    if (bcMult < 0) { bcMult = -bcMult; for (U i = 0; i < bcMult; ++i) bcDimLocal *= 2; } else { for (U i = 0; i < bcMult; ++i) bcDimLocal /= 2; }
    bcDimLocal = std::max<U>(1, bcDimLocal);
    bcDimLocal = std::min<U>(localDimension, bcDimLocal);
    bcDimLocal = localDimension / bcDimLocal;
    args.localDimension = args.trueLocalDimension = localDimension;
    args.globalDimension = args.trueGlobalDimension = globalDimension;
    args.bcDimension = (U)CommInfo.d * bcDimLocal;
    args.num_base_cases = args.num_levels = 0;

    // Only the leading block is needed to start the left half's recursion (a chain of latency-bound kernels): the
    // rest of the input follows on the second stream and is awaited just before the top level's R12 product.
    const U h1 = localDimension >> args.split;
    args.input = A.data();
    args.early_split = 0;
    const bool will_split = !(((localDimension * (U)CommInfo.d) <= args.bcDimension) || (h1 < args.split));   // cholinv.hpp:93
    args.input_lead = 0;
    if (will_split && h1 > 0 && h1 < ld) {
      capi_handle_t hh = capital::handle();
      // the left spine starts on the leading blocks: when the recursion halves cleanly down to `lead`, only that block is
      // copied ahead of it; the rest of the left half follows on the second stream and is awaited by the node of order 2 lead
      U lead = h1;
      if (args.split == 1 && CommInfo.d == 1 && CommInfo.c == 1) while (lead >= 8192 && lead % 2 == 0 && (lead >> 1) * (U)CommInfo.d > args.bcDimension) lead >>= 1;
      CAPITAL_CHECK(capi_dlacpy(hh, 1, lead, lead, A.data(), ld, R, ld));
      CAPITAL_CHECK(capi_event_record(hh, EV_INPUT_HEAD));
      CAPITAL_CHECK(capi_stream_select(hh, 1));
      CAPITAL_CHECK(capi_event_wait(hh, EV_INPUT_HEAD));       // (also orders this call after the previous call's packing)
      if (lead < h1) {
        CAPITAL_CHECK(capi_dlacpy(hh, 0, lead, h1 - lead, A.data() + (int64_t)lead * ld, ld, R + (int64_t)lead * ld, ld));
        CAPITAL_CHECK(capi_dlacpy(hh, 1, h1 - lead, h1 - lead, A.data() + lead + (int64_t)lead * ld, ld, R + lead + (int64_t)lead * ld, ld));
        CAPITAL_CHECK(capi_event_record(hh, EV_INPUT_MID));
        args.input_lead = lead;
      }
      CAPITAL_CHECK(capi_dlacpy(hh, 0, h1, ld - h1, A.data() + (int64_t)h1 * ld, ld, R + (int64_t)h1 * ld, ld));
      CAPITAL_CHECK(capi_dlacpy(hh, 1, ld - h1, ld - h1, A.data() + h1 + (int64_t)h1 * ld, ld, R + h1 + (int64_t)h1 * ld, ld));
      CAPITAL_CHECK(capi_event_record(hh, EV_INPUT_REST));
      CAPITAL_CHECK(capi_stream_select(hh, 0));
      args.early_split = h1;
    } else {
      CAPITAL_CHECK(capi_dlacpy(capital::handle(), 1, ld, ld, A.data(), ld, R, ld));
    }


    // the reference's simulate() (cholinv.hpp:50-83) pre-allocates every level's tables; here ONE arena covers the
    // deepest concurrent need: panels + partial sums of the top level, or the aggregated base case
    const bool single = (CommInfo.d == 1 && CommInfo.c == 1);
    // the large launches go out one resident round at a time: same time, half the L2-to-fabric traffic (on grids: bandwidth the collectives'
    // copy kernels share).  On one GPU the rounds of the lookahead's bulk streams interleave with each other and with the chain -- harmless,
    // but stream-ordered event brackets around a round then also contain its neighbours, which is why the roofline figure is taken from the
    // kernels' own interval stamps (capi_prof_collect_intervals).  CAPITAL_NO_LAUNCH_ROUNDS: one launch per product.
    capital::launch_rounds_scope rounds(true);
    const U h = localDimension - (localDimension >> args.split);
    const U agg = args.bcDimension;
    // (single GPU: the R12 copies of nested levels stay live while their trailing updates run beside the recursion: h^2 (1 + 1/4 + ...))
    // (grids: the update's W, its exchanged copy + staging, two panels, the partial sums + their packed image, relay space and the landing
    //  half of the multi-path pair transfers, per-chunk copies of a pipelined multiply)
    //  (the K-sliced grids, d == 1, broadcast and relay nothing: 8 h^2 as in round 2 -- at n = 65536 on 1 x 1 x 2 h^2 is 8 GiB)
    int64_t need = single ? (int64_t)h * h + (int64_t)h * h / 3 + 1024 : (int64_t)(CommInfo.d == 1 ? 8 : 12) * h * h + 4 * (int64_t)agg * agg + 4096;
    args.work.reserve(need);

    args.la_depth = 0;
    invoke(args, CommInfo, R, Ri, ld, (U)0, localDimension, globalDimension);

    if (packed) {
      const U e = args.early_split;     // columns [0,e) of both factors and rows [0,e) of R were packed during the right half
      if (e > 0) {
        pack_block(args, CAPI_UPPERTRI, R, (double*)args.R.data(), ld, e, ld, e, ld);                                     // R22
        pack_block(args, CAPI_UPPERTRI, Ri, (double*)args.Rinv.data(), ld, e, ld, e, ld);                                 // Rinv22
        if (args.complete_inv) pack_block(args, CAPI_RECT, Ri, (double*)args.Rinv.data(), ld, e, ld, 0, e);              // Rinv12
        CAPITAL_CHECK(capi_event_wait(capital::handle(), EV_EARLY_PACK));
      } else {
        serialize<uppertri, uppertri>::invoke(args.Rfull, args.R, 0, ld, 0, ld, 0, ld, 0, ld);
        serialize<uppertri, uppertri>::invoke(args.Rinvfull, args.Rinv, 0, ld, 0, ld, 0, ld, 0, ld);
      }
    }
    if (!IP::keep_arena) {
      // FlushIntermediates (policy.h:85-156: every level's buffers are released after use instead of cached in the tables): here
      // the intermediates are the work arena and, with Serialize, the two full-storage working images of the factors -- after
      // the call only what the caller asked for stays resident (n = 65536 on one GPU: 96 GiB instead of 171 GiB beside A).
      capital::sync();
      args.work.release();
      if (packed) { args.Rfull._destroy_(); args.Rinvfull._destroy_(); args.zeroed = false; }
    }
    CRITTER_STOP(CI::factor);
    // one 4-byte read behind the whole launch chain: a non-SPD input must not come back as NaN factors with status OK
    CAPITAL_CHECK(capi_get_info(capital::handle(), &args.potrf_info));
    // on a grid only the ranks that factored the failing aggregate hold its info word (layer 0 with ReplicateComp, the slice
    // root with NoReplication[Overlap]): the grid agrees before anyone unwinds, so every rank throws or none does
    const int failed = capital::ranks_with_nonzero(CommInfo.world, args.potrf_info);
    if (failed != 0)
      throw std::domain_error("cholinv::factor: the matrix is not positive definite (" +
                              (args.potrf_info != 0 ? "non-positive pivot " + std::to_string(args.potrf_info) + " of a diagonal block"
                                                    : std::string("reported by ") + std::to_string(failed) + " other rank(s) of the grid") +
                              "); R and Rinv are not valid");
  }

  // ---- TRSM mode -----------------------------------------------------------------------------------------------------------
  // R = chol(A) with the same split rule and base-case size as the reference recursion (cholinv.hpp:15-18,93,107), but
  //   2'. R12 = R11^-T A12 by capi_dtrsm (block TRSM: products on the MFMA tile kernel, 256-blocks inverted on the fly)
  //   3.  A22 <- A22 - R12^T R12 by capi_dsyrk (upper triangle only)
  // and no step 5; the base case is capi_dpotrf alone.  Nothing of R^-1 is formed.
  template <typename MatrixType, typename ArgType, typename CommType>
  static void factor_trsm(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    using U = typename ArgType::DimensionType;
    capi_handle_t h = capital::handle();
    const U ld = A.num_rows_local();
    CAPITAL_CHECK(capi_stream_select(h, 0));
    CAPITAL_CHECK(capi_reset_info(h));
    args.potrf_info = 0;
    args.R._register_(A.num_columns_global(), A.num_rows_global(), CommInfo.d, CommInfo.d);
    constexpr bool packed = !std::is_same<typename SP::structure, rect>::value;
    if (packed) args.Rfull._register_(A.num_columns_global(), A.num_rows_global(), CommInfo.d, CommInfo.d);
    double* R = packed ? args.Rfull.data() : (double*)args.R.data();
    if (!args.zeroed) { capital::dev_zero(R, ld * ld); args.zeroed = true; }
    CAPITAL_CHECK(capi_dlacpy(h, 1, ld, ld, A.data(), ld, R, ld));                                   // cholinv.hpp:13
    U bcDimLocal = (U)(CommInfo.c * CommInfo.d), bcMult = args.bc_mult_dim;                           // cholinv.hpp:15-18
    if (bcMult < 0) { bcMult = -bcMult; for (U i = 0; i < bcMult; ++i) bcDimLocal *= 2; } else { for (U i = 0; i < bcMult; ++i) bcDimLocal /= 2; }
    bcDimLocal = ld / std::min<U>(ld, std::max<U>(1, bcDimLocal));
    args.localDimension = args.trueLocalDimension = ld;
    args.globalDimension = args.trueGlobalDimension = A.num_rows_global();
    args.bcDimension = (U)CommInfo.d * bcDimLocal;
    args.num_base_cases = args.num_levels = 0;
    if (CommInfo.d == 1) {
      potrf_rec(args, R, ld, (U)0, ld);                 // (d == 1, c > 1: every layer holds the whole matrix and factors it)
    } else {
      // d x d x c grid (round 3): the arena holds, at the top level, the assembled R11 and R12 (4 h^2 each), the gathered pieces behind
      // them (4 h^2 each) and what the trailing update's SUMMA needs
      const U hh = ld - (ld >> args.split);
      args.work.reserve((int64_t)32 * hh * hh + 4 * (int64_t)args.bcDimension * args.bcDimension + 4096);
      potrf_rec_grid(args, CommInfo, R, ld, (U)0, ld);
    }
    if (packed) serialize<uppertri, uppertri>::invoke(args.Rfull, args.R, 0, ld, 0, ld, 0, ld, 0, ld);
    CAPITAL_CHECK(capi_get_info(h, &args.potrf_info));
    const int failed = capital::ranks_with_nonzero(CommInfo.world, args.potrf_info);       // (replicated base cases: all ranks see it; agreed on anyway)
    if (failed != 0)
      throw std::domain_error("cholinv::factor (TRSM mode): the matrix is not positive definite (non-positive pivot " + std::to_string(args.potrf_info) + " of a diagonal block)");
  }

  // TRSM mode on a d x d x c grid (d > 1).  The element-cyclic layout has block size ONE, so a triangular solve cannot be pipelined over
  // block rows the way ScaLAPACK does it; instead the solve is made LOCAL:
  //   2'. every rank assembles R11 (all-gather of the d^2 pieces over `slice`, re-indexed cyclic -> global: the base case's machinery at
  //       the order of a recursion level) and the whole A12; the P = d^2 c ranks of the grid each solve S2 / P global COLUMNS of
  //       R11^T X = A12 with the block TRSM (capi_dtrsm: no flop is done twice), the solved column ranges are all-gathered IN PLACE over
  //       the world communicator (they are contiguous column blocks of the global image), and every rank keeps its own cyclic piece;
  //   3.  A22 -= R12^T R12 by the triangular-output SUMMA of the reference schedule (partner exchange, row / column broadcasts, depth sum).
  // Base case: the aggregate is gathered and factored on every rank (ReplicateCommComp's pattern, policy.h:160-224, without the inverse).
  // Bytes per rank and level: ~7 h^2 for step 2' against 3 h^2 for the reference's TRMM SUMMA -- this mode trades communication for
  // arithmetic (n^3/3 executed, no inverse); on one node's mesh with multi-path transfers for the update it is the cheaper of the two
  // only when the links are not the limit.  Parity: R is unique, so the assembled factor must equal the one-GPU factor (tests).
  template <typename ArgType, typename CommType, typename U>
  static void potrf_rec_grid(ArgType& args, CommType&& t, double* R, U ld, U start, U dim) {
    using matmult::view;
    capi_handle_t h = capital::handle();
    matmult::arena& ws = args.work;
    const int64_t d = (int64_t)t.d, P = (int64_t)t.size, me = (int64_t)t.x + d * (int64_t)t.y;
    const U split1 = dim >> args.split;
    double* R11 = R + start + start * ld;
    if (((dim * (U)t.d) <= args.bcDimension) || (split1 < args.split)) {                              // cholinv.hpp:93
      CRITTER_START(CI::factor_diag);
      const int64_t L = dim, agg = L * d;
      const int64_t span = ((start + dim) != args.trueLocalDimension) ? agg : agg - (args.trueLocalDimension * d - args.trueGlobalDimension);
      const int64_t mark = ws.top;
      double* mine = ws.take(L * L);
      double* blocked = ws.take(L * L * d * d);
      double* cyc = ws.take(agg * agg);
      CAPITAL_CHECK(capi_dlacpy(h, 0, L, L, R11, ld, mine, L));
      CAPITAL_CHECK(capi_allgather(t.slice, mine, blocked, L * L));
      util::block_to_cyclic_rect(blocked, cyc, L, L, d);
      CAPITAL_CHECK(capi_dpotrf(h, CAPI_UPPER, span, cyc, agg));
      CAPITAL_CHECK(capi_dtrizero(h, CAPI_UPPER, agg, cyc, agg));
      util::cyclic_to_block_rect(blocked, cyc, L, L, d);
      CAPITAL_CHECK(capi_dlacpy(h, 0, L, L, blocked + me * L * L, L, R11, ld));
      ws.top = mark;
      ++args.num_base_cases;
      CRITTER_STOP(CI::factor_diag);
      return;
    }
    ++args.num_levels;
    const U split2 = dim - split1;
    double* R12 = R + start + (start + split1) * ld;
    view A22{R + (start + split1) + (start + split1) * ld, ld, split2, split2};
    potrf_rec_grid(args, t, R, ld, start, split1);
    CRITTER_START(CI::trsm);
    const int64_t mark = ws.top;
    view W{ws.take((int64_t)split1 * split2), split1, split1, split2};                                 // this rank's piece of R12, contiguous
    {
      const int64_t L1 = split1, L2 = split2, S1 = L1 * d, S2 = L2 * d;
      const int64_t ncw = (((S2 + P - 1) / P) + 1) & ~(int64_t)1, S2p = ncw * P;                       // columns per rank (even), padded width
      const int64_t inner = ws.top;
      double* Rf = ws.take(S1 * S1);
      double* Xf = ws.take(S1 * S2p);
      double* mine = ws.take(L1 * std::max(L1, L2));
      double* blocked = ws.take(L1 * std::max(L1, L2) * d * d);
      CAPITAL_CHECK(capi_dlacpy(h, 0, L1, L1, R11, ld, mine, L1));
      CAPITAL_CHECK(capi_allgather(t.slice, mine, blocked, L1 * L1));
      util::block_to_cyclic_rect(blocked, Rf, L1, L1, d);                                             // (zeroes the strictly lower part)
      CAPITAL_CHECK(capi_dlacpy(h, 0, L1, L2, R12, ld, mine, L1));
      CAPITAL_CHECK(capi_allgather(t.slice, mine, blocked, L1 * L2));
      if (S2p > S2) capital::dev_zero(Xf + S1 * S2, S1 * (S2p - S2));
      CAPITAL_CHECK(capi_block_to_cyclic_full(h, blocked, Xf, L1, L2, d));
      // this rank's columns: [rank ncw, (rank + 1) ncw) of the global block -- every layer holds the same image, so the P ranks share the work
      double* Xw = Xf + (int64_t)t.rank * ncw * S1;
      CAPITAL_CHECK(capi_dtrsm(h, CAPI_LEFT, CAPI_UPPER, CAPI_TRANS, CAPI_NONUNIT, S1, ncw, 1.0, Rf, S1, Xw, S1));
      CAPITAL_CHECK(capi_allgather(t.world, Xw, Xf, S1 * ncw));                                        // in place: send = recv + rank * count
      CAPITAL_CHECK(capi_cyclic_to_block(h, blocked, Xf, L1, L2, d));
      CAPITAL_CHECK(capi_dlacpy(h, 0, L1, L2, blocked + me * L1 * L2, L1, W.p, W.ld));
      CAPITAL_CHECK(capi_dlacpy(h, 0, L1, L2, W.p, W.ld, R12, ld));
      ws.top = inner;
    }
    CRITTER_STOP(CI::trsm);
    CRITTER_START(CI::tmu);
    {
      view Wx{ws.take(W.count()), split1, split1, split2};
      capital::dev_copy(Wx.p, W.p, W.count());
      util::transpose_raw(Wx.p, Wx.count(), ws.take(Wx.count()), t, matmult::summa::relay_space(t, Wx.count(), ws));
      matmult::summa::syrk(t, CAPI_UPPER, CAPI_TRANS, -1.0, W, Wx, 1.0, A22, ws);
    }
    ws.top = mark;
    CRITTER_STOP(CI::tmu);
    potrf_rec_grid(args, t, R, ld, start + split1, split2);
  }

  template <typename ArgType, typename U>
  static void potrf_rec(ArgType& args, double* R, U ld, U start, U dim) {
    capi_handle_t h = capital::handle();
    const U split1 = dim >> args.split;
    double* R11 = R + start + start * ld;
    if (dim <= args.bcDimension || split1 < args.split) {                                            // cholinv.hpp:93
      CRITTER_START(CI::factor_diag);
      CAPITAL_CHECK(capi_dpotrf(h, CAPI_UPPER, dim, R11, ld));
      CRITTER_STOP(CI::factor_diag);
      ++args.num_base_cases;
      return;
    }
    ++args.num_levels;
    const U split2 = dim - split1;
    double* R12 = R + start + (start + split1) * ld;
    double* R22 = R + (start + split1) + (start + split1) * ld;
    potrf_rec(args, R, ld, start, split1);
    CRITTER_START(CI::trsm);
    CAPITAL_CHECK(capi_dtrsm(h, CAPI_LEFT, CAPI_UPPER, CAPI_TRANS, CAPI_NONUNIT, split1, split2, 1.0, R11, ld, R12, ld));
    CRITTER_STOP(CI::trsm);
    CRITTER_START(CI::tmu);
    CAPITAL_CHECK(capi_dsyrk(h, CAPI_UPPER, CAPI_TRANS, split2, split1, -1.0, R12, ld, 1.0, R22, ld));
    CRITTER_STOP(CI::tmu);
    potrf_rec(args, R, ld, start + split1, split2);
  }

  // full local images of the factors (cholinv.hpp:30-46)
  template <typename ArgType, typename CommType>
  static matrix<typename ArgType::ScalarType, typename ArgType::DimensionType, rect> construct_R(ArgType& args, CommType&& CommInfo) {
    return construct(args.R, CommInfo);
  }
  template <typename ArgType, typename CommType>
  static matrix<typename ArgType::ScalarType, typename ArgType::DimensionType, rect> construct_Rinv(ArgType& args, CommType&& CommInfo) {
    if (args.solve_with_trsm) throw std::logic_error("cholinv: the TRSM mode forms no inverse (info::solve_with_trsm)");
    return construct(args.Rinv, CommInfo);
  }

private:
  // event slots of the top-level overlap (matmult::summa's pipe uses slots below 1000)
  static constexpr int EV_INPUT_HEAD = 1020, EV_INPUT_REST = 1021, EV_TOP_R12 = 1022, EV_EARLY_PACK = 1023, EV_INPUT_MID = 1019;
  // lookahead (single GPU): slot + depth; the bulk streams are stream indices LA_STREAM0 + depth
  static constexpr int EV_BC_POTRF = 1008, EV_BC_SCATTER = 1009;      // NoReplicationOverlap: scatter of R beside trtri
  static constexpr int EV_LA_LEAD = 1010, EV_LA_REST = 1014, LA_STREAM0 = 2, LA_MAX_DEPTH = 2;
  // smallest trailing block whose update is split; read per call so that tests can exercise the path at small orders.
  // OFF by default since round 4 (CAPITAL_LOOKAHEAD=1: from order 2048; CAPITAL_LOOKAHEAD_MIN=n: from order n; CAPITAL_NO_LOOKAHEAD: off whatever else is set).
  // Measured on one box with launches in resident rounds, alternating: n = 65536 1705.5 / 1706.0 ms with the lookahead against 1699.5 / 1700.4 without,
  // n = 32768 237.9 / 237.9 against 236.7 / 236.6 (profiles/r4_lookahead_ab.txt).  Why it cannot pay: the chain's kernels need what a resident tile
  // workgroup holds -- the diagonal-block kernel 132 KB of LDS, i.e. a CU with NO tile workgroup on it -- so beside a bulk launch the chain advances
  // only when a resident round drains (one kernel per 2-4 ms), while the bulk's rounds run at 0.86-0.88 of peak beside it instead of 0.91-0.92 alone.
  static int64_t lookahead_min() {
    if (getenv("CAPITAL_NO_LOOKAHEAD")) return -1;
    const char* e = getenv("CAPITAL_LOOKAHEAD_MIN");
    if (e) return (int64_t)atoll(e);
    return getenv("CAPITAL_LOOKAHEAD") ? (int64_t)2048 : -1;
  }

  // how many trailing updates may run beside the recursion at once (CAPITAL_LA_DEPTH, default and maximum LA_MAX_DEPTH = 2 bulk streams)
  static int lookahead_depth() {
    const char* e = getenv("CAPITAL_LA_DEPTH");
    return e ? std::max(0, std::min(atoi(e), (int)LA_MAX_DEPTH)) : (int)LA_MAX_DEPTH;
  }

  // columns [x0,x1), rows [y0,y1) of a full local image into the packed (uppertri) factor; shape = what is copied per column
  template <typename ArgType>
  static void pack_block(ArgType&, int shape, const double* full, double* packed_dst, int64_t ld, int64_t x0, int64_t x1, int64_t y0,
                         int64_t y1) {
    CAPITAL_CHECK(capi_serialize_shape(capital::handle(), shape, CAPI_RECT, CAPI_UPPERTRI, full, ld, ld, packed_dst, ld, ld,
                                       x0, x1, y0, y1, x0, x1, y0, y1));
  }

  template <typename M, typename CommType>
  static matrix<typename M::ScalarType, typename M::DimensionType, rect> construct(M& src, CommType&& CommInfo) {
    const auto ld = src.num_rows_local();
    matrix<typename M::ScalarType, typename M::DimensionType, rect> ret(src.num_columns_global(), src.num_rows_global(), CommInfo.d, CommInfo.d);
    serialize<typename SP::structure, rect>::invoke(src, ret, 0, ld, 0, ld, 0, ld, 0, ld);
    return ret;
  }

  template <typename ArgType, typename CommType, typename U>
  static void invoke(ArgType& args, CommType&& t, double* R, double* Ri, U ld, U start, U localDim, U globalDim, int wait_ev = -1) {
    using matmult::view;
    capi_handle_t h = capital::handle();
    const U split1 = localDim >> args.split;
    if (((localDim * (U)t.d) <= args.bcDimension) || (split1 < args.split)) {     // cholinv.hpp:93
      if (wait_ev >= 0) CAPITAL_CHECK(capi_event_wait(h, wait_ev));
      CRITTER_START(CI::factor_diag);
      base_case(args, t, R, Ri, ld, start, localDim);
      CRITTER_STOP(CI::factor_diag);
      return;
    }
    ++args.num_levels;
    const U split2 = localDim - split1;
    const bool single = (t.d == 1 && t.c == 1);
    matmult::arena& ws = args.work;
    // local blocks (column-major, column index first in the reference's (X,Y) convention)
    view R11i{Ri + start + start * ld, ld, split1, split1};
    view A12{R + start + (start + split1) * ld, ld, split1, split2};
    view A22{R + (start + split1) + (start + split1) * ld, ld, split2, split2};
    view R22i{Ri + (start + split1) + (start + split1) * ld, ld, split2, split2};
    view I12{Ri + start + (start + split1) * ld, ld, split1, split2};

    invoke(args, t, R, Ri, ld, start, split1, globalDim >> 1);                          // 1
    // everything of this block beyond its leading part was updated by the parent on a bulk stream: join it here
    if (wait_ev >= 0) CAPITAL_CHECK(capi_event_wait(h, wait_ev));
    if (start == 0 && args.input_lead > 0 && split1 == args.input_lead) CAPITAL_CHECK(capi_event_wait(h, EV_INPUT_MID));   // rest of the input's left half

    const bool top = (localDim == args.localDimension) && args.early_split == split1;
    if (top) CAPITAL_CHECK(capi_event_wait(h, EV_INPUT_REST));                          // the input's right part has landed
    CRITTER_START(CI::trsm);                                                            // 2
    int child_wait = -1;
    bool copy_deferred = false;
    const int64_t mark = ws.top;
    {
      view W{ws.take((int64_t)split1 * split2), split1, split1, split2};
      if (single) {
        CAPITAL_CHECK(capi_dtrmm_oop(h, CAPI_LEFT, CAPI_UPPER, CAPI_TRANS, CAPI_NONUNIT, split1, split2, 1.0, R11i.p, ld, A12.p, ld, W.p, W.ld));
      } else {
        // partner-exchanged copy of R11^-1 (cholinv.hpp:116-117)
        view Tx{ws.take((int64_t)split1 * split1), split1, split1, split1};
        // (on a grid that relays every rank joins the exchange with the same, packed, count: the diagonal ranks carry other pairs' units)
        if ((t.x == t.y && !t.multipath) || getenv("CAPITAL_NO_PACKED_COMM")) {
          CAPITAL_CHECK(capi_dlacpy(h, 0, split1, split1, R11i.p, ld, Tx.p, Tx.ld));
          util::transpose_raw(Tx.p, Tx.count(), ws.take(Tx.count()), t, matmult::summa::relay_space(t, Tx.count(), ws));
        } else if (t.x == t.y) {
          const int64_t np = (int64_t)split1 * (split1 + 1) / 2;
          CAPITAL_CHECK(capi_dlacpy(h, 0, split1, split1, R11i.p, ld, Tx.p, Tx.ld));
          util::transpose_raw(nullptr, np, nullptr, t, matmult::summa::relay_space(t, np, ws));
        } else {
          // the exchanged block is upper triangular: it crosses the link packed (n(n+1)/2) and is unpacked on arrival;
          // the strictly-lower half of Tx is never read (TRMM masks by selection)
          const int64_t np = (int64_t)split1 * (split1 + 1) / 2;
          double* pk = ws.take(np);
          CAPITAL_CHECK(capi_serialize_shape(h, CAPI_UPPERTRI, CAPI_RECT, CAPI_UPPERTRI, R11i.p, split1, ld, pk, split1, split1,
                                             0, split1, 0, split1, 0, split1, 0, split1));
          util::transpose_raw(pk, np, ws.take(np), t, matmult::summa::relay_space(t, np, ws));
          CAPITAL_CHECK(capi_serialize_shape(h, CAPI_UPPERTRI, CAPI_UPPERTRI, CAPI_RECT, pk, split1, split1, Tx.p, split1, split1,
                                             0, split1, 0, split1, 0, split1, 0, split1));
        }
        matmult::summa::trmm(t, CAPI_LEFT, CAPI_UPPER, CAPI_TRANS, CAPI_NONUNIT, 1.0, Tx, A12, W, ws);
      }
      const U lead = split2 >> args.split;
      const bool a22_splits = !(((split2 * (U)t.d) <= args.bcDimension) || (lead < args.split));
      const bool la = single && a22_splits && lead > 0 && lookahead_min() >= 0 && (int64_t)split2 >= lookahead_min() && args.la_depth < lookahead_depth();
      const bool pack_now = top && !std::is_same<typename SP::structure, rect>::value;
      // R12 into R (cholinv.hpp:122).  At the top level with lookahead and packed factors nothing reads it there before the
      // early packing: the copy then goes with the packing, behind the bulk of the update (W stays allocated until the join)
      const bool copy_later = la && pack_now && !getenv("CAPITAL_PACK_AT_ONCE");
      if (!copy_later) CAPITAL_CHECK(capi_dlacpy(h, 0, split1, split2, W.p, W.ld, A12.p, ld));
      copy_deferred = copy_later;
      CRITTER_STOP(CI::trsm);
      // R11, R12 and Rinv11 are final after step 2: they are packed on the second stream.  Without lookahead the packing
      // starts at once, beside the trailing update; with it, it waits for the bulk of the top-level update and then runs
      // beside the trailing block's latency-bound chain, where the GPU is otherwise idle (beside a tile kernel the packing
      // kernels' short workgroups keep whole CUs from the 512-thread-per-CU tile workgroups: the update lost ~10 %)
      auto early_pack = [&](int after_ev) {
        CAPITAL_CHECK(capi_stream_select(h, 1));
        CAPITAL_CHECK(capi_event_wait(h, after_ev));
        if (copy_later) CAPITAL_CHECK(capi_dlacpy(h, 0, split1, split2, W.p, W.ld, A12.p, ld));
        pack_block(args, CAPI_UPPERTRI, R, (double*)args.R.data(), ld, (U)0, split1, (U)0, split1);
        pack_block(args, CAPI_RECT, R, (double*)args.R.data(), ld, split1, ld, (U)0, split1);
        pack_block(args, CAPI_UPPERTRI, Ri, (double*)args.Rinv.data(), ld, (U)0, split1, (U)0, split1);
        CAPITAL_CHECK(capi_event_record(h, EV_EARLY_PACK));
        CAPITAL_CHECK(capi_stream_select(h, 0));
      };
      bool packed_early = false;

      CRITTER_START(CI::tmu);                                                           // 3
      // Lookahead: the trailing block's recursion starts with a long chain of latency-bound kernels on its leading
      // part, which needs only that part updated.  Update it first; the rest of the update runs on a low-priority
      // bulk stream beside that chain and is joined before the trailing block's own R12 product.
      if (la) {
        const int depth = args.la_depth;
        const U rest = split2 - lead;
        double* Wr = W.p + (int64_t)lead * W.ld;
        CAPITAL_CHECK(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, lead, split1, -1.0, W.p, W.ld, W.p, W.ld, 1.0, A22.p, ld));
        CAPITAL_CHECK(capi_event_record(h, EV_LA_LEAD + depth));
        CAPITAL_CHECK(capi_stream_select(h, LA_STREAM0 + depth));
        CAPITAL_CHECK(capi_event_wait(h, EV_LA_LEAD + depth));
        CAPITAL_CHECK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, lead, rest, split1, -1.0, W.p, W.ld, Wr, W.ld, 1.0, A22.p + (int64_t)lead * ld, ld));
        CAPITAL_CHECK(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, rest, split1, -1.0, Wr, W.ld, Wr, W.ld, 1.0,
                                  A22.p + lead + (int64_t)lead * ld, ld));
        CAPITAL_CHECK(capi_event_record(h, EV_LA_REST + depth));
        CAPITAL_CHECK(capi_stream_select(h, 0));
        child_wait = EV_LA_REST + depth;
        ++args.la_depth;                  // W stays allocated until the trailing block's recursion has joined
        if (pack_now && !getenv("CAPITAL_PACK_AT_ONCE")) { early_pack(EV_LA_REST + depth); packed_early = true; }
      } else if (single) {
        CAPITAL_CHECK(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, split2, split1, -1.0, W.p, W.ld, W.p, W.ld, 1.0, A22.p, ld));
      } else {
        view Wx{ws.take(W.count()), split1, split1, split2};
        capital::dev_copy(Wx.p, W.p, W.count());
        util::transpose_raw(Wx.p, Wx.count(), ws.take(Wx.count()), t, matmult::summa::relay_space(t, Wx.count(), ws));
        matmult::summa::syrk(t, CAPI_UPPER, CAPI_TRANS, -1.0, W, Wx, 1.0, A22, ws);
      }
      if (pack_now && !packed_early) {
        CAPITAL_CHECK(capi_event_record(h, EV_TOP_R12));
        early_pack(EV_TOP_R12);
      }
      if (child_wait < 0) ws.top = mark;
      CRITTER_STOP(CI::tmu);
    }

    invoke(args, t, R, Ri, ld, start + split1, split2, split2 * (U)t.d, child_wait);    // 4
    if (child_wait >= 0) { --args.la_depth; ws.top = mark; }

    CRITTER_START(CI::tmu);                                                             // 5
    if (!(!args.complete_inv && (globalDim == args.trueGlobalDimension))) {
      if (copy_deferred) CAPITAL_CHECK(capi_event_wait(h, EV_EARLY_PACK));              // R12 reached R on the second stream
      const int64_t mark = ws.top;
      view W2{ws.take((int64_t)split1 * split2), split1, split1, split2};
      matmult::summa::trmm(t, CAPI_LEFT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, 1.0, R11i, A12, W2, ws);
      if (single) {
        matmult::summa::trmm(t, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, -1.0, R22i, W2, I12, ws);
      } else {
        view W3{ws.take(W2.count()), split1, split1, split2};
        matmult::summa::trmm(t, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, -1.0, R22i, W2, W3, ws);
        CAPITAL_CHECK(capi_dlacpy(h, 0, split1, split2, W3.p, W3.ld, I12.p, ld));
      }
      ws.top = mark;
    }
    CRITTER_STOP(CI::tmu);
  }

  // cholinv.hpp:170-183 + policy.h:160-514
  template <typename ArgType, typename CommType, typename U>
  static void base_case(ArgType& args, CommType&& t, double* R, double* Ri, U ld, U start, U localDim) {
    capi_handle_t h = capital::handle();
    ++args.num_base_cases;
    double* Rb = R + start + start * ld;
    double* Ib = Ri + start + start * ld;
    if (t.d == 1) {
      // the whole diagonal block is local: factor and invert it in place (every layer holds the same block)
      CAPITAL_CHECK(capi_dpotrf_trtri(h, localDim, Rb, ld, Ib, ld));
      return;
    }
    const int64_t d = (int64_t)t.d, L = (int64_t)localDim, agg = L * d;
    // `span`: the aggregate minus the global padding when this is the last block (policy.h:196)
    const int64_t span = ((start + localDim) != args.trueLocalDimension) ? agg : agg - (args.trueLocalDimension * d - args.trueGlobalDimension);
    // With Serialize the pieces travel PACKED, L (L + 1) / 2 doubles instead of L^2 (policy.h:176: the messages are
    // base_case_table[...].num_elems() of an uppertri block) and are re-indexed by the triangle forms (util.hpp:57-102,167-201).
    constexpr bool packed_msg = !std::is_same<typename SP::structure, rect>::value;
    const int64_t piece = packed_msg ? L * (L + 1) / 2 : L * L;
    auto pack_piece = [&](const double* blk, double* dst) {          // local block (ld) -> message piece
      if (packed_msg) CAPITAL_CHECK(capi_serialize_shape(h, CAPI_UPPERTRI, CAPI_RECT, CAPI_UPPERTRI, blk, ld, ld, dst, L, L, 0, L, 0, L, 0, L, 0, L));
      else CAPITAL_CHECK(capi_dlacpy(h, 0, L, L, blk, ld, dst, L));
    };
    auto unpack_piece = [&](const double* src, double* blk) {        // message piece -> local block (ld); strictly-lower part stays zero
      if (packed_msg) CAPITAL_CHECK(capi_serialize_shape(h, CAPI_UPPERTRI, CAPI_UPPERTRI, CAPI_RECT, src, L, L, blk, ld, ld, 0, L, 0, L, 0, L, 0, L));
      else CAPITAL_CHECK(capi_dlacpy(h, 0, L, L, src, L, blk, ld));
    };
    auto to_cyclic = [&](const double* blocked, double* cyc) {
      if (packed_msg) CAPITAL_CHECK(capi_block_to_cyclic_tri(h, blocked, cyc, L, d));
      else util::block_to_cyclic_rect(blocked, cyc, L, L, d);
    };
    auto to_blocked = [&](double* blocked, const double* cyc) {
      if (packed_msg) CAPITAL_CHECK(capi_cyclic_to_block_tri(h, blocked, cyc, L, d));
      else util::cyclic_to_block_rect(blocked, cyc, L, L, d);
    };
    matmult::arena& ws = args.work;
    const int64_t mark = ws.top;
    double* mine = ws.take(piece);        // this rank's piece of A (in), then of R (out)
    double* minei = ws.take(piece);       // this rank's piece of R^-1 (out)
    const int64_t me = (int64_t)t.x + d * (int64_t)t.y;            // rank inside `slice` (topology.h:85,93-94)
    const bool worker = BP::every_layer || t.z == 0;
    if (BP::gather_all) {
      // ReplicateCommComp (policy.h:160-224) / ReplicateComp (:226-305): Allgather over the slice, every rank of the (or of
      // layer 0's) slice re-indexes, factors and inverts the aggregate and keeps its own piece
      double* blocked = ws.take(piece * d * d);
      double* cyc = ws.take(agg * agg);
      double* cyci = ws.take(agg * agg);
      if (worker) {
        pack_piece(Rb, mine);
        CAPITAL_CHECK(capi_allgather(t.slice, mine, blocked, piece));                      // C5, policy.h:176,240
        to_cyclic(blocked, cyc);
        capital::dev_zero(cyci, agg * agg);
        CAPITAL_CHECK(capi_dpotrf_trtri(h, span, cyc, agg, cyci, agg));                    // policy.h:199-201
      }
      if (!BP::every_layer) {                                                              // C7, policy.h:288-289: the aggregates travel
        CAPITAL_CHECK(capi_bcast(t.depth, cyc, agg * agg, 0));
        CAPITAL_CHECK(capi_bcast(t.depth, cyci, agg * agg, 0));
      }
      // util::cyclic_to_local (util.hpp:131-164): this rank's element-cyclic piece of both factors
      to_blocked(blocked, cyc);
      unpack_piece(blocked + me * piece, Rb);
      to_blocked(blocked, cyci);
      unpack_piece(blocked + me * piece, Ib);
    } else {
      // NoReplication (policy.h:307-414) / NoReplicationOverlap (:416-514): layer 0 gathers the pieces on the slice's rank 0,
      // which alone factors the aggregate and scatters the pieces of R and of R^-1; every rank then hands its two pieces down
      // its depth fibre.  Overlap: R's pieces are scattered (stream 1) WHILE the root inverts (compute stream) -- the
      // reference's MPI_Iscatter around trtri (:470-488).
      const bool root = worker && me == 0;
      double *blocked = nullptr, *cyc = nullptr, *cyci = nullptr;
      if (root) { blocked = ws.take(piece * d * d); cyc = ws.take(agg * agg); cyci = ws.take(agg * agg); }
      double* blockedi = (root && BP::overlap) ? ws.take(piece * d * d) : blocked;         // R's pieces are still in flight while R^-1's are cut
      if (worker) {
        pack_piece(Rb, mine);
        CAPITAL_CHECK(capi_gather(t.slice, mine, blocked, piece, 0));                      // C6, policy.h:322-332
        if (!BP::overlap) {
          if (root) {
            to_cyclic(blocked, cyc);
            capital::dev_zero(cyci, agg * agg);
            CAPITAL_CHECK(capi_dpotrf_trtri(h, span, cyc, agg, cyci, agg));                // :353,367 (the two scatters carry finished factors)
            to_blocked(blocked, cyc);
          }
          CAPITAL_CHECK(capi_scatter(t.slice, blocked, mine, piece, 0));                   // :361-365
          if (root) to_blocked(blocked, cyci);
          CAPITAL_CHECK(capi_scatter(t.slice, blocked, minei, piece, 0));                  // :373-377
        } else {
          if (root) {
            to_cyclic(blocked, cyc);
            CAPITAL_CHECK(capi_dpotrf(h, CAPI_UPPER, span, cyc, agg));                     // :462
            capital::dev_zero(cyci, agg * agg);
            CAPITAL_CHECK(capi_dlacpy(h, 1, span, span, cyc, agg, cyci, agg));             // :463 (memcpy to scratch; upper part is all trtri reads)
            to_blocked(blocked, cyc);
          }
          // every rank of the slice posts the scatter of R on stream 1, behind what the compute stream has queued so far
          CAPITAL_CHECK(capi_event_record(h, EV_BC_POTRF));
          CAPITAL_CHECK(capi_stream_select(h, 1));
          CAPITAL_CHECK(capi_event_wait(h, EV_BC_POTRF));
          CAPITAL_CHECK(capi_scatter(t.slice, blocked, mine, piece, 0));                   // :470-474 (MPI_Iscatter)
          CAPITAL_CHECK(capi_event_record(h, EV_BC_SCATTER));
          CAPITAL_CHECK(capi_stream_select(h, 0));
          if (root) {
            CAPITAL_CHECK(capi_dtrtri(h, CAPI_UPPER, CAPI_NONUNIT, span, cyci, agg));      // :476, beside the scatter
            CAPITAL_CHECK(capi_dtrizero(h, CAPI_UPPER, agg, cyci, agg));
            to_blocked(blockedi, cyci);
          }
          // one communicator, one stream at a time: the second scatter is issued only behind the first (MPI_Wait, :488)
          CAPITAL_CHECK(capi_event_wait(h, EV_BC_SCATTER));
          CAPITAL_CHECK(capi_scatter(t.slice, blockedi, minei, piece, 0));                 // :480-484
        }
      }
      CAPITAL_CHECK(capi_bcast(t.depth, mine, piece, 0));                                  // C7, policy.h:398-399,491,505: own pieces travel
      CAPITAL_CHECK(capi_bcast(t.depth, minei, piece, 0));
      unpack_piece(mine, Rb);
      unpack_piece(minei, Ib);
    }
    ws.top = mark;
  }
};

}  // namespace cholesky

#endif  // CAPITAL_CHOLESKY_CHOLINV_H_
