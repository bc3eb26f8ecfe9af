// alg/qr/cacqr/cacqr.h -- communication-avoiding CholeskyQR / CholeskyQR2 on MI355X
// (reference src/alg/qr/cacqr/cacqr.h:13-78, cacqr.hpp:7-29,174-193,219-270).
//
// Same call surface: cacqr<SerializePolicy,IntermediatesPolicy>::factor(A, args, rectTopo), construct_Q / construct_R,
// info<T,U,CholeskyInversionType>(num_iter, ci_args).  c == 1 is the 1-D variant (BASELINE configs 3 and 5, the hot path);
// c == d is the 3-D variant on a cubic grid (sweep_3d below); 1 < c < d with c | d the tunable grid (d/c cubes, sweep_tune).  1-D, per sweep (cacqr.hpp:7-29):
//     K7  G = Q^T Q            capi_dsyrk, split-K over the tall dimension, upper triangle only
//     C8  G = sum over ranks   capi_allreduce_sum over `world` (packed n(n+1)/2 doubles with Serialize)
//     K8+K9  R = chol(G), R^-1 capi_dpotrf_trtri, replicated on every GPU
//     K5  Q <- Q R^-1          capi_dtrmm_oop (right, upper), out of place into the block's second buffer
// CholeskyQR2 runs the sweep twice and combines R = R2 R1 (cacqr.hpp:181-189, K6).
// Underneath, the A -> Q copy of factor() (cacqr.hpp:226) is folded into the first sweep (it reads A, writes Q) and the
// second sweep ping-pongs between Q's data and scratch buffers, so every sweep streams the panel exactly twice
// (read for the Gram, read+write for the solve) with no in-place hazard.
#ifndef CAPITAL_QR_CACQR_H_
#define CAPITAL_QR_CACQR_H_

#include "./../../alg.h"
#include "./../../matmult/summa/summa.h"
#include "./../../cholesky/cholinv/cholinv.h"
#include "./policy.h"

namespace qr {

template <class SerializePolicy = policy::cacqr::Serialize, class IntermediatesPolicy = policy::cacqr::SaveIntermediates>
class cacqr : public SerializePolicy, public IntermediatesPolicy {
public:
  using SP = SerializePolicy;
  using IP = IntermediatesPolicy;

  template <typename ScalarT, typename DimensionT, typename CholeskyInversionType>
  class info {
  public:
    using ScalarType = ScalarT;
    using DimensionType = DimensionT;
    using alg_type = cacqr<SerializePolicy, IntermediatesPolicy>;
    using cholesky_inverse_type = CholeskyInversionType;
    template <typename CholeskyInversionArgType>
    info(size_t num_iter, CholeskyInversionArgType&& ci_args) : num_iter(num_iter), cholesky_inverse_args(std::forward<CholeskyInversionArgType>(ci_args)) {}
    info(const info& p) : num_iter(p.num_iter), cholesky_inverse_args(p.cholesky_inverse_args) {}
    const size_t num_iter;                                                       // 1: CholeskyQR, 2: CholeskyQR2 (bench/qr/cacqr.cpp:14,40)
    typename CholeskyInversionType::template info<ScalarType, DimensionType> cholesky_inverse_args;
    matrix<ScalarType, DimensionType, rect> Q;
    matrix<ScalarType, DimensionType, typename SerializePolicy::structure> R;
    // n x n work blocks (Gram/R of the current sweep, its inverse, R1 of the first sweep, packed transfer image)
    matrix<ScalarType, DimensionType, rect> G, Ginv, R1;
    matrix<ScalarType, DimensionType, uppertri> Gpacked;
    // LAPACK info of the Gram matrix's factorisation (1-D variant): 0, or the first non-positive pivot (1-based) -- CholeskyQR's
    // known failure mode is a numerically rank-deficient A^T A (kappa(A) beyond ~1e8).  factor() throws std::domain_error then.
    int potrf_info = 0;
  };

  template <typename MatrixType, typename ArgType, typename CommType>
  static void factor(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    static_assert(std::is_same<typename MatrixType::StructureType, rect>::value, "qr::cacqr requires matrices of rect structure");
    const auto gN = A.num_columns_global(), gM = A.num_rows_global();
    args.Q._register_(gN, gM, CommInfo.c, CommInfo.d);
    args.R._register_(gN, gN, CommInfo.c, CommInfo.c);
    if (CommInfo.c == 1) {
      invoke_1d(A, args, CommInfo);
    } else if (CommInfo.d % CommInfo.c == 0) {
      // c == d: one cube (sweep_3d); c < d: d/c cubes side by side (sweep_tune, cacqr.hpp:124-170) -- the same sweep per cube
      // plus one all-reduce of the Gram block across the cubes
      invoke_3d(A, args, CommInfo);
    } else {
      throw std::logic_error("qr::cacqr: the c x d x c grid needs c to divide d");
    }
    if (!IP::keep_work) { args.G._destroy_(); args.Ginv._destroy_(); args.R1._destroy_(); args.Gpacked._destroy_(); }
  }

  template <typename ArgType, typename CommType>
  static matrix<typename ArgType::ScalarType, typename ArgType::DimensionType, rect> construct_Q(ArgType& args, CommType&& CommInfo) {
    const auto lm = args.Q.num_rows_local(), ln = args.Q.num_columns_local();
    matrix<typename ArgType::ScalarType, typename ArgType::DimensionType, rect> ret(args.Q.num_columns_global(), args.Q.num_rows_global(), CommInfo.c, CommInfo.d);
    serialize<rect, rect>::invoke(args.Q, ret, 0, ln, 0, lm, 0, ln, 0, lm);
    return ret;
  }
  template <typename ArgType, typename CommType>
  static matrix<typename ArgType::ScalarType, typename ArgType::DimensionType, rect> construct_R(ArgType& args, CommType&& CommInfo) {
    const auto ln = args.R.num_columns_local();
    matrix<typename ArgType::ScalarType, typename ArgType::DimensionType, rect> ret(args.R.num_columns_global(), args.R.num_rows_global(), CommInfo.c, CommInfo.c);
    serialize<uppertri, uppertri>::invoke(args.R, ret, 0, ln, 0, ln, 0, ln, 0, ln);
    return ret;
  }

protected:
  // one CholeskyQR sweep: dst <- src * chol(src^T src)^-1 ; leaves R in args.G and R^-1 in args.Ginv
  // src_tiled / dst_tiled: the panel is a "panel32" image (include/capital_hip.h) instead of column-major -- CholeskyQR2's intermediate Q1
  template <typename ArgType, typename CommType>
  static void sweep_1d(const double* src, double* dst, int64_t m_loc, int64_t n, ArgType& args, CommType&& CommInfo, bool src_tiled = false,
                       bool dst_tiled = false) {
    capi_handle_t h = capital::handle();
    CRITTER_START(CQR::gram);
    if (src_tiled) CAPITAL_CHECK(capi_dsyrk_panel32(h, n, m_loc, 1.0, src, 0.0, args.G.data(), n));                  // K7 on the image
    else CAPITAL_CHECK(capi_dsyrk(h, CAPI_UPPER, CAPI_TRANS, n, m_loc, 1.0, src, m_loc, 0.0, args.G.data(), n));      // K7
    if (CommInfo.size > 1) {                                                                                       // C8
      if (SP::packed_gram) {
        serialize<uppertri, uppertri>::invoke(args.G, args.Gpacked, 0, n, 0, n, 0, n, 0, n);
        CAPITAL_CHECK(capi_allreduce_sum(CommInfo.world, args.Gpacked.data(), args.Gpacked.num_elems()));
        serialize<uppertri, uppertri>::invoke(args.Gpacked, args.G, 0, n, 0, n, 0, n, 0, n);
      } else {
        CAPITAL_CHECK(capi_allreduce_sum(CommInfo.world, args.G.data(), n * n));
      }
    }
    CRITTER_STOP(CQR::gram);
    CRITTER_START(CQR::formR);
    CAPITAL_CHECK(capi_dpotrf_trtri(h, n, args.G.data(), n, args.Ginv.data(), n));                                    // K8 + K9
    if (src_tiled || dst_tiled)
      CAPITAL_CHECK(capi_dtrmm_right_panel32(h, m_loc, n, 1.0, args.Ginv.data(), n, src, src_tiled ? 0 : m_loc, dst, dst_tiled ? 0 : m_loc));       // K5
    else
      CAPITAL_CHECK(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, m_loc, n, 1.0, args.Ginv.data(), n, src, m_loc, dst, m_loc));  // K5
    CRITTER_STOP(CQR::formR);
  }

  template <typename MatrixType, typename ArgType, typename CommType>
  static void invoke_1d(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    capi_handle_t h = capital::handle();
    const int64_t n = A.num_columns_global(), m_loc = A.num_rows_local();
    args.G._register_(n, n, 1, 1);
    args.Ginv._register_(n, n, 1, 1);
    if (SP::packed_gram) args.Gpacked._register_(n, n, 1, 1);
    CAPITAL_CHECK(capi_reset_info(h));
    args.potrf_info = 0;
    // CholeskyQR2 at the full-width kernels' shape (n = 256, tall, whole 32-row tiles): Q1 = A R1^-1 is never seen by the caller; it is
    // written by sweep 1 and read twice by sweep 2 as a panel32 image -- one contiguous stream per pass instead of 256 column streams
    // (CAPITAL_NO_PANEL32: column-major throughout, A/B).  The arithmetic, and so Q and R, are the same bit for bit.
    const bool q1_tiled = args.num_iter > 1 && n == 256 && m_loc % 32 == 0 && m_loc >= 64 * n && !getenv("CAPITAL_NO_PANEL32") &&
                          !getenv("CAPI_NO_TS") && !getenv("CAPI_TS_ROWS16");
    sweep_1d(A.data(), args.Q.data(), m_loc, n, args, CommInfo, false, q1_tiled);
    if (args.num_iter > 1) {
      args.R1._register_(n, n, 1, 1);
      capital::dev_copy(args.R1.data(), args.G.data(), n * n);                                                        // save_R_1d
      sweep_1d(args.Q.data(), args.Q.scratch(), m_loc, n, args, CommInfo, q1_tiled, false);
      args.Q.swap();
      // R = R2 * R1 (cacqr.hpp:185-187): Ginv is free again and receives the product
      CAPITAL_CHECK(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n, n, 1.0, args.R1.data(), n, args.G.data(), n, args.Ginv.data(), n));
      finalize_R(args.Ginv, args, n);
    } else {
      finalize_R(args.G, args, n);   // the reference leaves the Gram matrix in R here with Serialize; R is what is documented
    }
    // one 4-byte read behind the launch chain (the reference drops LAPACK's info, lapack/interface.hpp:39,54)
    CAPITAL_CHECK(capi_get_info(h, &args.potrf_info));
    if (args.potrf_info != 0)
      throw std::domain_error("cacqr::factor: the Gram matrix is not positive definite (pivot " + std::to_string(args.potrf_info) +
                              "): A is numerically rank deficient for CholeskyQR; Q and R are not valid");
  }

  // ---- 3-D variant, c == d (cacqr.hpp:75-116 sweep_3d, :195-215 invoke_3d) ---------------------------------------------
  // Gram matrix by SUMMA-style exchange: the A block of the row-root (x == z) is broadcast along `row`, multiplied with
  // the local block, the partial Grams are reduced over `column` onto (y == z) and broadcast over `depth` from y:
  // every rank ends with the element-cyclic (x,y) block of G = A^T A, replicated over z -- the input format of cholinv.
  template <typename Sq, typename ArgType>
  static void sweep_3d(const double* src, double* dst, int64_t m_loc, int64_t n_loc, ArgType& args, Sq& sq, matmult::arena& ws,
                       matrix<double, int64_t, rect>& G, capi_comm_t across_cubes = nullptr) {
    using CI = typename std::remove_reference<ArgType>::type::cholesky_inverse_type;
    capi_handle_t h = capital::handle();
    const int64_t mark = ws.top;
    matmult::view Aloc{const_cast<double*>(src), m_loc, m_loc, n_loc};
    CRITTER_START(CQR::gram);
    matmult::view Abc = matmult::summa::panel(sq.row, sq.x == sq.z, (int)sq.z, Aloc, ws);                           // C9 Bcast(row)
    double* part = ws.take(n_loc * n_loc);
    CAPITAL_CHECK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, n_loc, n_loc, m_loc, 1.0, Abc.p, Abc.ld, src, m_loc, 0.0, part, n_loc));   // K3
    double* red = ws.take(n_loc * n_loc);
    CAPITAL_CHECK(capi_reduce_sum(sq.column, part, red, n_loc * n_loc, (int)sq.z));                                   // C9 Reduce(column)
    double* gsrc = (sq.y == sq.z) ? red : part;    // the depth root (z == y) holds the reduced block; others receive into `part`
    // tunable grid (cacqr.hpp:147): the cubes' blocks are summed over column_alt (every rank takes part; only the depth
    // roots' sums are used)
    if (across_cubes) CAPITAL_CHECK(capi_allreduce_sum(across_cubes, gsrc, n_loc * n_loc));
    CAPITAL_CHECK(capi_bcast(sq.depth, gsrc, n_loc * n_loc, (int)sq.y));                                              // C9 Bcast(depth)
    capital::dev_copy(G.data(), gsrc, n_loc * n_loc);
    CRITTER_STOP(CQR::gram);
    CRITTER_START(CQR::formR);
    CI::factor(G, args.cholesky_inverse_args, sq);                                                                   // cacqr.hpp:103
    auto Rinv = CI::construct_Rinv(args.cholesky_inverse_args, sq);
    if (args.cholesky_inverse_args.complete_inv) {
      matmult::view Tv{Rinv.data(), n_loc, n_loc, n_loc}, Cv{dst, m_loc, m_loc, n_loc};
      matmult::summa::trmm(sq, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, 1.0, Tv, Aloc, Cv, ws);             // cacqr.hpp:108-112
      capital::sync();   // Rinv (a temporary) must outlive the multiply
    } else {
      // solve (cacqr.hpp:44-73): the top-level R^-1_12 was not formed, so Q = A R^-1 goes block by block over the same
      // local split cholinv used:  Q1 = A1 R11^-1,  Q2 = (A2 - Q1 R12) R22^-1.
      // (The reference passes alpha = 1, beta = -1 to the middle product, cacqr.hpp:58, which yields Q1 R12 - A2 and a
      //  sign-flipped Q2 with Q R != A; the path is outside its validated combinations, SURVEY 8c.  Built to the algebra.)
      auto Rfull = CI::construct_R(args.cholesky_inverse_args, sq);
      const int64_t s1 = n_loc >> args.cholesky_inverse_args.split, s2 = n_loc - s1;
      matmult::view X11{Rinv.data(), n_loc, s1, s1}, X22{Rinv.data() + s1 + s1 * n_loc, n_loc, s2, s2};
      matmult::view R12{Rfull.data() + s1 * n_loc, n_loc, s1, s2};
      matmult::view A1{const_cast<double*>(src), m_loc, m_loc, s1}, A2{const_cast<double*>(src) + s1 * m_loc, m_loc, m_loc, s2};
      matmult::view Q1{dst, m_loc, m_loc, s1}, Q2{dst + s1 * m_loc, m_loc, m_loc, s2};
      matmult::summa::trmm(sq, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, 1.0, X11, A1, Q1, ws);
      matmult::view Tmp{ws.take(m_loc * s2), m_loc, m_loc, s2};
      capital::dev_copy(Tmp.p, A2.p, m_loc * s2);
      matmult::summa::gemm(sq, CAPI_NOTRANS, CAPI_NOTRANS, -1.0, Q1, R12, 1.0, Tmp, ws);
      matmult::summa::trmm(sq, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, 1.0, X22, Tmp, Q2, ws);
      capital::sync();   // Rinv / Rfull (temporaries) must outlive the multiplies
    }
    CRITTER_STOP(CQR::formR);
    ws.top = mark;
  }

  template <typename MatrixType, typename ArgType, typename CommType>
  static void invoke_3d(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    using CI = typename std::remove_reference<ArgType>::type::cholesky_inverse_type;
    topo::square sq(CommInfo.cube, CommInfo.c, CommInfo.layout, CommInfo.num_chunks);
    const int64_t n = A.num_columns_global(), n_loc = A.num_columns_local(), m_loc = A.num_rows_local();
    matrix<double, int64_t, rect> G(n, n, CommInfo.c, CommInfo.c);
    matmult::arena& ws = matmult::summa::scratch_arena();
    ws.reserve(8 * m_loc * n_loc + 8 * n_loc * n_loc + 1024);
    capi_comm_t across = CommInfo.c < CommInfo.d ? CommInfo.column_alt : nullptr;
    sweep_3d(A.data(), args.Q.data(), m_loc, n_loc, args, sq, ws, G, across);
    matrix<double, int64_t, rect> Rfinal = CI::construct_R(args.cholesky_inverse_args, sq);
    if (args.num_iter > 1) {
      matrix<double, int64_t, rect> R1 = Rfinal;                                                                     // save_R_3d
      sweep_3d(args.Q.data(), args.Q.scratch(), m_loc, n_loc, args, sq, ws, G, across);
      args.Q.swap();
      matrix<double, int64_t, rect> R2 = CI::construct_R(args.cholesky_inverse_args, sq);
      // R = R2 * R1 on the grid (cacqr.hpp:208-211): right multiply by the triangular R1
      matmult::view Tv{R1.data(), n_loc, n_loc, n_loc}, Bv{R2.data(), n_loc, n_loc, n_loc}, Cv{Rfinal.data(), n_loc, n_loc, n_loc};
      matmult::summa::trmm(sq, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, 1.0, Tv, Bv, Cv, ws);
      capital::sync();
    }
    serialize<uppertri, uppertri>::invoke(Rfinal, args.R, 0, n_loc, 0, n_loc, 0, n_loc, 0, n_loc);                    // cacqr.hpp:214
    capital::sync();
  }

  template <typename ArgType>
  static void finalize_R(matrix<double, typename ArgType::DimensionType, rect>& src, ArgType& args, int64_t n) {
    serialize<uppertri, uppertri>::invoke(src, args.R, 0, n, 0, n, 0, n, 0, n);
  }
};

}  // namespace qr

#endif  // CAPITAL_QR_CACQR_H_
