// alg/qr/cacqr/cacqr.h -- communication-avoiding CholeskyQR / CholeskyQR2 on MI355X
// (reference src/alg/qr/cacqr/cacqr.h:13-78, cacqr.hpp:7-29,174-193,219-270).
//
// Same call surface: cacqr<SerializePolicy,IntermediatesPolicy>::factor(A, args, rectTopo), construct_Q / construct_R,
// info<T,U,CholeskyInversionType>(num_iter, ci_args).  The 1-D variant (c == 1, BASELINE configs 3 and 5) is the hot
// path; per sweep (cacqr.hpp:7-29):
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
  };

  template <typename MatrixType, typename ArgType, typename CommType>
  static void factor(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    static_assert(std::is_same<typename MatrixType::StructureType, rect>::value, "qr::cacqr requires matrices of rect structure");
    const auto gN = A.num_columns_global(), gM = A.num_rows_global();
    args.Q._register_(gN, gM, CommInfo.c, CommInfo.d);
    args.R._register_(gN, gN, CommInfo.c, CommInfo.c);
    if (CommInfo.c != 1)
      throw std::logic_error("qr::cacqr: the 3-D / tunable-grid sweeps (cacqr.hpp:75-170, c > 1) are the next row of the scope table and are not built yet; use c == 1");
    invoke_1d(A, args, CommInfo);
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
  template <typename ArgType, typename CommType>
  static void sweep_1d(const double* src, double* dst, int64_t m_loc, int64_t n, ArgType& args, CommType&& CommInfo) {
    capi_handle_t h = capital::handle();
    CRITTER_START(CQR::gram);
    CAPITAL_CHECK(capi_dsyrk(h, CAPI_UPPER, CAPI_TRANS, n, m_loc, 1.0, src, m_loc, 0.0, args.G.data(), n));           // K7
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
    sweep_1d(A.data(), args.Q.data(), m_loc, n, args, CommInfo);
    if (args.num_iter > 1) {
      args.R1._register_(n, n, 1, 1);
      capital::dev_copy(args.R1.data(), args.G.data(), n * n);                                                        // save_R_1d
      sweep_1d(args.Q.data(), args.Q.scratch(), m_loc, n, args, CommInfo);
      args.Q.swap();
      // R = R2 * R1 (cacqr.hpp:185-187): Ginv is free again and receives the product
      CAPITAL_CHECK(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n, n, 1.0, args.R1.data(), n, args.G.data(), n, args.Ginv.data(), n));
      finalize_R(args.Ginv, args, n);
    } else {
      finalize_R(args.G, args, n);   // the reference leaves the Gram matrix in R here with Serialize; R is what is documented
    }
  }

  template <typename ArgType>
  static void finalize_R(matrix<double, typename ArgType::DimensionType, rect>& src, ArgType& args, int64_t n) {
    serialize<uppertri, uppertri>::invoke(src, args.R, 0, n, 0, n, 0, n, 0, n);
  }
};

}  // namespace qr

#endif  // CAPITAL_QR_CACQR_H_
