// test/qr/validate.h -- residual and orthogonality of CholeskyQR(2) (reference test/qr/validate.h, validate.hpp:4-52)
// for the 1-D grid (c == 1), where the reference's SquareTopo degenerates to one rank per cube:
//   residual       max over ranks is taken by the caller of  || Q_p R - A_p ||_F / || A_p ||_F      (validate.hpp:34-52)
//   orthogonality  || Q^T Q - I ||_F / n with Q^T Q summed over all ranks                          (validate.hpp:4-32)
#ifndef CAPITAL_TEST_QR_VALIDATE_H_
#define CAPITAL_TEST_QR_VALIDATE_H_

#include "../../src/alg/qr/cacqr/cacqr.h"

namespace qr {

template <typename AlgType>
class validate {
public:
  template <typename MatrixType, typename ArgType, typename RectCommType>
  static typename MatrixType::ScalarType residual(const MatrixType& A, ArgType& args, RectCommType&& RectTopo) {
    if (RectTopo.c != 1) throw std::logic_error("qr::validate: c == 1 only (see cacqr.h)");
    capi_handle_t h = capital::handle();
    auto R = AlgType::construct_R(args, RectTopo);
    auto Q = AlgType::construct_Q(args, RectTopo);
    const int64_t m = Q.num_rows_local(), n = Q.num_columns_local();
    CAPITAL_CHECK(capi_dtrizero(h, CAPI_UPPER, n, R.data(), n));                  // util::remove_triangle, validate.hpp:41
    MatrixType P(A.num_columns_global(), A.num_rows_global(), RectTopo.c, RectTopo.d);
    CAPITAL_CHECK(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, m, n, n, 1.0, Q.data(), m, R.data(), n, 0.0, P.data(), m));
    double sums[2];
    CAPITAL_CHECK(capi_diff_norms(h, 0, m, n, P.data(), m, A.data(), m, sums));
    return std::sqrt(sums[0]) / std::sqrt(sums[1]);
  }

  template <typename MatrixType, typename ArgType, typename RectCommType>
  static typename MatrixType::ScalarType orthogonality(const MatrixType& A, ArgType& args, RectCommType&& RectTopo) {
    (void)A;
    if (RectTopo.c != 1) throw std::logic_error("qr::validate: c == 1 only (see cacqr.h)");
    capi_handle_t h = capital::handle();
    auto Q = AlgType::construct_Q(args, RectTopo);
    const int64_t m = Q.num_rows_local(), n = Q.num_columns_local();
    matrix<double, int64_t, rect> I(n, n, 1, 1), E(n, n, 1, 1);
    CAPITAL_CHECK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, n, n, m, 1.0, Q.data(), m, Q.data(), m, 0.0, I.data(), n));
    CAPITAL_CHECK(capi_allreduce_sum(RectTopo.world, I.data(), n * n));           // validate.hpp:21-23 (column_alt spans the world at c == 1)
    E.distribute_identity(0, 0, 1, 1, 1.0);
    double sums[2];
    CAPITAL_CHECK(capi_diff_norms(h, 0, n, n, I.data(), n, E.data(), n, sums));
    return std::sqrt(sums[0]) / std::sqrt((double)n * (double)n);                 // control = 1 per entry (validate.hpp:27-28)
  }
};

}  // namespace qr

#endif  // CAPITAL_TEST_QR_VALIDATE_H_
