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
  // Both metrics read the factors where factor() left them (args.Q, args.R) and stream the tall panel in row chunks through
  // one bounded scratch block: the reference's validator materialises Q, R and the product Q R at full size (three more
  // m x n blocks), which at config 5's per-GPU slice (2^23 x 1024 = 64 GiB per block) would not fit beside A and Q's two buffers.
  static constexpr int64_t chunk_rows = int64_t(1) << 20;

  template <typename MatrixType, typename ArgType, typename RectCommType>
  static typename MatrixType::ScalarType residual(const MatrixType& A, ArgType& args, RectCommType&& RectTopo) {
    if (RectTopo.c != 1) throw std::logic_error("qr::validate: c == 1 only (see cacqr.h)");
    capi_handle_t h = capital::handle();
    auto R = AlgType::construct_R(args, RectTopo);
    const int64_t m = args.Q.num_rows_local(), n = args.Q.num_columns_local();
    CAPITAL_CHECK(capi_dtrizero(h, CAPI_UPPER, n, R.data(), n));                  // util::remove_triangle, validate.hpp:41
    const int64_t cm = std::min(m, chunk_rows);
    double* P = capital::dev_alloc(cm * n);
    double err2 = 0.0, ref2 = 0.0;
    for (int64_t r0 = 0; r0 < m; r0 += cm) {
      const int64_t rows = std::min(cm, m - r0);
      CAPITAL_CHECK(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, rows, n, n, 1.0, args.Q.data() + r0, m, R.data(), n, 0.0, P, rows));
      double sums[2];
      CAPITAL_CHECK(capi_diff_norms(h, 0, rows, n, P, rows, A.data() + r0, m, sums));   // synchronous: P is free again
      err2 += sums[0];
      ref2 += sums[1];
    }
    capital::dev_free(P);
    return std::sqrt(err2) / std::sqrt(ref2);
  }

  // Q^T Q of this rank's rows, summed over the ranks (validate.hpp:12-23), left on the device in `I` (n x n)
  template <typename ArgType, typename RectCommType>
  static void gram_of_Q(ArgType& args, RectCommType&& RectTopo, matrix<double, int64_t, rect>& I) {
    capi_handle_t h = capital::handle();
    const int64_t m = args.Q.num_rows_local(), n = args.Q.num_columns_local();
    CAPITAL_CHECK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, n, n, m, 1.0, args.Q.data(), m, args.Q.data(), m, 0.0, I.data(), n));
    CAPITAL_CHECK(capi_allreduce_sum(RectTopo.world, I.data(), n * n));           // validate.hpp:21-23 (column_alt spans the world at c == 1)
  }

  template <typename MatrixType, typename ArgType, typename RectCommType>
  static typename MatrixType::ScalarType orthogonality(const MatrixType& A, ArgType& args, RectCommType&& RectTopo) {
    (void)A;
    if (RectTopo.c != 1) throw std::logic_error("qr::validate: c == 1 only (see cacqr.h)");
    capi_handle_t h = capital::handle();
    const int64_t n = args.Q.num_columns_local();
    matrix<double, int64_t, rect> I(n, n, 1, 1), E(n, n, 1, 1);
    gram_of_Q(args, RectTopo, I);
    E.distribute_identity(0, 0, 1, 1, 1.0);
    double sums[2];
    CAPITAL_CHECK(capi_diff_norms(h, 0, n, n, I.data(), n, E.data(), n, sums));
    return std::sqrt(sums[0]) / std::sqrt((double)n * (double)n);                 // control = 1 per entry (validate.hpp:27-28)
  }
};

}  // namespace qr

#endif  // CAPITAL_TEST_QR_VALIDATE_H_
