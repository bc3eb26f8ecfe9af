// test/cholesky/validate.h -- residual of the recursive Cholesky (reference test/cholesky/validate.h,
// validate.hpp:7-49): || triu(R^T R - A) ||_F / || triu(A) ||_F, R^T R formed by a SUMMA gemm on the grid.
#ifndef CAPITAL_TEST_CHOLESKY_VALIDATE_H_
#define CAPITAL_TEST_CHOLESKY_VALIDATE_H_

#include "../../src/alg/cholesky/cholinv/cholinv.h"

namespace cholesky {

template <typename AlgType>
class validate {
public:
  template <typename MatrixType, typename ArgType, typename CommType>
  static typename MatrixType::ScalarType residual(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    using T = typename MatrixType::ScalarType;
    if (args.dir != 'U') throw std::invalid_argument("validate: only dir == 'U' (cholinv.hpp:9)");
    if (CommInfo.d == 1) return residual_one_slice(A, args, CommInfo);
    auto R = AlgType::construct_R(args, CommInfo);
    util::remove_triangle(R, CommInfo.x, CommInfo.y, CommInfo.d, args.dir);      // validate.hpp:11
    MatrixType P(A.num_columns_global(), A.num_rows_global(), CommInfo.d, CommInfo.d);
    blas::ArgPack_gemm<T> gemmArgs(blas::Order::AblasColumnMajor, blas::Transpose::AblasTrans, blas::Transpose::AblasNoTrans, 1., 0.);
    if (CommInfo.d == 1) {
      // the partner exchange is the identity on a 1 x 1 slice: no second copy of a block that may be 32 GiB
      matmult::summa::invoke(R, R, P, CommInfo, gemmArgs);
    } else {
      auto RT = R;
      util::transpose(RT, CommInfo);                                             // validate.hpp:13
      matmult::summa::invoke(RT, R, P, CommInfo, gemmArgs);                       // validate.hpp:35
    }
    return util::residual_local(P, A, 0, CommInfo.slice, CommInfo.x, CommInfo.y, CommInfo.d, CommInfo.d);
  }

  // d == 1 (one GPU, or the 1 x 1 x c replication grid): every rank holds the whole matrix, and the reference's procedure --
  // a full copy of R, the product R^T R at full size and SUMMA's scratch (2 x three blocks) -- would need ~10 blocks of
  // n^2 doubles beside the factorisation's own seven: at n = 65536 (32 GiB per block) that does not fit 288 GB.  Same metric,
  // || triu(R^T R - A) ||_F / || triu(A) ||_F (validate.hpp:31-46), accumulated over column blocks of 4096 through one
  // n x 4096 scratch block; only the upper trapezoid of each block is formed (rows beyond the block's last column multiply zeros).
  template <typename MatrixType, typename ArgType, typename CommType>
  static typename MatrixType::ScalarType residual_one_slice(const MatrixType& A, ArgType& args, CommType&& CommInfo) {
    capi_handle_t h = capital::handle();
    auto R = AlgType::construct_R(args, CommInfo);
    const int64_t n = A.num_rows_local(), w = std::min<int64_t>(n, 4096);
    CAPITAL_CHECK(capi_dtrizero(h, CAPI_UPPER, n, R.data(), n));                                     // util::remove_triangle, validate.hpp:11
    double* P = capital::dev_alloc(n * w);
    double err2 = 0.0, ref2 = 0.0;
    for (int64_t j = 0; j < n; j += w) {
      const int64_t cols = std::min(w, n - j), rows = j + cols;
      // P(0:rows, 0:cols) = R(0:rows, 0:rows)^T R(0:rows, j:j+cols)
      CAPITAL_CHECK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, rows, cols, rows, 1.0, R.data(), n, R.data() + j * n, n, 0.0, P, rows));
      double sums[2];
      if (j > 0) {                                                                                   // rows above the diagonal block: all of them count
        CAPITAL_CHECK(capi_diff_norms(h, 0, j, cols, P, rows, A.data() + j * n, n, sums));
        err2 += sums[0]; ref2 += sums[1];
      }
      CAPITAL_CHECK(capi_diff_norms(h, 1, cols, cols, P + j, rows, A.data() + j + j * n, n, sums));   // the diagonal block's upper triangle
      err2 += sums[0]; ref2 += sums[1];
    }
    capital::dev_free(P);
    return std::sqrt(err2) / std::sqrt(ref2);
  }
};

}  // namespace cholesky

#endif  // CAPITAL_TEST_CHOLESKY_VALIDATE_H_
