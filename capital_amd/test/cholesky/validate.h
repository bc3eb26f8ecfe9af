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
};

}  // namespace cholesky

#endif  // CAPITAL_TEST_CHOLESKY_VALIDATE_H_
