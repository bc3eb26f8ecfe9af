// blas/engine.h -- the reference's blas::engine operator API (src/blas/engine.h:23-130,
// src/blas/interface.h:58-66) with the MKL shim (src/blas/interface.hpp:43-97) replaced by the MI355X C-ABI.
// Same enums, same ArgPack types, same static signatures; pointers are DEVICE pointers and the calls are
// asynchronous on the context's HIP stream.  Only T = double is specialised, as in the reference.
#ifndef CAPITAL_BLAS_ENGINE_H_
#define CAPITAL_BLAS_ENGINE_H_

#include "./../util/shared.h"

namespace blas {

enum class Order : unsigned char { AblasRowMajor = 0x0, AblasColumnMajor = 0x1 };
enum class Transpose : unsigned char { AblasNoTrans = 0x0, AblasTrans = 0x1 };
enum class Side : unsigned char { AblasLeft = 0x0, AblasRight = 0x1 };
enum class UpLo : unsigned char { AblasLower = 0x0, AblasUpper = 0x1 };
enum class Diag : unsigned char { AblasNonUnit = 0x0, AblasUnit = 0x1 };
enum class Method : unsigned char { AblasGemm = 0x0, AblasTrmm = 0x1, AblasSyrk = 0x10 };

template <typename T>
class ArgPack {
public:
  Method method;
};

template <typename T>
class ArgPack_gemm : public ArgPack<T> {
public:
  ArgPack_gemm(Order o, Transpose ta, Transpose tb, T a, T b) : order(o), transposeA(ta), transposeB(tb), alpha(a), beta(b) {
    this->method = Method::AblasGemm;
  }
  Order order;
  Transpose transposeA, transposeB;
  T alpha, beta;
};

template <typename T>
class ArgPack_trmm : public ArgPack<T> {
public:
  ArgPack_trmm(Order o, Side s, UpLo u, Transpose ta, Diag d, T a) : order(o), side(s), uplo(u), transposeA(ta), diag(d), alpha(a) {
    this->method = Method::AblasTrmm;
  }
  Order order;
  Side side;
  UpLo uplo;
  Transpose transposeA;
  Diag diag;
  T alpha;
};

template <typename T>
class ArgPack_syrk : public ArgPack<T> {
public:
  ArgPack_syrk(Order o, UpLo u, Transpose ta, T a, T b) : order(o), uplo(u), transposeA(ta), alpha(a), beta(b) {
    this->method = Method::AblasSyrk;
  }
  Order order;
  UpLo uplo;
  Transpose transposeA;
  T alpha, beta;
};

class engine {
public:
  engine() = delete;

  template <typename T>
  static void _gemm(T* A, T* B, T* C, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc, const ArgPack_gemm<T>& p);
  template <typename T>
  static void _trmm(T* A, T* B, int64_t m, int64_t n, int64_t lda, int64_t ldb, const ArgPack_trmm<T>& p);
  template <typename T>
  static void _syrk(T* A, T* C, int64_t n, int64_t k, int64_t lda, int64_t ldc, const ArgPack_syrk<T>& p);

private:
  static void need_colmajor(Order o) {
    // every hot-path caller passes AblasColumnMajor; a row-major request is the column-major call on
    // swapped operands, which the device layer does not need to duplicate
    if (o != Order::AblasColumnMajor) throw std::invalid_argument("blas::engine: only AblasColumnMajor is served by the device layer");
  }
};

template <>
inline void engine::_gemm(double* A, double* B, double* C, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc,
                          const ArgPack_gemm<double>& p) {
  need_colmajor(p.order);
  CAPITAL_CHECK(capi_dgemm(capital::handle(), (int)p.transposeA, (int)p.transposeB, m, n, k, p.alpha, A, lda, B, ldb, p.beta, C, ldc));
}
template <>
inline void engine::_trmm(double* A, double* B, int64_t m, int64_t n, int64_t lda, int64_t ldb, const ArgPack_trmm<double>& p) {
  need_colmajor(p.order);
  CAPITAL_CHECK(capi_dtrmm(capital::handle(), (int)p.side, (int)p.uplo, (int)p.transposeA, (int)p.diag, m, n, p.alpha, A, lda, B, ldb));
}
template <>
inline void engine::_syrk(double* A, double* C, int64_t n, int64_t k, int64_t lda, int64_t ldc, const ArgPack_syrk<double>& p) {
  need_colmajor(p.order);
  CAPITAL_CHECK(capi_dsyrk(capital::handle(), (int)p.uplo, (int)p.transposeA, n, k, p.alpha, A, lda, p.beta, C, ldc));
}

}  // namespace blas

#endif  // CAPITAL_BLAS_ENGINE_H_
