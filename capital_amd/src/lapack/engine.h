// lapack/engine.h -- the reference's lapack::engine operator API (src/lapack/engine.h:23-102,
// src/lapack/interface.h:49-59) on the MI355X C-ABI instead of LAPACKE (src/lapack/interface.hpp:30-88).
// n and lda stay 32-bit `int` as in the reference; A is a DEVICE pointer.  LAPACK's info, which the reference
// throws away, stays on the device and can be read with lapack::engine::info().
#ifndef CAPITAL_LAPACK_ENGINE_H_
#define CAPITAL_LAPACK_ENGINE_H_

#include "./../util/shared.h"

namespace lapack {

enum class Order : unsigned char { AlapackRowMajor = 0x0, AlapackColumnMajor = 0x1 };
enum class UpLo : unsigned char { AlapackLower = 0x0, AlapackUpper = 0x1 };
enum class Diag : unsigned char { AlapackNonUnit = 0x0, AlapackUnit = 0x1 };
enum class Method : unsigned char { AlapackPotrf = 0x0, AlapackTrtri = 0x1, AlapackGeqrf = 0x10, AlapackOrgqr = 0x11 };

class ArgPack {
public:
  Method method;
};
class ArgPack_potrf : public ArgPack {
public:
  ArgPack_potrf(Order o, UpLo u) : order(o), uplo(u) { method = Method::AlapackPotrf; }
  Order order;
  UpLo uplo;
};
class ArgPack_trtri : public ArgPack {
public:
  ArgPack_trtri(Order o, UpLo u, Diag d) : order(o), uplo(u), diag(d) { method = Method::AlapackTrtri; }
  Order order;
  UpLo uplo;
  Diag diag;
};
class ArgPack_geqrf : public ArgPack {
public:
  explicit ArgPack_geqrf(Order o) : order(o) { method = Method::AlapackGeqrf; }
  Order order;
};
class ArgPack_orgqr : public ArgPack {
public:
  explicit ArgPack_orgqr(Order o) : order(o) { method = Method::AlapackOrgqr; }
  Order order;
};

class engine {
public:
  engine() = delete;
  template <typename T>
  static void _potrf(T* A, int n, int lda, const ArgPack_potrf& p);
  template <typename T>
  static void _trtri(T* A, int n, int lda, const ArgPack_trtri& p);
  // declared by the reference (lapack/interface.h:55-59, interface.hpp:60-88) with no caller anywhere;
  // served by the Householder path of csrc/qr_f64.hip.  A and tau are DEVICE pointers.
  template <typename T>
  static void _geqrf(T* A, T* tau, int m, int n, int lda, const ArgPack_geqrf& p);
  template <typename T>
  static void _orgqr(T* A, T* tau, int m, int n, int k, int lda, const ArgPack_orgqr& p);
  // synchronises; 0 or the 1-based index of the first non-positive pivot since the last reset
  static int info() { int v = 0; CAPITAL_CHECK(capi_get_info(capital::handle(), &v)); return v; }
  static void reset_info() { CAPITAL_CHECK(capi_reset_info(capital::handle())); }
};

template <>
inline void engine::_potrf(double* A, int n, int lda, const ArgPack_potrf& p) {
  if (p.order != Order::AlapackColumnMajor) throw std::invalid_argument("lapack::engine: only AlapackColumnMajor is served");
  CAPITAL_CHECK(capi_dpotrf(capital::handle(), (int)p.uplo, n, A, lda));
}
template <>
inline void engine::_trtri(double* A, int n, int lda, const ArgPack_trtri& p) {
  if (p.order != Order::AlapackColumnMajor) throw std::invalid_argument("lapack::engine: only AlapackColumnMajor is served");
  CAPITAL_CHECK(capi_dtrtri(capital::handle(), (int)p.uplo, (int)p.diag, n, A, lda));
}

template <>
inline void engine::_geqrf(double* A, double* tau, int m, int n, int lda, const ArgPack_geqrf& p) {
  if (p.order != Order::AlapackColumnMajor) throw std::invalid_argument("lapack::engine: only AlapackColumnMajor is served");
  CAPITAL_CHECK(capi_dgeqrf(capital::handle(), m, n, A, lda, tau));
}
template <>
inline void engine::_orgqr(double* A, double* tau, int m, int n, int k, int lda, const ArgPack_orgqr& p) {
  if (p.order != Order::AlapackColumnMajor) throw std::invalid_argument("lapack::engine: only AlapackColumnMajor is served");
  CAPITAL_CHECK(capi_dorgqr(capital::handle(), m, n, k, A, lda, tau));
}

}  // namespace lapack

#endif  // CAPITAL_LAPACK_ENGINE_H_
