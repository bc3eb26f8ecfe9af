// alg/matmult/summa/summa.h -- 3-D SUMMA on the GPUs of one node (reference src/alg/matmult/summa/summa.h:24-34,
// summa.hpp:6-254).
//
// Same four invoke() overloads keyed on the blas ArgPack.  Per multiply, rank (x,y,z) of a d x d x c grid receives the
// A panel of the row-root and the B panel of the column-root of K-class q, multiplies locally on the MFMA tile kernel
// and sums the partial products over `depth`:
//     C1 MPI_Bcast(row)  -> capi_bcast(row)      C2 MPI_Bcast(column) -> capi_bcast(column)
//     K1/K4 cblas_dgemm / cblas_dtrmm -> capi_dgemm / capi_dgemmt / capi_dtrmm_acc
//     C3 MPI_Allreduce(depth) -> capi_allreduce_sum(depth)      M3 beta-axpy -> capi_dgeadd
// Differences from the reference, all behind the same results:
//   * operands are "views" (pointer, leading dimension) into device-resident blocks, so at d == c == 1 the multiply
//     runs in place with no packing, zero-fill or axpy passes at all;
//   * the trailing update only computes the triangle that is used (capi_dgemmt) instead of a full GEMM
//     (summa.hpp:115-146 runs cblas_dgemm and repacks);
//   * any c dividing d is accepted: layer z owns K-classes q = z, z+c, ... (the reference needs c == d);
//     d == 1 with c > 1 slices the local K range instead (pure replication, used for the 2-GPU grid).
#ifndef CAPITAL_MATMULT_SUMMA_H_
#define CAPITAL_MATMULT_SUMMA_H_

#include "./../../alg.h"

namespace matmult {

// contiguous device scratch handed out stack-wise; sized once per factorisation (the reference's simulate() tables)
struct arena {
  double* base = nullptr;
  int64_t cap = 0, top = 0;
  arena() = default;
  arena(const arena&) = delete;                 // owns device memory: a copy would free it twice, an assignment would leak it
  arena& operator=(const arena&) = delete;
  arena(arena&& o) noexcept : base(o.base), cap(o.cap), top(o.top) { o.base = nullptr; o.cap = o.top = 0; }
  arena& operator=(arena&& o) noexcept {
    if (this != &o) { release(); base = o.base; cap = o.cap; top = o.top; o.base = nullptr; o.cap = o.top = 0; }
    return *this;
  }
  ~arena() { release(); }
  void release() { capital::dev_free(base); base = nullptr; cap = top = 0; }
  void reserve(int64_t count) {
    if (count > cap) {
      capital::sync();
      capital::dev_free(base);
      base = capital::dev_alloc(count);
      cap = count;
    }
    top = 0;
  }
  double* take(int64_t count) {
    count = (count + 1) & ~(int64_t)1;  // keep 16-byte alignment
    if (top + count > cap) throw std::runtime_error("matmult::arena exhausted (sizing rule out of date)");
    double* p = base + top;
    top += count;
    return p;
  }
};

struct view {
  double* p;
  int64_t ld, rows, cols;
  bool contiguous() const { return ld == rows; }
  int64_t count() const { return rows * cols; }
};

class summa {
public:
  // ---- public overloads on matrix<> objects (summa.h:24-34) ------------------------------------------------------
  template <typename MatrixAType, typename MatrixBType, typename MatrixCType, typename CommType>
  static void invoke(MatrixAType& A, MatrixBType& B, MatrixCType& C, CommType&& CommInfo, blas::ArgPack_gemm<typename MatrixAType::ScalarType>& pack) {
    static_assert(std::is_same<typename MatrixAType::StructureType, rect>::value && std::is_same<typename MatrixBType::StructureType, rect>::value &&
                      std::is_same<typename MatrixCType::StructureType, rect>::value,
                  "summa gemm works on rect blocks (the reference's callers only pass rect: cacqr.hpp:62, validate.hpp)");
    arena& ws = scratch_arena();
    if (!(CommInfo.d == 1 && CommInfo.c == 1)) ws.reserve(2 * (A.num_elems() + B.num_elems() + C.num_elems()) + 64);
    gemm(CommInfo, (int)pack.transposeA, (int)pack.transposeB, pack.alpha, as_view(A), as_view(B), pack.beta, as_view(C), ws);
  }
  // B <- alpha * op(A) * B  or  alpha * B * op(A), A triangular (summa.hpp:46-83)
  template <typename MatrixAType, typename MatrixBType, typename CommType>
  static void invoke(MatrixAType& A, MatrixBType& B, CommType&& CommInfo, blas::ArgPack_trmm<typename MatrixAType::ScalarType>& pack) {
    static_assert(std::is_same<typename MatrixAType::StructureType, rect>::value && std::is_same<typename MatrixBType::StructureType, rect>::value,
                  "packed operands are unpacked by the algorithm layer before they reach the device summa");
    arena& ws = scratch_arena();
    ws.reserve(3 * A.num_elems() + 4 * B.num_elems() + 64);
    view out{ws.take(B.num_elems()), B.num_rows_local(), B.num_rows_local(), B.num_columns_local()};
    trmm(CommInfo, (int)pack.side, (int)pack.uplo, (int)pack.transposeA, (int)pack.diag, pack.alpha, as_view(A), as_view(B), out, ws);
    capital::dev_copy(B.data(), out.p, B.num_elems());
  }
  // C <- alpha * A^T A + beta * C (or A A^T), summa.hpp:85-96: the second operand is the partner-exchanged copy of A
  template <typename MatrixSrcType, typename MatrixDestType, typename CommType>
  static void invoke(MatrixSrcType& A, MatrixDestType& C, CommType&& CommInfo, blas::ArgPack_syrk<typename MatrixSrcType::ScalarType>& pack) {
    MatrixSrcType B = A;
    invoke(A, B, C, CommInfo, pack);
  }
  // summa.hpp:98-109: B is exchanged with the transpose partner, then the triangular-output multiply
  template <typename MatrixSrcType, typename MatrixDestType, typename CommType>
  static void invoke(MatrixSrcType& A, MatrixSrcType& B, MatrixDestType& C, CommType&& CommInfo, blas::ArgPack_syrk<typename MatrixSrcType::ScalarType>& pack) {
    static_assert(std::is_same<typename MatrixSrcType::StructureType, rect>::value && std::is_same<typename MatrixDestType::StructureType, rect>::value,
                  "packed operands are unpacked by the algorithm layer before they reach the device summa");
    util::transpose(B, CommInfo);
    arena& ws = scratch_arena();
    ws.reserve(2 * (A.num_elems() + B.num_elems() + C.num_elems()) + 64);
    syrk(CommInfo, (int)pack.uplo, (int)pack.transposeA, pack.alpha, as_view(A), as_view(B), pack.beta, as_view(C), ws);
  }

  // ---- view-level engine (used directly by cholinv / cacqr) ---------------------------------------------------------
  // C <- alpha*op(A)*op(B) + beta*C.  Operand blocks are the LOCAL blocks of this rank; roots are chosen per K-class.
  // With num_chunks > 0 (one K-class per layer, op(B) = B): A first, then B and the product column chunk by column chunk --
  // bcast of chunk j+1 || MFMA on chunk j || depth all-reduce of chunk j-1 (summa.hpp:195-215,238-249).
  template <typename CommType>
  static void gemm(CommType&& t, int transA, int transB, double alpha, view A, view B, double beta, view C, arena& ws) {
    capi_handle_t h = capital::handle();
    const int64_t M = C.rows, N = C.cols, K = transA ? A.rows : A.cols;
    if (t.d == 1 && t.c == 1) {
      CAPITAL_CHECK(capi_dgemm(h, transA, transB, M, N, K, alpha, A.p, A.ld, B.p, B.ld, beta, C.p, C.ld));
      return;
    }
    if (t.d == 1 && colsplit() && transB == CAPI_NOTRANS) {
      // replicated layers: layer z forms C[:, b_z : b_z+1) = alpha op(A) B[:, b_z : b_z+1) + beta C[:, ...) in place, then the blocks travel
      colsplit_run(t, N, false, C.p, C.ld,
                   [&](int64_t c0, int64_t c1) {
                     CAPITAL_CHECK(capi_dgemm(h, transA, CAPI_NOTRANS, M, c1 - c0, K, alpha, A.p, A.ld, B.p + c0 * B.ld, B.ld, beta, C.p + c0 * C.ld, C.ld));
                   },
                   [&](int64_t) { return M; }, ws);
      return;
    }
    const int64_t mark = ws.top;
    view acc = t.c > 1 ? view{ws.take(M * N), M, M, N} : C;
    const size_t steps = t.d > 1 ? t.d / t.c : 1;
    const bool piped = t.num_chunks > 0 && transB == CAPI_NOTRANS && (t.d > 1 || t.c > 1);
    const int nch = piped ? chunk_count(t.num_chunks, N) : 1;
    pipe P(piped && nch > 1);
    enum { E0 = 0, EA = 1, EB = 2, EC = 70, ER = 140 };
    if (!P.on) {
      if (t.d == 1) {
        // pure replication: layer z multiplies its slice of the local K range
        int64_t k0, k1;
        kslice(K, t.c, t.z, k0, k1);
        const double* a = transA ? A.p + k0 : A.p + k0 * A.ld;
        const double* b = transB ? B.p + k0 * B.ld : B.p + k0;
        CAPITAL_CHECK(capi_dgemm(h, transA, transB, M, N, k1 - k0, alpha, a, A.ld, b, B.ld, 0.0, acc.p, acc.ld));
      } else {
        for (size_t s = 0; s < steps; ++s) {
          view a = panel(t, AX_ROW, s, A, ws), b = panel(t, AX_COLUMN, s, B, ws);
          const double bt = (t.c > 1) ? (s ? 1.0 : 0.0) : (s ? 1.0 : beta);
          CAPITAL_CHECK(capi_dgemm(h, transA, transB, M, N, K, alpha, a.p, a.ld, b.p, b.ld, bt, acc.p, acc.ld));
        }
      }
      if (t.c > 1) {
        allreduce_depth(t, acc.p, acc.count(), ws);
        CAPITAL_CHECK(capi_dgeadd(h, 0, M, N, 1.0, acc.p, acc.ld, beta, C.p, C.ld));
      }
      ws.top = mark;
      return;
    }
    // ---- pipelined over the output's column chunks; every K-class step s of this layer contributes to every chunk
    const bool sliced = (t.d == 1);
    int64_t k0 = 0, k1 = K;
    std::vector<view> a(steps, A), b(steps, B);
    P.main(); P.rec(E0);
    P.comm(); P.wait(E0);
    if (sliced) {
      kslice(K, t.c, t.z, k0, k1);
    } else {
      for (size_t s = 0; s < steps; ++s) {
        a[s] = panel(t, AX_ROW, s, A, ws);
        const bool rootB = t.y == t.z + s * t.c;
        b[s] = view{(rootB && B.contiguous()) ? B.p : ws.take(B.count()), B.rows, B.rows, B.cols};
      }
    }
    P.rec(EA);
    int64_t cmax = 0;
    for (int j = 0; j < nch; ++j) { int64_t c0, c1; chunk_range(N, nch, j, c0, c1); cmax = std::max(cmax, c1 - c0); }
    double* relay = relay_space(t, std::max(B.rows, acc.rows) * cmax, ws);
    for (int j = 0; j < nch; ++j) {
      int64_t c0, c1;
      chunk_range(N, nch, j, c0, c1);
      if (!sliced && c1 > c0)
        for (size_t s = 0; s < steps; ++s) {
          const bool rootB = t.y == t.z + s * t.c;
          if (rootB && !B.contiguous()) CAPITAL_CHECK(capi_dlacpy(h, 0, B.rows, c1 - c0, B.p + c0 * B.ld, B.ld, b[s].p + c0 * b[s].ld, b[s].ld));
          bcast_axis(t, AX_COLUMN, s, b[s].p + c0 * b[s].ld, b[s].rows * (c1 - c0), relay);
        }
      P.rec(EB + j);
    }
    P.main(); P.wait(EA);
    for (int j = 0; j < nch; ++j) {
      int64_t c0, c1;
      chunk_range(N, nch, j, c0, c1);
      P.wait(EB + j);
      if (c1 > c0)
        for (size_t s = 0; s < steps; ++s) {
          const double bt = s ? 1.0 : (t.c > 1 ? 0.0 : beta);
          const double* ap = sliced ? (transA ? a[s].p + k0 : a[s].p + k0 * a[s].ld) : a[s].p;
          CAPITAL_CHECK(capi_dgemm(h, transA, CAPI_NOTRANS, M, c1 - c0, k1 - k0, alpha, ap, a[s].ld, b[s].p + k0 + c0 * b[s].ld, b[s].ld, bt,
                                   acc.p + c0 * acc.ld, acc.ld));
        }
      P.rec(EC + j);
    }
    if (t.c > 1) {
      P.comm();
      double* half = depth_space(t, acc.rows * cmax, ws);
      for (int j = 0; j < nch; ++j) {
        int64_t c0, c1;
        chunk_range(N, nch, j, c0, c1);
        P.wait(EC + j);
        if (c1 > c0) allreduce_depth(t, acc.p + c0 * acc.ld, acc.rows * (c1 - c0), half, relay);
      }
      P.rec(ER);
      P.main(); P.wait(ER);
      CAPITAL_CHECK(capi_dgeadd(h, 0, M, N, 1.0, acc.p, acc.ld, beta, C.p, C.ld));
    } else {
      P.main();
    }
    ws.top = mark;
  }

  // Cross-stream pipeline helper: with num_chunks > 0 the collectives of a multiply run on the handle's communication
  // stream beside the tile kernel on the compute stream (the reference's MPI_Ibcast/MPI_Iallreduce chunking,
  // summa.hpp:195-215,238-249).  Host calls are issued in an order that is also valid when executed sequentially.
  struct pipe {
    capi_handle_t h;
    bool on;
    int base;
    explicit pipe(bool enable) : h(capital::handle()), on(enable) {
      static CAPITAL_RANK_LOCAL int next = 0;
      base = next;
      next = (next + 256) % 1024;
    }
    void comm() { if (on) CAPITAL_CHECK(capi_stream_select(h, 1)); }
    void main() { if (on) CAPITAL_CHECK(capi_stream_select(h, 0)); }
    void rec(int s) { if (on) CAPITAL_CHECK(capi_event_record(h, base + s)); }
    void wait(int s) { if (on) CAPITAL_CHECK(capi_event_wait(h, base + s)); }
  };
  static int chunk_count(size_t num_chunks, int64_t cols) {
    // (a chunk costs two cross-stream hand-offs -- event record on one HIP stream, wait on the other: tens of microseconds between two
    //  hardware queues, measured in round 3 --, so it has to carry work in the 100-microsecond range: 512 columns; tests shrink it)
    static const int64_t min_cols = getenv("CAPITAL_MIN_CHUNK_COLS") ? std::max(2, atoi(getenv("CAPITAL_MIN_CHUNK_COLS"))) : 512;
    int n = (int)std::min<int64_t>((int64_t)std::min<size_t>(num_chunks, 64), std::max<int64_t>(cols / min_cols, 1));
    return n < 1 ? 1 : n;
  }
  // column range of chunk j (multiples of 2 columns keep 16-byte alignment)
  static void chunk_range(int64_t cols, int nch, int j, int64_t& c0, int64_t& c1) {
    const int64_t per = ((cols + nch - 1) / nch + 1) & ~(int64_t)1;
    c0 = std::min<int64_t>(cols, per * j);
    c1 = std::min<int64_t>(cols, c0 + per);
  }

  // Cout <- alpha*op(T)*B (Left) or alpha*B*op(T) (Right); T is this rank's block of a triangular matrix.
  // Left: T travels along `row`, B along `column`; Right: B along `row`, T along `column` (summa.hpp:59-71).
  // Cout must be contiguous (ld == rows) when the grid has more than one rank.
  template <typename CommType>
  static void trmm(CommType&& t, int side, int uplo, int trans, int diag, double alpha, view T, view B, view Cout, arena& ws) {
    capi_handle_t h = capital::handle();
    const int64_t M = B.rows, N = B.cols;
    if (t.d == 1 && t.c == 1) {
      CAPITAL_CHECK(capi_dtrmm_oop(h, side, uplo, trans, diag, M, N, alpha, T.p, T.ld, B.p, B.ld, Cout.p, Cout.ld));
      return;
    }
    const bool eff_upper = (uplo == CAPI_UPPER) != (trans == CAPI_TRANS);
    if (t.d == 1 && colsplit() && (side == CAPI_LEFT || (eff_upper && trans == CAPI_NOTRANS))) {
      // replicated layers, by output columns.  Left: Cout[:, cols] = alpha op(T) B[:, cols].  Right with an upper, untransposed T: column j
      // of the product only meets rows <= j of T: Cout[:, c0:c1) = alpha ( B[:, 0:c0) T[0:c0, c0:c1) + B[:, c0:c1) T[c0:c1, c0:c1) )
      colsplit_run(t, N, side == CAPI_RIGHT, Cout.p, Cout.ld,
                   [&](int64_t c0, int64_t c1) {
                     double* Cj = Cout.p + c0 * Cout.ld;
                     if (side == CAPI_LEFT) {
                       CAPITAL_CHECK(capi_dtrmm_oop(h, side, uplo, trans, diag, M, c1 - c0, alpha, T.p, T.ld, B.p + c0 * B.ld, B.ld, Cj, Cout.ld));
                     } else {
                       if (c0 > 0)
                         CAPITAL_CHECK(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, M, c1 - c0, c0, alpha, B.p, B.ld, T.p + c0 * T.ld, T.ld, 0.0, Cj, Cout.ld));
                       CAPITAL_CHECK(capi_dtrmm_acc(h, side, uplo, trans, diag, M, c1 - c0, alpha, T.p + c0 + c0 * T.ld, T.ld, B.p + c0 * B.ld, B.ld,
                                                    c0 > 0 ? 1.0 : 0.0, Cj, Cout.ld));
                     }
                   },
                   [&](int64_t) { return M; }, ws);
      return;
    }
    const int64_t mark = ws.top;
    const size_t steps = t.d > 1 ? t.d / t.c : 1;
    // The output-column pipeline serves the one-step cases (c == d as in the reference, or d == 1): every left multiply, and the
    // right multiply with an upper, untransposed T (the inverse completion, cholinv.hpp:150-154; cacqr.hpp:108-112) -- there column
    // chunk j of the output needs B's column chunks 0..j, which arrive in that order.
    const bool piped_left = side == CAPI_LEFT;
    const bool piped_right = side == CAPI_RIGHT && eff_upper && trans == CAPI_NOTRANS && t.d > 1;
    const bool piped = t.num_chunks > 0 && (piped_left || piped_right) && Cout.contiguous();
    const int nch = piped ? chunk_count(t.num_chunks, N) : 1;
    pipe P(piped && nch > 1);
    enum { E0 = 0, ET = 1, EB = 2, EC = 70, ER = 140 };
    int64_t cmax = N;
    if (P.on) { cmax = 0; for (int j = 0; j < nch; ++j) { int64_t c0, c1; chunk_range(N, nch, j, c0, c1); cmax = std::max(cmax, c1 - c0); } }
    if (t.d == 1) {
      // K-slice [k0,k1) of a triangular operand: a triangle on the diagonal plus a rectangle beside it
      const int64_t K = T.rows;
      int64_t k0, k1;
      kslice(K, t.c, t.z, k0, k1);
      for (int j = 0; j < nch; ++j) {
        int64_t c0 = 0, c1 = N;
        if (nch > 1) chunk_range(N, nch, j, c0, c1);
        const int64_t nc = c1 - c0;
        if (nc <= 0) continue;
        if (side == CAPI_LEFT) {
          const double* Bj = B.p + c0 * B.ld;
          double* Cj = Cout.p + c0 * Cout.ld;
          CAPITAL_CHECK(capi_dgeadd(h, 0, M, nc, 0.0, Cj, Cout.ld, 0.0, Cj, Cout.ld));   // Cj = 0
          if (k1 > k0) {
            const double* Tkk = T.p + k0 + k0 * T.ld;
            // rows of op(T) that see K-columns [k0,k1): the diagonal block rows and, above (eff upper) or below them, a rectangle
            CAPITAL_CHECK(capi_dtrmm_oop(h, side, uplo, trans, diag, k1 - k0, nc, alpha, Tkk, T.ld, Bj + k0, B.ld, Cj + k0, Cout.ld));
            const int64_t r0 = eff_upper ? 0 : k1, r1 = eff_upper ? k0 : K;
            if (r1 > r0) {
              // E[r0:r1, k0:k1] = trans ? T[k0:k1, r0:r1]^T : T[r0:r1, k0:k1]
              const double* Tr = trans ? T.p + k0 + r0 * T.ld : T.p + r0 + k0 * T.ld;
              CAPITAL_CHECK(capi_dgemm(h, trans, CAPI_NOTRANS, r1 - r0, nc, k1 - k0, alpha, Tr, T.ld, Bj + k0, B.ld, 0.0, Cj + r0, Cout.ld));
            }
          }
          P.rec(EC + j);
        } else {
          CAPITAL_CHECK(capi_dgeadd(h, 0, M, N, 0.0, Cout.p, Cout.ld, 0.0, Cout.p, Cout.ld));
          if (k1 > k0) {
            const double* Tkk = T.p + k0 + k0 * T.ld;
            CAPITAL_CHECK(capi_dtrmm_oop(h, side, uplo, trans, diag, M, k1 - k0, alpha, Tkk, T.ld, B.p + k0 * B.ld, B.ld, Cout.p + k0 * Cout.ld, Cout.ld));
            const int64_t q0 = eff_upper ? k1 : 0, q1 = eff_upper ? K : k0;
            if (q1 > q0) {
              // E[k0:k1, q0:q1] = trans ? T[q0:q1, k0:k1]^T : T[k0:k1, q0:q1]
              const double* Tr = trans ? T.p + q0 + k0 * T.ld : T.p + k0 + q0 * T.ld;
              CAPITAL_CHECK(capi_dgemm(h, CAPI_NOTRANS, trans, M, q1 - q0, k1 - k0, alpha, B.p + k0 * B.ld, B.ld, Tr, T.ld, 0.0, Cout.p + q0 * Cout.ld, Cout.ld));
            }
          }
        }
      }
    } else if (!P.on) {
      for (size_t s = 0; s < steps; ++s) {
        view tt, bb;
        if (side == CAPI_LEFT) { tt = panel_tri(t, AX_ROW, s, T, uplo, ws); bb = panel(t, AX_COLUMN, s, B, ws); }
        else { bb = panel(t, AX_ROW, s, B, ws); tt = panel_tri(t, AX_COLUMN, s, T, uplo, ws); }
        CAPITAL_CHECK(capi_dtrmm_acc(h, side, uplo, trans, diag, M, N, alpha, tt.p, tt.ld, bb.p, bb.ld, s ? 1.0 : 0.0, Cout.p, Cout.ld));
      }
    } else {
      // every K-class step of this layer: the T panels first (whole, packed), then the B panels column chunk by column chunk on the
      // communication stream; chunk j of the output sums the steps' products
      const int axT = side == CAPI_LEFT ? AX_ROW : AX_COLUMN, axB = side == CAPI_LEFT ? AX_COLUMN : AX_ROW;
      P.main(); P.rec(E0);
      P.comm(); P.wait(E0);
      std::vector<view> tt(steps, T), bb(steps, view{nullptr, B.rows, B.rows, B.cols});
      for (size_t s = 0; s < steps; ++s) tt[s] = panel_tri(t, axT, s, T, uplo, ws);
      P.rec(ET);
      for (size_t s = 0; s < steps; ++s) {
        const bool rootB = (side == CAPI_LEFT ? t.y : t.x) == t.z + s * t.c;
        bb[s].p = (rootB && B.contiguous()) ? B.p : ws.take(B.count());
      }
      double* relay = relay_space(t, B.rows * cmax, ws);
      for (int j = 0; j < nch; ++j) {
        int64_t c0, c1;
        chunk_range(N, nch, j, c0, c1);
        if (c1 > c0)
          for (size_t s = 0; s < steps; ++s) {
            const bool rootB = (side == CAPI_LEFT ? t.y : t.x) == t.z + s * t.c;
            if (rootB && !B.contiguous()) CAPITAL_CHECK(capi_dlacpy(h, 0, B.rows, c1 - c0, B.p + c0 * B.ld, B.ld, bb[s].p + c0 * bb[s].ld, bb[s].ld));
            bcast_axis(t, axB, s, bb[s].p + c0 * bb[s].ld, bb[s].rows * (c1 - c0), relay);
          }
        P.rec(EB + j);
      }
      P.main(); P.wait(ET);
      for (int j = 0; j < nch; ++j) {
        int64_t c0, c1;
        chunk_range(N, nch, j, c0, c1);
        P.wait(EB + j);
        if (c1 > c0) {
          double* Cj = Cout.p + c0 * Cout.ld;
          for (size_t s = 0; s < steps; ++s) {
            const double b0 = s ? 1.0 : 0.0;
            if (side == CAPI_LEFT) {
              CAPITAL_CHECK(capi_dtrmm_acc(h, side, uplo, trans, diag, M, c1 - c0, alpha, tt[s].p, tt[s].ld, bb[s].p + c0 * bb[s].ld, bb[s].ld, b0, Cj, Cout.ld));
            } else {
              // Cout[:, chunk] += alpha ( B[:, 0:c0] T[0:c0, chunk] + B[:, chunk] T[chunk, chunk] ): a rectangle above the chunk's triangle
              if (c0 > 0)
                CAPITAL_CHECK(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, M, c1 - c0, c0, alpha, bb[s].p, bb[s].ld, tt[s].p + c0 * tt[s].ld, tt[s].ld, b0, Cj, Cout.ld));
              CAPITAL_CHECK(capi_dtrmm_acc(h, side, uplo, trans, diag, M, c1 - c0, alpha, tt[s].p + c0 + c0 * tt[s].ld, tt[s].ld, bb[s].p + c0 * bb[s].ld, bb[s].ld,
                                           (c0 > 0 || s) ? 1.0 : 0.0, Cj, Cout.ld));
            }
          }
        }
        P.rec(EC + j);
      }
    }
    if (t.c > 1) {
      if (P.on) {
        P.comm();
        double* relay = relay_space(t, Cout.rows * cmax, ws);
        double* half = depth_space(t, Cout.rows * cmax, ws);
        for (int j = 0; j < nch; ++j) {
          int64_t c0, c1;
          chunk_range(N, nch, j, c0, c1);
          P.wait(EC + j);
          if (c1 > c0) allreduce_depth(t, Cout.p + c0 * Cout.ld, Cout.rows * (c1 - c0), half, relay);
        }
        P.rec(ER);
        P.main(); P.wait(ER);
      } else {
        allreduce_view(t, Cout, ws);
      }
    } else if (P.on) {
      P.main();
    }
    ws.top = mark;
  }

  // C(uplo) <- alpha * Bx^T * A + beta*C (trans) or alpha * A * Bx^T + beta*C; Bx = partner-exchanged block (summa.hpp:111-158)
  template <typename CommType>
  static void syrk(CommType&& t, int uplo, int trans, double alpha, view A, view Bx, double beta, view C, arena& ws) {
    capi_handle_t h = capital::handle();
    const int64_t N = C.cols, K = trans ? A.rows : A.cols;
    // operand order of the local multiply (summa.hpp:137-146)
    const int tA = trans ? CAPI_TRANS : CAPI_NOTRANS, tB = trans ? CAPI_NOTRANS : CAPI_TRANS;
    if (t.d == 1 && t.c == 1) {
      const view& L = trans ? Bx : A;
      const view& Rr = trans ? A : Bx;
      CAPITAL_CHECK(capi_dgemmt(h, uplo, tA, tB, N, K, alpha, L.p, L.ld, Rr.p, Rr.ld, beta, C.p, C.ld));
      return;
    }
    if (t.d == 1 && colsplit() && trans && uplo == CAPI_UPPER) {
      // replicated layers: layer z forms its column range of the upper triangle in place (the rectangle above the range's diagonal block +
      // the block), rows 0 .. c1 of those columns travel
      colsplit_run(t, N, true, C.p, C.ld,
                   [&](int64_t c0, int64_t c1) {
                     const double* rj = A.p + c0 * A.ld;
                     if (c0 > 0)
                       CAPITAL_CHECK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, c0, c1 - c0, K, alpha, Bx.p, Bx.ld, rj, A.ld, beta, C.p + c0 * C.ld, C.ld));
                     CAPITAL_CHECK(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, c1 - c0, K, alpha, Bx.p + c0 * Bx.ld, Bx.ld, rj, A.ld, beta,
                                               C.p + c0 + c0 * C.ld, C.ld));
                   },
                   [&](int64_t c1) { return c1; }, ws);
      return;
    }
    const int64_t mark = ws.top;
    const size_t steps = t.d > 1 ? t.d / t.c : 1;
    // the pipeline is built for what cholinv issues: upper triangle, transposed form, one K-class per layer
    const bool piped = t.num_chunks > 0 && trans && uplo == CAPI_UPPER;
    const int nch = piped ? chunk_count(t.num_chunks, N) : 1;
    pipe P(piped && nch > 1);
    enum { E0 = 0, EL = 1, EB = 2, EC = 70, ER = 140 };
    view acc = t.c > 1 ? view{ws.take(N * N), N, N, N} : C;
    if (!P.on) {
      if (t.d == 1) {
        int64_t k0, k1;
        kslice(K, t.c, t.z, k0, k1);
        const view& L = trans ? Bx : A;
        const view& Rr = trans ? A : Bx;
        const double* l = trans ? L.p + k0 : L.p + k0 * L.ld;
        const double* r = trans ? Rr.p + k0 : Rr.p + k0 * Rr.ld;
        CAPITAL_CHECK(capi_dgemmt(h, uplo, tA, tB, N, k1 - k0, alpha, l, L.ld, r, Rr.ld, 0.0, acc.p, acc.ld));
      } else {
        for (size_t s = 0; s < steps; ++s) {
          view l, r;
          if (trans) { l = panel(t, AX_ROW, s, Bx, ws); r = panel(t, AX_COLUMN, s, A, ws); }   // distribute(B,A)
          else { l = panel(t, AX_ROW, s, A, ws); r = panel(t, AX_COLUMN, s, Bx, ws); }        // distribute(A,B)
          const double bt = (t.c > 1) ? (s ? 1.0 : 0.0) : (s ? 1.0 : beta);
          CAPITAL_CHECK(capi_dgemmt(h, uplo, tA, tB, N, K, alpha, l.p, l.ld, r.p, r.ld, bt, acc.p, acc.ld));
        }
      }
      if (t.c > 1) {
        // only the computed triangle is folded into C (and only it travels)
        allreduce_tri(t, acc, uplo, ws);
        CAPITAL_CHECK(capi_dgeadd(h, uplo == CAPI_UPPER ? 1 : 2, N, N, 1.0, acc.p, acc.ld, beta, C.p, C.ld));
      }
      ws.top = mark;
      return;
    }
    // ---- pipelined: C(upper)[:, chunk] = alpha * sum over this layer's K-class steps of L_s[:, 0:c1]^T * R_s[:, chunk]; K x N, k-contiguous
    int64_t k0 = 0, k1 = K;
    std::vector<view> l(steps, Bx), r(steps, A);
    const bool sliced = (t.d == 1);
    int64_t cmax = 0;
    for (int j = 0; j < nch; ++j) { int64_t c0, c1; chunk_range(N, nch, j, c0, c1); cmax = std::max(cmax, c1 - c0); }
    P.main(); P.rec(E0);
    P.comm(); P.wait(E0);
    if (sliced) {
      kslice(K, t.c, t.z, k0, k1);
    } else {
      for (size_t s = 0; s < steps; ++s) {
        l[s] = panel(t, AX_ROW, s, Bx, ws);
        const bool rootR = t.y == t.z + s * t.c;
        r[s] = view{(rootR && A.contiguous()) ? A.p : ws.take(A.count()), A.rows, A.rows, A.cols};
      }
    }
    P.rec(EL);
    double* relay = relay_space(t, std::max(A.rows, N) * cmax, ws);
    for (int j = 0; j < nch; ++j) {
      int64_t c0, c1;
      chunk_range(N, nch, j, c0, c1);
      if (!sliced && c1 > c0)
        for (size_t s = 0; s < steps; ++s) {
          const bool rootR = t.y == t.z + s * t.c;
          if (rootR && !A.contiguous()) CAPITAL_CHECK(capi_dlacpy(h, 0, A.rows, c1 - c0, A.p + c0 * A.ld, A.ld, r[s].p + c0 * r[s].ld, r[s].ld));
          bcast_axis(t, AX_COLUMN, s, r[s].p + c0 * r[s].ld, r[s].rows * (c1 - c0), relay);
        }
      P.rec(EB + j);
    }
    P.main(); P.wait(EL);
    if (t.c > 1) capital::dev_zero(acc.p, acc.count());       // (the part below the computed trapezoids is folded into C as zeros)
    for (int j = 0; j < nch; ++j) {
      int64_t c0, c1;
      chunk_range(N, nch, j, c0, c1);
      P.wait(EB + j);
      if (c1 > c0)
        for (size_t s = 0; s < steps; ++s) {
          const double bt = s ? 1.0 : (t.c > 1 ? 0.0 : beta);
          const double* rj = r[s].p + k0 + c0 * r[s].ld;
          if (c0 > 0)   // rows above the diagonal block of this chunk: a plain rectangle
            CAPITAL_CHECK(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, c0, c1 - c0, k1 - k0, alpha, l[s].p + k0, l[s].ld, rj, r[s].ld, bt, acc.p + c0 * acc.ld, acc.ld));
          CAPITAL_CHECK(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, c1 - c0, k1 - k0, alpha, l[s].p + k0 + c0 * l[s].ld, l[s].ld, rj, r[s].ld, bt,
                                    acc.p + c0 + c0 * acc.ld, acc.ld));
        }
      P.rec(EC + j);
    }
    if (t.c > 1) {
      P.comm();
      double* half = depth_space(t, acc.rows * cmax, ws);
      for (int j = 0; j < nch; ++j) {
        int64_t c0, c1;
        chunk_range(N, nch, j, c0, c1);
        P.wait(EC + j);
        // (rows 0 .. c1 of the chunk's columns hold the computed part; the zeros below it are not sent)
        if (c1 > c0) allreduce_cols(t, acc.p + c0 * acc.ld, acc.ld, c1, c1 - c0, half, relay, ws);
      }
      P.rec(ER);
      P.main(); P.wait(ER);
      CAPITAL_CHECK(capi_dgeadd(h, 1, N, N, 1.0, acc.p, acc.ld, beta, C.p, C.ld));
    } else {
      P.main();
    }
    ws.top = mark;
  }

  static arena& scratch_arena() {
    static CAPITAL_RANK_LOCAL arena a;
    return a;
  }
  template <typename MatrixType>
  static view as_view(MatrixType& m) { return view{m.data(), m.num_rows_local(), m.num_rows_local(), m.num_columns_local()}; }

  // [k0,k1): slice z of c equal (even-sized, for 16-byte alignment) pieces of a local K range
  static void kslice(int64_t K, size_t c, size_t z, int64_t& k0, int64_t& k1) {
    int64_t per = ((K + (int64_t)c - 1) / (int64_t)c + 1) & ~(int64_t)1;
    k0 = std::min<int64_t>(K, per * (int64_t)z);
    k1 = std::min<int64_t>(K, k0 + per);
  }

  // ---- replicated layers (d == 1, c > 1: the 2-GPU grid) by OUTPUT COLUMNS ------------------------------------------------------------
  // Every layer holds the whole operands.  Round 2 sliced the K range and all-reduced the full outputs over `depth` (h^2 doubles in each
  // direction per product, plus an accumulator and an add pass).  Here layer z computes the output columns [b_z, b_z+1) -- all of K, in
  // place, with the caller's beta -- and the blocks are exchanged (c broadcasts over `depth`, both directions of the link at once with
  // c = 2): half the bytes of the all-reduce, no accumulator, and every element is computed exactly once, by the kernel and in the
  // summation order of the single-GPU run.  Triangular work (a TRMM with the triangle on the right, a triangular output) grows linearly
  // with the column index, so the boundaries sit at N sqrt(z / c).  CAPITAL_KSLICE=1 restores the K-slicing (A/B).
  static bool colsplit() { return getenv("CAPITAL_KSLICE") == nullptr; }      // (read per multiply: bench.py probes both forms in one process)
  static int64_t colsplit_bound(int64_t N, size_t c, size_t z, bool triangular) {
    if (z == 0) return 0;
    if (z >= c) return N;
    const double f = triangular ? std::sqrt((double)z / (double)c) : (double)z / (double)c;
    const int64_t b = ((int64_t)std::llround(f * (double)N) + 1) & ~(int64_t)1;
    return std::min<int64_t>(N, std::max<int64_t>(0, b));
  }
  // compute(c0, c1) enqueues the kernels that finish output columns [c0, c1) in place at `base` (leading dimension ld); rows_of(c1) = how
  // many leading rows of those columns carry the result (all of them, or c1 for an upper-triangular output).  With num_chunks > 0 the
  // exchange of chunk j (communication stream) runs beside the computation of chunk j + 1.
  template <typename CommType, typename Compute, typename RowsOf>
  static void colsplit_run(CommType&& t, int64_t N, bool triangular, double* base, int64_t ld, Compute&& compute, RowsOf&& rows_of, arena& ws) {
    capi_handle_t h = capital::handle();
    const int64_t mark = ws.top;
    std::vector<int64_t> b(t.c + 1);
    int64_t widest = 0;
    for (size_t z = 0; z <= t.c; ++z) b[z] = colsplit_bound(N, t.c, z, triangular);
    for (size_t z = 0; z < t.c; ++z) widest = std::max(widest, b[z + 1] - b[z]);
    const int nch = t.num_chunks > 0 ? chunk_count(t.num_chunks, widest) : 1;
    pipe P(nch > 1);
    enum { EC = 70, ER = 140 };
    double* tmp = nullptr;                                       // staging of a strided block (allocated once: it is used on the communication stream)
    int64_t tmp_need = 0;
    for (size_t z = 0; z < t.c; ++z)
      for (int j = 0; j < nch; ++j) {
        int64_t c0, c1;
        chunk_range(b[z + 1] - b[z], nch, j, c0, c1);
        const int64_t rows = rows_of(b[z] + c1);
        if (rows != ld) tmp_need = std::max(tmp_need, rows * (c1 - c0));
      }
    if (tmp_need > 0) tmp = ws.take(tmp_need);
    P.main();
    for (int j = 0; j < nch; ++j) {
      int64_t c0, c1;
      chunk_range(b[t.z + 1] - b[t.z], nch, j, c0, c1);
      if (c1 > c0) compute(b[t.z] + c0, b[t.z] + c1);
      P.rec(EC + j);
    }
    P.comm();
    for (int j = 0; j < nch; ++j) {
      P.wait(EC + j);
      for (size_t z = 0; z < t.c; ++z) {
        int64_t c0, c1;
        chunk_range(b[z + 1] - b[z], nch, j, c0, c1);
        if (c1 <= c0) continue;
        const int64_t g0 = b[z] + c0, g1 = b[z] + c1, rows = rows_of(g1);
        double* blk = base + g0 * ld;
        if (rows == ld) {
          CAPITAL_CHECK(capi_bcast(t.depth, blk, rows * (g1 - g0), (int)z));
        } else {
          if (t.z == z) CAPITAL_CHECK(capi_dlacpy(h, 0, rows, g1 - g0, blk, ld, tmp, rows));
          CAPITAL_CHECK(capi_bcast(t.depth, tmp, rows * (g1 - g0), (int)z));
          if (t.z != z) CAPITAL_CHECK(capi_dlacpy(h, 0, rows, g1 - g0, tmp, rows, blk, ld));
        }
      }
    }
    P.rec(ER);
    P.main(); P.wait(ER);
    ws.top = mark;
  }

  enum { AX_ROW = 0, AX_COLUMN = 1 };

  // ---- pair collectives: over every link of the node (capi_pairs_transfer, csrc/pair_paths.h) when the grid's rows / columns /
  //      depth fibres are pairs, else as RCCL calls on the sub-communicators ---------------------------------------------------------
  // relay space for transfers of up to `count` doubles (nullptr when this grid does not relay)
  template <typename CommType>
  static double* relay_space(CommType&& t, int64_t count, arena& ws) {
    if (!t.multipath) return nullptr;
    const int64_t n = capi_pairs_scratch_count(t.size, count);
    return n > 0 ? ws.take(n) : nullptr;
  }
  // landing space for the partner's half in a pair all-reduce of up to `count` doubles
  template <typename CommType>
  static double* depth_space(CommType&& t, int64_t count, arena& ws) { return t.pairs_in_depth() ? ws.take(count / 2 + 2) : nullptr; }

  // MPI_Bcast(row | column) of `count` doubles at `buf`, root = the member whose x (y) is this layer's K-class at step s (summa.hpp:185,193)
  template <typename CommType>
  static void bcast_axis(CommType&& t, int axis, size_t s, double* buf, int64_t count, double* relay) {
    const size_t q = t.z + s * t.c;
    if (t.pairs_along(axis)) {
      const bool root = (axis == AX_ROW ? t.x : t.y) == q;
      const std::vector<int> dst = t.bcast_dst(axis, s);
      CAPITAL_CHECK(capi_pairs_transfer(t.world, dst.data(), root ? buf : nullptr, root ? nullptr : buf, count, relay));
    } else {
      CAPITAL_CHECK(capi_bcast(axis == AX_ROW ? t.row : t.column, buf, count, (int)q));
    }
  }
  // MPI_Allreduce(depth, SUM) in place (summa.hpp:236).  On a depth PAIR: each member keeps one half, the halves that are not kept
  // cross over (all links), are added, and the sums cross back -- the bytes of a two-rank ring, at the multi-path rate.  a + b is
  // commutative in IEEE arithmetic, so both members end with bit-identical sums.
  template <typename CommType>
  static void allreduce_depth(CommType&& t, double* buf, int64_t count, double* half_space, double* relay) {
    const int64_t c4 = count & ~(int64_t)3, half = c4 / 2;
    if (!t.pairs_in_depth() || half == 0) { CAPITAL_CHECK(capi_allreduce_sum(t.depth, buf, count)); return; }
    const std::vector<int> dst = t.depth_dst();
    double* keep = buf + (t.z == 0 ? 0 : half);
    double* give = buf + (t.z == 0 ? half : 0);
    CAPITAL_CHECK(capi_pairs_transfer(t.world, dst.data(), give, half_space, half, relay));
    CAPITAL_CHECK(capi_daxpby(capital::handle(), half, 1.0, half_space, keep));
    CAPITAL_CHECK(capi_pairs_transfer(t.world, dst.data(), keep, give, half, relay));
    if (count > c4) CAPITAL_CHECK(capi_allreduce_sum(t.depth, buf + c4, count - c4));     // (at most three trailing values)
  }
  template <typename CommType>
  static void allreduce_depth(CommType&& t, double* buf, int64_t count, arena& ws) {
    const int64_t mark = ws.top;
    double* half = depth_space(t, count, ws);
    double* relay = t.pairs_in_depth() ? relay_space(t, count / 2 + 2, ws) : nullptr;
    allreduce_depth(t, buf, count, half, relay);
    ws.top = mark;
  }
  // sum over depth of `rows` x `cols` values at p (leading dimension ld); a strided block travels as a contiguous copy
  // (half_space / relay sized for rows * cols; arena space taken here stays taken until the caller resets its mark: the copies
  //  live on the communication stream of a pipelined multiply)
  template <typename CommType>
  static void allreduce_cols(CommType&& t, double* p, int64_t ld, int64_t rows, int64_t cols, double* half_space, double* relay, arena& ws) {
    if (rows == ld) { allreduce_depth(t, p, rows * cols, half_space, relay); return; }
    capi_handle_t h = capital::handle();
    double* tmp = ws.take(rows * cols);
    CAPITAL_CHECK(capi_dlacpy(h, 0, rows, cols, p, ld, tmp, rows));
    allreduce_depth(t, tmp, rows * cols, half_space, relay);
    CAPITAL_CHECK(capi_dlacpy(h, 0, rows, cols, tmp, rows, p, ld));
  }

  // The panel a rank multiplies with: the root's own block (packed to contiguous if it is a strided view) broadcast
  // over `comm`.  Non-roots receive into arena memory.  Blocks have equal shapes on all ranks of a communicator.
  static view panel(capi_comm_t comm, bool is_root, int root, const view& mine, arena& ws) {
    int size = 1;
    CAPITAL_CHECK(capi_comm_size(comm, &size));
    if (size == 1) return mine;
    view out{nullptr, mine.rows, mine.rows, mine.cols};
    if (is_root && mine.contiguous()) {
      out.p = mine.p;
    } else {
      out.p = ws.take(mine.count());
      if (is_root) CAPITAL_CHECK(capi_dlacpy(capital::handle(), 0, mine.rows, mine.cols, mine.p, mine.ld, out.p, out.ld));
    }
    CAPITAL_CHECK(capi_bcast(comm, out.p, out.count(), root));
    return out;
  }
  // the same along an axis of the grid at K-class step s (multi-path on grids of pairs)
  template <typename CommType>
  static view panel(CommType&& t, int axis, size_t s, const view& mine, arena& ws) {
    const size_t q = t.z + s * t.c;
    const bool is_root = (axis == AX_ROW ? t.x : t.y) == q;
    if (!t.pairs_along(axis)) return panel(axis == AX_ROW ? t.row : t.column, is_root, (int)q, mine, ws);
    view out{nullptr, mine.rows, mine.rows, mine.cols};
    if (is_root && mine.contiguous()) {
      out.p = mine.p;
    } else {
      out.p = ws.take(mine.count());
      if (is_root) CAPITAL_CHECK(capi_dlacpy(capital::handle(), 0, mine.rows, mine.cols, mine.p, mine.ld, out.p, out.ld));
    }
    const int64_t mark = ws.top;
    bcast_axis(t, axis, s, out.p, out.count(), relay_space(t, out.count(), ws));
    ws.top = mark;                                              // (relay space is reused in stream order)
    return out;
  }
  // A TRIANGULAR operand travels packed (n(n+1)/2 doubles instead of n^2, the reference's Serialize policy on the wire,
  // summa.hpp:147-148,216-217): the root packs, everyone unpacks the triangle into a full-stride buffer whose other half is
  // never read (the kernels truncate each tile's k-range at the diagonal and mask the diagonal panels by selection).
  template <typename CommType>
  static view panel_tri(CommType&& t, int axis, size_t s, const view& mine, int uplo, arena& ws) {
    capi_comm_t comm = axis == AX_ROW ? t.row : t.column;
    int size = 1;
    CAPITAL_CHECK(capi_comm_size(comm, &size));
    if (size == 1) return mine;
    static const bool off = getenv("CAPITAL_NO_PACKED_COMM") != nullptr;
    if (off || mine.rows != mine.cols) return panel(t, axis, s, mine, ws);
    capi_handle_t h = capital::handle();
    const size_t q = t.z + s * t.c;
    const bool is_root = (axis == AX_ROW ? t.x : t.y) == q;
    const int64_t n = mine.rows, np = n * (n + 1) / 2;
    const int st = uplo == CAPI_UPPER ? CAPI_UPPERTRI : CAPI_LOWERTRI;
    double* packed = ws.take(np);
    if (is_root) CAPITAL_CHECK(capi_serialize_shape(h, st, CAPI_RECT, st, mine.p, n, mine.ld, packed, n, n, 0, n, 0, n, 0, n, 0, n));
    {
      const int64_t mark = ws.top;
      bcast_axis(t, axis, s, packed, np, relay_space(t, np, ws));
      ws.top = mark;
    }
    if (is_root) return mine;                                   // the root multiplies with its own block
    view out{ws.take(n * n), n, n, n};
    CAPITAL_CHECK(capi_serialize_shape(h, st, st, CAPI_RECT, packed, n, n, out.p, n, n, 0, n, 0, n, 0, n, 0, n));
    return out;
  }
  // sum over depth of the `uplo` triangle of a contiguous n x n accumulator, packed on the wire
  template <typename CommType>
  static void allreduce_tri(CommType&& t, view acc, int uplo, arena& ws) {
    static const bool off = getenv("CAPITAL_NO_PACKED_COMM") != nullptr;
    if (off || acc.rows != acc.cols || !acc.contiguous()) { allreduce_depth(t, acc.p, acc.count(), ws); return; }
    capi_handle_t h = capital::handle();
    const int64_t n = acc.rows, np = n * (n + 1) / 2;
    const int st = uplo == CAPI_UPPER ? CAPI_UPPERTRI : CAPI_LOWERTRI;
    double* packed = ws.take(np);
    CAPITAL_CHECK(capi_serialize_shape(h, st, CAPI_RECT, st, acc.p, n, n, packed, n, n, 0, n, 0, n, 0, n, 0, n));
    allreduce_depth(t, packed, np, ws);
    CAPITAL_CHECK(capi_serialize_shape(h, st, st, CAPI_RECT, packed, n, n, acc.p, n, n, 0, n, 0, n, 0, n, 0, n));
  }
  template <typename CommType>
  static void allreduce_view(CommType&& t, view v, arena& ws) {
    if (v.contiguous()) { allreduce_depth(t, v.p, v.count(), ws); return; }
    double* tmp = ws.take(v.count());
    CAPITAL_CHECK(capi_dlacpy(capital::handle(), 0, v.rows, v.cols, v.p, v.ld, tmp, v.rows));
    allreduce_depth(t, tmp, v.count(), ws);
    CAPITAL_CHECK(capi_dlacpy(capital::handle(), 0, v.rows, v.cols, tmp, v.rows, v.p, v.ld));
  }
};

}  // namespace matmult

#endif  // CAPITAL_MATMULT_SUMMA_H_
