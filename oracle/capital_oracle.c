/*
 * capital_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See capital_oracle.h for scope, pinning status and the rule about who may call it.
 *
 * Every function cites the reference file:line whose behaviour it restates
 * (paths relative to the reference root).  The BLAS/LAPACK entries restate the
 * published semantics of the MKL calls the reference makes (Intel MKL is a
 * third-party dependency that is not vendored and has no pinned version:
 * config.mk:11, src/util/shared.h:24).
 */
#define _XOPEN_SOURCE 600
#define _DEFAULT_SOURCE
#include "capital_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MIN(a, b) ((a) < (b) ? (a) : (b))
#define MAX(a, b) ((a) > (b) ? (a) : (b))

static int g_threads = 0;
void orc_set_threads(int n) {
  g_threads = n;
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}
int orc_get_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Host-BLAS backend (bench.py's cpu_baseline leg, and the schedule-level pin of tests/golden).
 * The reference delegates ALL arithmetic to seven CBLAS / LAPACKE entry points of a host library
 * (src/blas/interface.hpp:54,74,92; src/lapack/interface.hpp:39,54): orc_host_blas_bind() loads such
 * a library at run time (MKL's libmkl_rt, OpenBLAS, or the OpenBLAS bundled with numpy / scipy,
 * whose symbols carry a prefix / suffix and, for the 64_ build, 64-bit integers) and from then on
 * orc_dgemm / orc_dtrmm / orc_dsyrk / orc_dpotrf / orc_dtrtri forward to it with the constants the
 * reference passes (column-major, interface.hpp:49-52).  The schedules above them are unchanged, so
 * "the reference's schedule on the node's own host BLAS" is what gets timed.
 * ------------------------------------------------------------------------------------------ */
#include <dlfcn.h>
#include <stdio.h>
enum { HB_COL = 102, HB_NOTRANS = 111, HB_TRANS = 112, HB_UPPER = 121, HB_LOWER = 122, HB_NONUNIT = 131, HB_UNIT = 132, HB_LEFT = 141, HB_RIGHT = 142 };
static struct {
  int on, ilp64;
  void* lib;
  void *gemm, *trmm, *syrk, *potrf, *trtri;
} g_hb;

int orc_host_blas_bind(const char* path, const char* prefix, const char* suffix, int ilp64) {
  void* lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!lib) return -1;
  const char* names[5] = {"cblas_dgemm", "cblas_dtrmm", "cblas_dsyrk", "LAPACKE_dpotrf", "LAPACKE_dtrtri"};
  void* f[5];
  for (int i = 0; i < 5; ++i) {
    char sym[128];
    snprintf(sym, sizeof(sym), "%s%s%s", prefix ? prefix : "", names[i], suffix ? suffix : "");
    f[i] = dlsym(lib, sym);
    if (!f[i]) { dlclose(lib); return -2 - i; }
  }
  g_hb.lib = lib; g_hb.ilp64 = ilp64;
  g_hb.gemm = f[0]; g_hb.trmm = f[1]; g_hb.syrk = f[2]; g_hb.potrf = f[3]; g_hb.trtri = f[4];
  g_hb.on = 1;
  return 0;
}
void orc_host_blas_enable(int on) { g_hb.on = on && g_hb.lib; }
int orc_host_blas_active(void) { return g_hb.on; }

#define HB_CALL(ITYPE)                                                                                              \
  static void hb_gemm_##ITYPE(int ta, int tb, int64_t m, int64_t n, int64_t k, double al, const double* A, int64_t lda, \
                              const double* B, int64_t ldb, double be, double* C, int64_t ldc) {                    \
    ((void (*)(int, int, int, ITYPE, ITYPE, ITYPE, double, const double*, ITYPE, const double*, ITYPE, double, double*, ITYPE))g_hb.gemm)( \
        HB_COL, ta ? HB_TRANS : HB_NOTRANS, tb ? HB_TRANS : HB_NOTRANS, (ITYPE)m, (ITYPE)n, (ITYPE)k, al, A, (ITYPE)lda, B, (ITYPE)ldb, be, C, (ITYPE)ldc); \
  }                                                                                                                 \
  static void hb_trmm_##ITYPE(int side, int uplo, int tr, int diag, int64_t m, int64_t n, double al, const double* T, \
                              int64_t ldt, double* B, int64_t ldb) {                                                \
    ((void (*)(int, int, int, int, int, ITYPE, ITYPE, double, const double*, ITYPE, double*, ITYPE))g_hb.trmm)(     \
        HB_COL, side ? HB_RIGHT : HB_LEFT, uplo ? HB_UPPER : HB_LOWER, tr ? HB_TRANS : HB_NOTRANS, diag ? HB_UNIT : HB_NONUNIT, \
        (ITYPE)m, (ITYPE)n, al, T, (ITYPE)ldt, B, (ITYPE)ldb);                                                      \
  }                                                                                                                 \
  static void hb_syrk_##ITYPE(int uplo, int tr, int64_t n, int64_t k, double al, const double* A, int64_t lda, double be, \
                              double* C, int64_t ldc) {                                                             \
    ((void (*)(int, int, int, ITYPE, ITYPE, double, const double*, ITYPE, double, double*, ITYPE))g_hb.syrk)(       \
        HB_COL, uplo ? HB_UPPER : HB_LOWER, tr ? HB_TRANS : HB_NOTRANS, (ITYPE)n, (ITYPE)k, al, A, (ITYPE)lda, be, C, (ITYPE)ldc); \
  }                                                                                                                 \
  static int hb_potrf_##ITYPE(int uplo, int64_t n, double* A, int64_t lda) {                                        \
    return (int)((ITYPE (*)(int, char, ITYPE, double*, ITYPE))g_hb.potrf)(HB_COL, uplo ? 'U' : 'L', (ITYPE)n, A, (ITYPE)lda); \
  }                                                                                                                 \
  static int hb_trtri_##ITYPE(int uplo, int diag, int64_t n, double* A, int64_t lda) {                              \
    return (int)((ITYPE (*)(int, char, char, ITYPE, double*, ITYPE))g_hb.trtri)(HB_COL, uplo ? 'U' : 'L', diag ? 'U' : 'N', (ITYPE)n, A, (ITYPE)lda); \
  }
HB_CALL(int)
HB_CALL(int64_t)

/* ------------------------------------------------------------------------------------------
 * K1-K3  dgemm  (cblas_dgemm, src/blas/interface.hpp:43-59; call sites summa.hpp:28-30,139-145,
 *                cacqr.hpp:95-96)
 *   C <- alpha*op(A)*op(B) + beta*C,  op(A) m x k, op(B) k x n, column-major.
 * Packed / register-blocked so that the cpu_baseline leg is a fair "port" baseline.
 * ------------------------------------------------------------------------------------------ */
#define MR 8
#define NR 6
#define KC 256
#define MC 96
#define NC 192

typedef double v4d __attribute__((vector_size(32), aligned(8)));

__attribute__((target_clones("arch=haswell", "default")))
static void micro_kernel(int64_t kc, const double* restrict Ap, const double* restrict Bp, double* restrict acc /* MR*NR col-major */) {
  v4d c0[NR], c1[NR];
  for (int j = 0; j < NR; ++j) { c0[j] = (v4d){0, 0, 0, 0}; c1[j] = (v4d){0, 0, 0, 0}; }
  for (int64_t p = 0; p < kc; ++p) {
    v4d a0 = *(const v4d*)(Ap + p * MR);
    v4d a1 = *(const v4d*)(Ap + p * MR + 4);
    const double* b = Bp + p * NR;
    for (int j = 0; j < NR; ++j) {
      v4d bj = {b[j], b[j], b[j], b[j]};
      c0[j] += a0 * bj;
      c1[j] += a1 * bj;
    }
  }
  for (int j = 0; j < NR; ++j) {
    *(v4d*)(acc + j * MR) = c0[j];
    *(v4d*)(acc + j * MR + 4) = c1[j];
  }
}

static inline double opelem(const double* X, int64_t ld, int trans, int64_t r, int64_t c) {
  return trans ? X[c + r * ld] : X[r + c * ld];
}

/* pack op(A)[i0:i0+mc, p0:p0+kc] into MR-row panels, zero padded */
static void pack_A(const double* A, int64_t lda, int transA, int64_t i0, int64_t mc, int64_t p0, int64_t kc, double* Ap) {
  for (int64_t ir = 0; ir < mc; ir += MR) {
    int64_t mr = MIN(MR, mc - ir);
    for (int64_t p = 0; p < kc; ++p) {
      double* dst = Ap + ir * kc + p * MR;
      if (!transA) {
        const double* src = A + (i0 + ir) + (p0 + p) * lda;
        for (int64_t i = 0; i < mr; ++i) dst[i] = src[i];
      } else {
        for (int64_t i = 0; i < mr; ++i) dst[i] = A[(p0 + p) + (i0 + ir + i) * lda];
      }
      for (int64_t i = mr; i < MR; ++i) dst[i] = 0.0;
    }
  }
}
/* pack op(B)[p0:p0+kc, j0:j0+nc] into NR-column panels, zero padded */
static void pack_B(const double* B, int64_t ldb, int transB, int64_t p0, int64_t kc, int64_t j0, int64_t nc, double* Bp) {
  for (int64_t jr = 0; jr < nc; jr += NR) {
    int64_t nr = MIN(NR, nc - jr);
    for (int64_t p = 0; p < kc; ++p) {
      double* dst = Bp + jr * kc + p * NR;
      for (int64_t j = 0; j < nr; ++j) dst[j] = opelem(B, ldb, transB, p0 + p, j0 + jr + j);
      for (int64_t j = nr; j < NR; ++j) dst[j] = 0.0;
    }
  }
}

void orc_dgemm(int transA, int transB, int64_t m, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, const double* B, int64_t ldb,
               double beta, double* C, int64_t ldc) {
  if (m <= 0 || n <= 0) return;
  if (g_hb.on) { (g_hb.ilp64 ? hb_gemm_int64_t : hb_gemm_int)(transA, transB, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc); return; }
  /* beta scaling first (BLAS: beta==0 means C is not read) */
  if (beta != 1.0) {
#pragma omp parallel for schedule(static) if (m * n > 65536)
    for (int64_t j = 0; j < n; ++j) {
      double* c = C + j * ldc;
      if (beta == 0.0) for (int64_t i = 0; i < m; ++i) c[i] = 0.0;
      else for (int64_t i = 0; i < m; ++i) c[i] *= beta;
    }
  }
  if (k <= 0 || alpha == 0.0) return;
  int64_t mblocks = (m + MC - 1) / MC, nblocks = (n + NC - 1) / NC;
  int64_t ntasks = mblocks * nblocks;
#pragma omp parallel if (ntasks > 1 && m * n * k > 200000)
  {
    double* Ap = (double*)aligned_alloc(64, sizeof(double) * (MC + MR) * KC);
    double* Bp = (double*)aligned_alloc(64, sizeof(double) * (NC + NR) * KC);
    double acc[MR * NR] __attribute__((aligned(64)));
#pragma omp for schedule(dynamic, 1)
    for (int64_t t = 0; t < ntasks; ++t) {
      int64_t bi = t % mblocks, bj = t / mblocks;
      int64_t i0 = bi * MC, mc = MIN(MC, m - i0);
      int64_t j0 = bj * NC, nc = MIN(NC, n - j0);
      for (int64_t p0 = 0; p0 < k; p0 += KC) {
        int64_t kc = MIN(KC, k - p0);
        pack_A(A, lda, transA, i0, mc, p0, kc, Ap);
        pack_B(B, ldb, transB, p0, kc, j0, nc, Bp);
        for (int64_t jr = 0; jr < nc; jr += NR) {
          int64_t nr = MIN(NR, nc - jr);
          for (int64_t ir = 0; ir < mc; ir += MR) {
            int64_t mr = MIN(MR, mc - ir);
            micro_kernel(kc, Ap + ir * kc, Bp + jr * kc, acc);
            for (int64_t j = 0; j < nr; ++j) {
              double* c = C + (i0 + ir) + (j0 + jr + j) * ldc;
              for (int64_t i = 0; i < mr; ++i) c[i] += alpha * acc[i + j * MR];
            }
          }
        }
      }
    }
    free(Ap);
    free(Bp);
  }
}

/* ------------------------------------------------------------------------------------------
 * K4-K6  dtrmm  (cblas_dtrmm, src/blas/interface.hpp:61-79; call sites summa.hpp:64,71,
 *                cacqr.hpp:25,185-187).  Only the `uplo` triangle of T is referenced.
 * Blocked: E = op(T) is "effectively upper" when (uplo==Upper) xor trans.
 * ------------------------------------------------------------------------------------------ */
#define TB 128

/* dense copy of block (bi,bj) of E=op(T) restricted to the referenced triangle, into W (TB x TB, ld TB) */
static void tri_block_dense(const double* T, int64_t ldt, int uplo, int trans, int diag,
                            int64_t r0, int64_t nr, int64_t c0, int64_t nc, double* W) {
  for (int64_t j = 0; j < nc; ++j)
    for (int64_t i = 0; i < nr; ++i) {
      int64_t er = r0 + i, ec = c0 + j;           /* element of E */
      int64_t tr = trans ? ec : er, tc = trans ? er : ec; /* element of T */
      double v;
      if (tr == tc) v = diag == ORC_UNIT ? 1.0 : T[tr + tc * ldt];
      else if ((uplo == ORC_UPPER && tr < tc) || (uplo == ORC_LOWER && tr > tc)) v = T[tr + tc * ldt];
      else v = 0.0;
      W[i + j * TB] = v;
    }
}

void orc_dtrmm(int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb) {
  if (m <= 0 || n <= 0) return;
  if (g_hb.on) { (g_hb.ilp64 ? hb_trmm_int64_t : hb_trmm_int)(side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb); return; }
  int eff_upper = ((uplo == ORC_UPPER) != (trans == ORC_TRANS));
  int64_t nt = side == ORC_LEFT ? m : n;      /* order of T */
  int64_t nb = (nt + TB - 1) / TB;
  double* W = (double*)malloc(sizeof(double) * TB * TB);
  if (side == ORC_LEFT) {
    double* tmp = (double*)malloc(sizeof(double) * TB * n);
    /* B_i = sum_k E_ik B_k ; k>=i (eff upper): ascending i keeps B_k (k>i) original */
    for (int64_t s = 0; s < nb; ++s) {
      int64_t bi = eff_upper ? s : nb - 1 - s;
      int64_t i0 = bi * TB, mi = MIN(TB, m - i0);
      tri_block_dense(T, ldt, uplo, trans, diag, i0, mi, i0, mi, W);
      orc_dgemm(ORC_NOTRANS, ORC_NOTRANS, mi, n, mi, alpha, W, TB, B + i0, ldb, 0.0, tmp, TB);
      int64_t klo = eff_upper ? i0 + mi : 0, khi = eff_upper ? m : i0;
      if (khi > klo) {
        /* E[i0:i0+mi, klo:khi] = trans ? T[klo:khi, i0:..]^T : T[i0:.., klo:khi] */
        if (!trans) orc_dgemm(ORC_NOTRANS, ORC_NOTRANS, mi, n, khi - klo, alpha, T + i0 + klo * ldt, ldt, B + klo, ldb, 1.0, tmp, TB);
        else        orc_dgemm(ORC_TRANS,   ORC_NOTRANS, mi, n, khi - klo, alpha, T + klo + i0 * ldt, ldt, B + klo, ldb, 1.0, tmp, TB);
      }
      for (int64_t j = 0; j < n; ++j) memcpy(B + i0 + j * ldb, tmp + j * TB, sizeof(double) * mi);
    }
    free(tmp);
  } else {
    double* tmp = (double*)malloc(sizeof(double) * m * TB);
    /* B_j = sum_k B_k E_kj ; k<=j (eff upper): descending j keeps B_k (k<j) original */
    for (int64_t s = 0; s < nb; ++s) {
      int64_t bj = eff_upper ? nb - 1 - s : s;
      int64_t j0 = bj * TB, nj = MIN(TB, n - j0);
      tri_block_dense(T, ldt, uplo, trans, diag, j0, nj, j0, nj, W);
      orc_dgemm(ORC_NOTRANS, ORC_NOTRANS, m, nj, nj, alpha, B + j0 * ldb, ldb, W, TB, 0.0, tmp, m);
      int64_t klo = eff_upper ? 0 : j0 + nj, khi = eff_upper ? j0 : n;
      if (khi > klo) {
        if (!trans) orc_dgemm(ORC_NOTRANS, ORC_NOTRANS, m, nj, khi - klo, alpha, B + klo * ldb, ldb, T + klo + j0 * ldt, ldt, 1.0, tmp, m);
        else        orc_dgemm(ORC_NOTRANS, ORC_TRANS,   m, nj, khi - klo, alpha, B + klo * ldb, ldb, T + j0 + klo * ldt, ldt, 1.0, tmp, m);
      }
      for (int64_t j = 0; j < nj; ++j) memcpy(B + (j0 + j) * ldb, tmp + j * m, sizeof(double) * m);
    }
    free(tmp);
  }
  free(W);
}

/* ------------------------------------------------------------------------------------------
 * K7  dsyrk  (cblas_dsyrk, src/blas/interface.hpp:81-97; only call site cacqr.hpp:14-15)
 *   trans==Trans: C <- alpha*A^T*A + beta*C, A is k x n; NoTrans: alpha*A*A^T, A is n x k.
 *   Only the `uplo` triangle of C is read or written.
 * ------------------------------------------------------------------------------------------ */
void orc_dsyrk(int uplo, int trans, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, double beta, double* C, int64_t ldc) {
  if (n <= 0) return;
  if (g_hb.on) { (g_hb.ilp64 ? hb_syrk_int64_t : hb_syrk_int)(uplo, trans, n, k, alpha, A, lda, beta, C, ldc); return; }
  const int64_t SB = 192;
  int64_t nb = (n + SB - 1) / SB;
  double* W = (double*)malloc(sizeof(double) * SB * SB);
  for (int64_t bj = 0; bj < nb; ++bj) {
    int64_t j0 = bj * SB, nj = MIN(SB, n - j0);
    for (int64_t bi = 0; bi < nb; ++bi) {
      int64_t i0 = bi * SB, ni = MIN(SB, n - i0);
      if ((uplo == ORC_UPPER && bi > bj) || (uplo == ORC_LOWER && bi < bj)) continue;
      const double* Ai = trans ? A + i0 * lda : A + i0;
      const double* Aj = trans ? A + j0 * lda : A + j0;
      int tA = trans ? ORC_TRANS : ORC_NOTRANS, tB = trans ? ORC_NOTRANS : ORC_TRANS;
      if (bi != bj) {
        orc_dgemm(tA, tB, ni, nj, k, alpha, Ai, lda, Aj, lda, beta, C + i0 + j0 * ldc, ldc);
      } else {
        orc_dgemm(tA, tB, ni, nj, k, alpha, Ai, lda, Aj, lda, 0.0, W, SB);
        for (int64_t j = 0; j < nj; ++j) {
          int64_t ilo = uplo == ORC_UPPER ? 0 : j, ihi = uplo == ORC_UPPER ? j + 1 : ni;
          double* c = C + i0 + (j0 + j) * ldc;
          for (int64_t i = ilo; i < ihi; ++i) c[i] = (beta == 0.0 ? 0.0 : beta * c[i]) + W[i + j * SB];
        }
      }
    }
  }
  free(W);
}

/* ------------------------------------------------------------------------------------------
 * dtrsm (checker for the product's extra TRSM; LAPACK reference semantics).
 * Column-by-column substitution; O(n^2 m) plain loops.
 * ------------------------------------------------------------------------------------------ */
void orc_dtrsm(int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb) {
  if (m <= 0 || n <= 0) return;
  for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < m; ++i) B[i + j * ldb] *= alpha;
  int eff_upper = ((uplo == ORC_UPPER) != (trans == ORC_TRANS));
#define EL(r, c) (trans ? T[(c) + (r) * ldt] : T[(r) + (c) * ldt]) /* E = op(T) */
  if (side == ORC_LEFT) {
    /* solve E X = B, E is m x m */
#pragma omp parallel for schedule(static) if (n > 8)
    for (int64_t j = 0; j < n; ++j) {
      double* b = B + j * ldb;
      if (eff_upper) {
        for (int64_t i = m - 1; i >= 0; --i) {
          double s = b[i];
          for (int64_t p = i + 1; p < m; ++p) s -= EL(i, p) * b[p];
          b[i] = diag == ORC_UNIT ? s : s / EL(i, i);
        }
      } else {
        for (int64_t i = 0; i < m; ++i) {
          double s = b[i];
          for (int64_t p = 0; p < i; ++p) s -= EL(i, p) * b[p];
          b[i] = diag == ORC_UNIT ? s : s / EL(i, i);
        }
      }
    }
  } else {
    /* solve X E = B, E is n x n : row i of X */
#pragma omp parallel for schedule(static) if (m > 8)
    for (int64_t i = 0; i < m; ++i) {
      if (eff_upper) {
        for (int64_t j = 0; j < n; ++j) {
          double s = B[i + j * ldb];
          for (int64_t p = 0; p < j; ++p) s -= B[i + p * ldb] * EL(p, j);
          B[i + j * ldb] = diag == ORC_UNIT ? s : s / EL(j, j);
        }
      } else {
        for (int64_t j = n - 1; j >= 0; --j) {
          double s = B[i + j * ldb];
          for (int64_t p = j + 1; p < n; ++p) s -= B[i + p * ldb] * EL(p, j);
          B[i + j * ldb] = diag == ORC_UNIT ? s : s / EL(j, j);
        }
      }
    }
  }
#undef EL
}

/* ------------------------------------------------------------------------------------------
 * K8  dpotrf  (LAPACKE_dpotrf col-major, src/lapack/interface.hpp:30-43; call sites
 *              cholinv/policy.h:199,265,353,462, cacqr.hpp:20).  Blocked right-looking.
 *   uplo==Upper: A = U^T U, U overwrites the upper triangle; strictly lower untouched.
 *   Returns 0 or the (1-based) order of the first non-positive leading minor.
 * ------------------------------------------------------------------------------------------ */
static int potf2_upper(int64_t n, double* A, int64_t lda) {
  for (int64_t j = 0; j < n; ++j) {
    double ajj = A[j + j * lda];
    for (int64_t p = 0; p < j; ++p) ajj -= A[p + j * lda] * A[p + j * lda];
    if (!(ajj > 0.0)) { A[j + j * lda] = ajj; return (int)(j + 1); }
    ajj = sqrt(ajj);
    A[j + j * lda] = ajj;
    for (int64_t c = j + 1; c < n; ++c) {
      double s = A[j + c * lda];
      for (int64_t p = 0; p < j; ++p) s -= A[p + j * lda] * A[p + c * lda];
      A[j + c * lda] = s / ajj;
    }
  }
  return 0;
}

int orc_dpotrf(int uplo, int64_t n, double* A, int64_t lda) {
  if (n <= 0) return 0;
  if (g_hb.on) return (g_hb.ilp64 ? hb_potrf_int64_t : hb_potrf_int)(uplo, n, A, lda);
  if (uplo == ORC_LOWER) {
    /* not on the hot path (cholinv asserts dir=='U', cholinv.hpp:9): transpose, factor, transpose back */
    double* W = (double*)malloc(sizeof(double) * n * n);
    for (int64_t j = 0; j < n; ++j) for (int64_t i = j; i < n; ++i) W[j + i * n] = A[i + j * lda];
    int info = orc_dpotrf(ORC_UPPER, n, W, n);
    for (int64_t j = 0; j < n; ++j) for (int64_t i = j; i < n; ++i) A[i + j * lda] = W[j + i * n];
    free(W);
    return info;
  }
  const int64_t PB = 96;
  for (int64_t j0 = 0; j0 < n; j0 += PB) {
    int64_t jb = MIN(PB, n - j0), rest = n - j0 - jb;
    int info = potf2_upper(jb, A + j0 + j0 * lda, lda);
    if (info) return (int)(j0 + info);
    if (rest > 0) {
      double* A12 = A + j0 + (j0 + jb) * lda;
      orc_dtrsm(ORC_LEFT, ORC_UPPER, ORC_TRANS, ORC_NONUNIT, jb, rest, 1.0, A + j0 + j0 * lda, lda, A12, lda);
      orc_dsyrk(ORC_UPPER, ORC_TRANS, rest, jb, -1.0, A12, lda, 1.0, A + (j0 + jb) + (j0 + jb) * lda, lda);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * K9  dtrtri  (LAPACKE_dtrtri col-major, src/lapack/interface.hpp:45-58; call sites
 *              cholinv/policy.h:201,267,367,476, cacqr.hpp:22).  In-place triangular inverse;
 *   the other triangle is untouched.  Returns 0 or index of the first zero diagonal entry.
 * ------------------------------------------------------------------------------------------ */
static void trti2_upper(int diag, int64_t n, double* A, int64_t lda) {
  for (int64_t j = 0; j < n; ++j) {
    double ajj;
    if (diag == ORC_NONUNIT) { A[j + j * lda] = 1.0 / A[j + j * lda]; ajj = -A[j + j * lda]; }
    else ajj = -1.0;
    /* x = inv(A[0:j,0:j]) (already inverted, upper) * A[0:j, j] */
    for (int64_t i = 0; i < j; ++i) {
      double s = (diag == ORC_NONUNIT ? A[i + i * lda] : 1.0) * A[i + j * lda];
      for (int64_t p = i + 1; p < j; ++p) s += A[i + p * lda] * A[p + j * lda];
      A[i + j * lda] = s;
    }
    for (int64_t i = 0; i < j; ++i) A[i + j * lda] *= ajj;
  }
}

int orc_dtrtri(int uplo, int diag, int64_t n, double* A, int64_t lda) {
  if (n <= 0) return 0;
  if (g_hb.on) return (g_hb.ilp64 ? hb_trtri_int64_t : hb_trtri_int)(uplo, diag, n, A, lda);
  if (diag == ORC_NONUNIT)
    for (int64_t i = 0; i < n; ++i) if (A[i + i * lda] == 0.0) return (int)(i + 1);
  if (uplo == ORC_LOWER) {
    double* W = (double*)malloc(sizeof(double) * n * n);
    for (int64_t j = 0; j < n; ++j) for (int64_t i = j; i < n; ++i) W[j + i * n] = A[i + j * lda];
    int info = orc_dtrtri(ORC_UPPER, diag, n, W, n);
    for (int64_t j = 0; j < n; ++j) for (int64_t i = j; i < n; ++i) A[i + j * lda] = W[j + i * n];
    free(W);
    return info;
  }
  const int64_t IB = 96;
  for (int64_t j0 = 0; j0 < n; j0 += IB) {
    int64_t jb = MIN(IB, n - j0);
    /* A[0:j0, j0:j0+jb] <- -inv(A00) * A01 * inv(A11) */
    if (j0 > 0) {
      double* A01 = A + j0 * lda;
      orc_dtrmm(ORC_LEFT, ORC_UPPER, ORC_NOTRANS, diag, j0, jb, 1.0, A, lda, A01, lda);
      orc_dtrsm(ORC_RIGHT, ORC_UPPER, ORC_NOTRANS, diag, j0, jb, -1.0, A + j0 + j0 * lda, lda, A01, lda);
    }
    trti2_upper(diag, jb, A + j0 + j0 * lda, lda);
  }
  return 0;
}

/* Householder QR, unblocked LAPACK reference algorithms (dgeqr2 / dorg2r with dlarfg, dlarf): what LAPACKE_dgeqrf and
 * LAPACKE_dorgqr (src/lapack/interface.hpp:60-88) compute up to blocking.  dlarfg's rescaling loop for subnormal
 * norms is omitted, as in the HIP path. */
int orc_dgeqrf(int64_t m, int64_t n, double* A, int64_t lda, double* tau) {
  const int64_t k = MIN(m, n);
  for (int64_t j = 0; j < k; ++j) {
    double* a = A + j + j * lda;
    double ss = 0.0;
    for (int64_t r = 1; r < m - j; ++r) ss += a[r] * a[r];
    if (ss == 0.0) { tau[j] = 0.0; continue; }
    const double alpha = a[0], nrm = sqrt(alpha * alpha + ss), beta = alpha >= 0.0 ? -nrm : nrm;
    const double scal = 1.0 / (alpha - beta);
    for (int64_t r = 1; r < m - j; ++r) a[r] *= scal;
    tau[j] = (beta - alpha) / beta;
    a[0] = beta;
    for (int64_t c = j + 1; c < n; ++c) {                 /* apply H_j to the columns on the right */
      double* b = A + j + c * lda;
      double w = b[0];
      for (int64_t r = 1; r < m - j; ++r) w += a[r] * b[r];
      w *= tau[j];
      b[0] -= w;
      for (int64_t r = 1; r < m - j; ++r) b[r] -= a[r] * w;
    }
  }
  return 0;
}

int orc_dorgqr(int64_t m, int64_t n, int64_t k, double* A, int64_t lda, const double* tau) {
  if (n > m || k > n) return -1;
  double* Q = (double*)calloc((size_t)m * (size_t)n, sizeof(double));
  for (int64_t c = 0; c < n; ++c) Q[c + c * m] = 1.0;
  for (int64_t j = k - 1; j >= 0; --j) {                  /* Q <- H_j Q, v_j = [1; A[j+1:m, j]] */
    const double* a = A + j + j * lda;
    for (int64_t c = 0; c < n; ++c) {
      double* q = Q + j + c * m;
      double w = q[0];
      for (int64_t r = 1; r < m - j; ++r) w += a[r] * q[r];
      w *= tau[j];
      q[0] -= w;
      for (int64_t r = 1; r < m - j; ++r) q[r] -= a[r] * w;
    }
  }
  for (int64_t c = 0; c < n; ++c) memcpy(A + c * lda, Q + c * m, sizeof(double) * (size_t)m);
  free(Q);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Generators  (src/matrix/structure.hpp:36-129).  Restated loop-for-loop, including the
 * padding rule (a trailing local row/column of zeros when the grid does not divide the
 * dimension) and the per-element reseeding of distribute_symmetric.
 * ------------------------------------------------------------------------------------------ */
static void pad_lens(int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY, int64_t px, int64_t py,
                     int64_t PX, int64_t PY, int64_t* padX, int64_t* padY) {
  *padX = (((gdimX % PX != 0) && ((dimX - 1) * PX + px >= gdimX)) ? dimX - 1 : dimX);
  *padY = (((gdimY % PY != 0) && ((dimY - 1) * PY + py >= gdimY)) ? dimY - 1 : dimY);
}

/* structure.hpp:68-103 */
void orc_distribute_symmetric(double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                              int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key, int dd) {
  srand48(key);
  int64_t padX, padY;
  pad_lens(dimX, dimY, gdimX, gdimY, px, py, PX, PY, &padX, &padY);
  int64_t gx = px;
  for (int64_t i = 0; i < padX; ++i) {
    int64_t gy = py;
    for (int64_t j = 0; j < padY; ++j) {
      if (gx > gy) srand48(gx + gdimY * gy);
      else srand48(gy + gdimY * gx);
      data[i * dimY + j] = drand48();
      if (dd && gx == gy && i == j) data[i * dimY + j] += (double)gdimX;
      gy += PY;
    }
    if (padY != dimY) data[i * dimY + dimY - 1] = 0;
    gx += PX;
  }
  if (padX != dimX) for (int64_t j = 0; j < dimY; ++j) data[dimY * (dimX - 1) + j] = 0;
}

/* structure.hpp:105-129 */
void orc_distribute_random(double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                           int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key) {
  srand48(key);
  int64_t padX, padY;
  pad_lens(dimX, dimY, gdimX, gdimY, px, py, PX, PY, &padX, &padY);
  for (int64_t i = 0; i < padX; ++i) {
    for (int64_t j = 0; j < padY; ++j) data[i * dimY + j] = drand48();
    if (padY != dimY) data[i * dimY + dimY - 1] = 0;
  }
  if (padX != dimX) for (int64_t j = 0; j < dimY; ++j) data[dimY * (dimX - 1) + j] = 0;
}

/* structure.hpp:36-66 */
void orc_distribute_identity(double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                             int64_t px, int64_t py, int64_t PX, int64_t PY, double val) {
  int64_t padX, padY;
  pad_lens(dimX, dimY, gdimX, gdimY, px, py, PX, PY, &padX, &padY);
  int64_t gx = px;
  for (int64_t i = 0; i < padX; ++i) {
    int64_t gy = py;
    for (int64_t j = 0; j < padY; ++j) {
      data[i * dimY + j] = 0;
      if (gx == gy && i == j) data[i * dimY + j] += val;
      gy += PY;
    }
    if (padY != dimY) data[i * dimY + dimY - 1] = 0;
    gx += PX;
  }
  if (padX != dimX) for (int64_t j = 0; j < dimY; ++j) data[dimY * (dimX - 1) + j] = 0;
}

double orc_drand48_after_seed(int64_t seed) { srand48(seed); return drand48(); }
void orc_drand48_stream(int64_t seed, int64_t count, double* out) {
  srand48(seed);
  for (int64_t i = 0; i < count; ++i) out[i] = drand48();
}

/* ------------------------------------------------------------------------------------------
 * Layout offsets (src/matrix/structure.h:13,39,59) and serialize (src/matrix/serialize.hpp:12-150):
 * column i of the range copies rangeY (rect), i+1 (upper-tri source or dest) or rangeY-i
 * (lower-tri) elements, starting at (sx+i, sy) resp. (sx+i, sy+i).
 * ------------------------------------------------------------------------------------------ */
int64_t orc_offset(int st, int64_t x, int64_t y, int64_t dimX, int64_t dimY) {
  (void)dimX;
  if (st == 0) return x * dimY + y;
  if (st == 1) return ((x * (x + 1)) >> 1) + y;
  return x * dimY + y - (x * (x + 1) / 2);
}

void orc_serialize(int ss, int ds, const double* src, int64_t sdimX, int64_t sdimY,
                   double* dst, int64_t ddimX, int64_t ddimY,
                   int64_t ssx, int64_t sex, int64_t ssy, int64_t sey,
                   int64_t dsx, int64_t dex, int64_t dsy, int64_t dey) {
  (void)dex; (void)dey;
  int64_t rangeX = sex - ssx, rangeY = sey - ssy;
  int lower = (ss == 2 || ds == 2), upper = (ss == 1 || ds == 1);
  for (int64_t i = 0; i < rangeX; ++i) {
    int64_t so, d_o, cnt;
    if (lower) {
      so = orc_offset(ss, ssx + i, ssy + i, sdimX, sdimY); d_o = orc_offset(ds, dsx + i, dsy + i, ddimX, ddimY); cnt = rangeY - i;
    } else {
      so = orc_offset(ss, ssx + i, ssy, sdimX, sdimY); d_o = orc_offset(ds, dsx + i, dsy, ddimX, ddimY); cnt = upper ? i + 1 : rangeY;
    }
    memcpy(dst + d_o, src + so, sizeof(double) * cnt);
  }
}

/* src/util/util.hpp:105-128: blocked (d*d rank pieces, each rows_local x cols_local col-major,
 * piece index = x + y... as gathered over `slice`: rank r = column-rank z*d + j in the loop) ->
 * element-cyclic aggregate matrix; then zero the strictly lower part. */
void orc_block_to_cyclic_rect(const double* blocked, double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  int64_t w = 0, off = rl * cl, rg = rl * d, cg = cl * d;
  for (int64_t i = 0; i < cl; ++i)
    for (int64_t j = 0; j < d; ++j)
      for (int64_t k = 0; k < rl; ++k)
        for (int64_t z = 0; z < d; ++z) cyclic[w++] = blocked[z * off * d + k + j * off + i * rl];
  for (int64_t i = 0; i < cg; ++i) for (int64_t j = i + 1; j < rg; ++j) cyclic[i * rg + j] = 0.;
}
/* src/util/util.hpp:203-217 */
void orc_cyclic_to_block_rect(double* blocked, const double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  int64_t r = 0, off = rl * cl;
  for (int64_t i = 0; i < cl; ++i)
    for (int64_t j = 0; j < d; ++j)
      for (int64_t k = 0; k < rl; ++k)
        for (int64_t z = 0; z < d; ++z) blocked[z * off * d + k + j * off + i * rl] = cyclic[r++];
}
/* src/util/util.hpp:56-102 (packed upper-triangular pieces) */
void orc_block_to_cyclic_triangle(const double* blocked, double* cyclic, int64_t num_elems, int64_t rl, int64_t cl, int64_t d) {
  int64_t rg = rl * d, cg = cl * d;
  int64_t offset = num_elems / (d * d), off1 = 0, off3 = d * offset;
  for (int64_t i = 0; i < cl; ++i) {
    off1 += i;
    for (int64_t j = 0; j < d; ++j) {
      int64_t off2 = j * offset + off1;
      int64_t w = ((i * d) + j) * rg;
      for (int64_t k = 0; k < i; ++k)
        for (int64_t z = 0; z < d; ++z) cyclic[w++] = blocked[off2 + z * off3 + k];
      for (int64_t z = 0; z <= j; ++z) cyclic[w++] = blocked[off2 + z * off3 + i];
    }
  }
  for (int64_t i = 0; i < cg; ++i) for (int64_t j = i + 1; j < rg; ++j) cyclic[i * rg + j] = 0.;
}
/* util::cyclic_to_block_triangle (util.hpp:167-201): the inverse walk -- every packed entry of every piece is written.
 * Entries on a piece's local diagonal that lie below the aggregate's diagonal (x < y) receive 0; the reference skips them
 * (they keep whatever the buffer held), and it zeroes the SOURCE as it reads, which is not restated: the source is const here. */
void orc_cyclic_to_block_triangle(double* blocked, const double* cyclic, int64_t num_elems, int64_t rl, int64_t cl, int64_t d) {
  int64_t rg = rl * d, offset = num_elems / (d * d), off3 = d * offset;
  int64_t off1 = 0;
  for (int64_t i = 0; i < cl; ++i) {
    off1 += i;
    for (int64_t j = 0; j < d; ++j) {
      int64_t off2 = j * offset + off1;
      int64_t read_idx = (i * d + j) * rg;
      for (int64_t k = 0; k < i; ++k)
        for (int64_t z = 0; z < d; ++z) blocked[off2 + z * off3 + k] = cyclic[read_idx++];
      for (int64_t z = 0; z < d; ++z) {
        blocked[off2 + z * off3 + i] = z <= j ? cyclic[read_idx] : 0.0;
        ++read_idx;
      }
    }
  }
}

/* src/util/util.hpp:131-164: pick this rank's element-cyclic piece out of the aggregate factor
 * (in place, into the leading local_dim x local_dim corner with ld = bc_dim), zeroing below the
 * GLOBAL diagonal. */
void orc_cyclic_to_local(double* T, double* TI, int64_t ld_, int64_t bc, int64_t d, int64_t sr) {
  int64_t ro = sr / d, co = sr % d;
  for (int64_t i = 0; i < ld_; ++i)
    for (int64_t j = 0; j < ld_; ++j) {
      int64_t rc = i * d + co, rr = j * d + ro, w = i * bc + j;
      if (rc >= rr) { T[w] = T[rc * bc + rr]; TI[w] = TI[rc * bc + rr]; }
      else { T[w] = 0.; TI[w] = 0.; }
    }
}

/* element-cyclic ownership: local (col i,row j) <-> global (px+i*PX, py+j*PY)  (matrix.hpp:8-11) */
void orc_cyclic_extract(const double* G, int64_t gcols, int64_t grows, int64_t ldg,
                        double* L, int64_t px, int64_t py, int64_t PX, int64_t PY) {
  int64_t lc = gcols / PX + (gcols % PX ? 1 : 0), lr = grows / PY + (grows % PY ? 1 : 0);
  for (int64_t i = 0; i < lc; ++i)
    for (int64_t j = 0; j < lr; ++j) {
      int64_t gx = px + i * PX, gy = py + j * PY;
      L[i * lr + j] = (gx < gcols && gy < grows) ? G[gx * ldg + gy] : 0.0;
    }
}
void orc_cyclic_insert(double* G, int64_t gcols, int64_t grows, int64_t ldg,
                       const double* L, int64_t px, int64_t py, int64_t PX, int64_t PY) {
  int64_t lc = gcols / PX + (gcols % PX ? 1 : 0), lr = grows / PY + (grows % PY ? 1 : 0);
  for (int64_t i = 0; i < lc; ++i)
    for (int64_t j = 0; j < lr; ++j) {
      int64_t gx = px + i * PX, gy = py + j * PY;
      if (gx < gcols && gy < grows) G[gx * ldg + gy] = L[i * lr + j];
    }
}

/* ------------------------------------------------------------------------------------------
 * cholinv  (src/alg/cholesky/cholinv/cholinv.hpp:6-183 + policy.h:160-224)
 *
 * The recursion is restated on ONE address space with the grid collapsed: local index ranges
 * of the reference's element-cyclic blocks correspond to global ranges scaled by d, so the
 * recursion on global indices with (local size)*d is the same tree.  R and Rinv are full n x n
 * (the reference's NoSerialize layout; Serialize only changes storage, serialize.hpp).
 * Steps per level (cholinv.hpp:107-155):
 *   1. recurse on the leading s1 x s1 block                      -> R11, Rinv11
 *   2. R12 = Rinv11^T * A12        (trmm Left/Upper/Trans, :118-121)
 *   3. A22 <- A22 - R12^T R12      (summa "syrk", executed by the reference as GEMM, :132-134)
 *   4. recurse on the trailing s2 x s2 block                      -> R22, Rinv22
 *   5. unless top level && !complete_inv: Rinv12 = -Rinv11 * R12 * Rinv22 (:147-155)
 * base case (policy.h:190-223): potrf on the aggregated block, copy, trtri, zero below diagonal.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int64_t n, true_global, bc_dimension; /* global sizes */
  int complete_inv, split, d;
  double *R, *Rinv;
  int info;
} cholinv_ctx;

int64_t orc_cholinv_bc_dimension(int64_t n_local, int c, int d, int bc_mult_dim) {
  /* cholinv.hpp:15-18 */
  int64_t bc = (int64_t)c * d, mult = bc_mult_dim;
  if (mult < 0) { mult = -mult; for (int i = 0; i < mult; ++i) bc *= 2; }
  else { for (int i = 0; i < mult; ++i) bc /= 2; }
  bc = MAX((int64_t)1, bc);
  bc = MIN(n_local, bc);
  bc = n_local / bc;
  return d * bc;
}

static void cholinv_invoke(cholinv_ctx* cx, int64_t start, int64_t local_dim, int64_t global_dim) {
  int64_t n = cx->n, d = cx->d;
  double *R = cx->R, *Ri = cx->Rinv;
  int64_t split1 = local_dim >> cx->split;
  int64_t g0 = start * d;                 /* global start of this diagonal block */
  if ((local_dim * d <= cx->bc_dimension) || (split1 < cx->split)) {
    /* base case: order = min(local_dim*d, n - g0) (the `span` rule, policy.h:196) */
    int64_t b = MIN(local_dim * d, n - g0);
    double* Rb = R + g0 + g0 * n;
    double* Ib = Ri + g0 + g0 * n;
    int info = orc_dpotrf(ORC_UPPER, b, Rb, n);
    if (info && !cx->info) cx->info = (int)(g0 + info);
    for (int64_t j = 0; j < b; ++j) {
      memcpy(Ib + j * n, Rb + j * n, sizeof(double) * (j + 1));
      for (int64_t i = j + 1; i < b; ++i) { Rb[i + j * n] = 0.0; Ib[i + j * n] = 0.0; } /* cyclic_to_local zeroing */
    }
    orc_dtrtri(ORC_UPPER, ORC_NONUNIT, b, Ib, n);
    return;
  }
  int64_t split2 = local_dim - split1;
  int64_t h1 = split1 * d, g1 = g0 + h1;           /* global sizes/offsets */
  int64_t h2 = MIN(split2 * d, n - g1);
  /* 1 */
  cholinv_invoke(cx, start, split1, global_dim >> 1);
  /* 2: R12 = Rinv11^T * A12 */
  double* R12 = R + g0 + g1 * n;
  orc_dtrmm(ORC_LEFT, ORC_UPPER, ORC_TRANS, ORC_NONUNIT, h1, h2, 1.0, Ri + g0 + g0 * n, n, R12, n);
  /* 3: A22 -= R12^T R12 (upper triangle is all that is ever read again) */
  orc_dsyrk(ORC_UPPER, ORC_TRANS, h2, h1, -1.0, R12, n, 1.0, R + g1 + g1 * n, n);
  /* 4 */
  cholinv_invoke(cx, start + split1, split2, split2 * d);
  /* 5 */
  if (!(!cx->complete_inv && (global_dim == cx->true_global))) {
    double* I12 = Ri + g0 + g1 * n;
    for (int64_t j = 0; j < h2; ++j) memcpy(I12 + j * n, R12 + j * n, sizeof(double) * h1);
    orc_dtrmm(ORC_LEFT, ORC_UPPER, ORC_NOTRANS, ORC_NONUNIT, h1, h2, 1.0, Ri + g0 + g0 * n, n, I12, n);
    orc_dtrmm(ORC_RIGHT, ORC_UPPER, ORC_NOTRANS, ORC_NONUNIT, h1, h2, -1.0, Ri + g1 + g1 * n, n, I12, n);
  }
}

int orc_cholinv_factor(const double* A, int64_t n, int complete_inv, int split, int bc_mult_dim,
                       int c, int d, double* R, double* Rinv) {
  if (split <= 0 || n <= 0 || c <= 0 || d <= 0) return -1;
  /* cholinv.hpp:13: copy the upper triangle of the input into R; Rinv starts at zero */
  for (int64_t j = 0; j < n; ++j) {
    memcpy(R + j * n, A + j * n, sizeof(double) * (j + 1));
    for (int64_t i = j + 1; i < n; ++i) R[i + j * n] = 0.0;
  }
  memset(Rinv, 0, sizeof(double) * n * n);
  int64_t n_local = n / d + (n % d ? 1 : 0);
  cholinv_ctx cx = {n, n, orc_cholinv_bc_dimension(n_local, c, d, bc_mult_dim), complete_inv, split, d, R, Rinv, 0};
  cholinv_invoke(&cx, 0, n_local, n);
  return cx.info;
}

/* ------------------------------------------------------------------------------------------
 * cacqr 1-D  (src/alg/qr/cacqr/cacqr.hpp:7-29 sweep_1d, :174-193 invoke_1d, :219-229 factor)
 * Rank p of P owns global rows p, p+P, ... (matrix.hpp:8-11 with PY=P).  Per sweep:
 *   G_p = Q_p^T Q_p (dsyrk Upper/Trans, :14-15); G = sum_p G_p (Allreduce, policy.h:18-24);
 *   R = chol(G), Rinv = trtri(copy) on every rank (:18-22); Q_p <- Q_p * Rinv (dtrmm, :24-25).
 * Second sweep, then R <- R2 * R1 (dtrmm Right/Upper, :185-187).
 * The simulated Allreduce sums rank Grams in rank order.
 * ------------------------------------------------------------------------------------------ */
static int cacqr_sweep_1d(double* Q, int64_t m, int64_t n, int P, double* G, double* Ginv) {
  memset(G, 0, sizeof(double) * n * n);
  double* Gp = (double*)calloc((size_t)(n * n), sizeof(double));
  int64_t mloc_max = m / P + (m % P ? 1 : 0);
  double* Qp = P == 1 ? NULL : (double*)malloc(sizeof(double) * mloc_max * n);
  int info = 0;
  for (int p = 0; p < P; ++p) {
    int64_t mloc = (m - p + P - 1) / P;
    if (P == 1) {                                     /* one rank: its block IS the matrix, no gather */
      orc_dsyrk(ORC_UPPER, ORC_TRANS, n, m, 1.0, Q, m, 0.0, Gp, n);
    } else {
      for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < mloc; ++i) Qp[i + j * mloc] = Q[(p + i * P) + j * m];
      orc_dsyrk(ORC_UPPER, ORC_TRANS, n, mloc, 1.0, Qp, mloc, 0.0, Gp, n);
    }
    for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i <= j; ++i) G[i + j * n] += Gp[i + j * n];
  }
  info = orc_dpotrf(ORC_UPPER, n, G, n);
  memcpy(Ginv, G, sizeof(double) * n * n);
  orc_dtrtri(ORC_UPPER, ORC_NONUNIT, n, Ginv, n);
  for (int p = 0; p < P; ++p) {
    int64_t mloc = (m - p + P - 1) / P;
    if (P == 1) { orc_dtrmm(ORC_RIGHT, ORC_UPPER, ORC_NOTRANS, ORC_NONUNIT, m, n, 1.0, Ginv, n, Q, m); continue; }
    for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < mloc; ++i) Qp[i + j * mloc] = Q[(p + i * P) + j * m];
    orc_dtrmm(ORC_RIGHT, ORC_UPPER, ORC_NOTRANS, ORC_NONUNIT, mloc, n, 1.0, Ginv, n, Qp, mloc);
    for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < mloc; ++i) Q[(p + i * P) + j * m] = Qp[i + j * mloc];
  }
  free(Gp);
  free(Qp);
  return info;
}

int orc_cacqr_factor_1d(double* Q, int64_t m, int64_t n, int P, int num_iter, double* R) {
  double* Rinv = (double*)malloc(sizeof(double) * n * n);
  int info = cacqr_sweep_1d(Q, m, n, P, R, Rinv);
  if (num_iter > 1) {
    double* R1 = (double*)malloc(sizeof(double) * n * n);
    memcpy(R1, R, sizeof(double) * n * n);
    int info2 = cacqr_sweep_1d(Q, m, n, P, R, Rinv);
    if (!info) info = info2;
    orc_dtrmm(ORC_RIGHT, ORC_UPPER, ORC_NOTRANS, ORC_NONUNIT, n, n, 1.0, R1, n, R, n); /* R = R2*R1 */
    free(R1);
  }
  for (int64_t j = 0; j < n; ++j) for (int64_t i = j + 1; i < n; ++i) R[i + j * n] = 0.0;
  free(Rinv);
  return info;
}

/* ------------------------------------------------------------------------------------------
 * Validators.  util::residual_local (src/util/util.hpp:25-53): sqrt(sum err^2)/sqrt(sum control^2).
 * ------------------------------------------------------------------------------------------ */
/* test/cholesky/validate.hpp:31-46 ('U'): err = (R^T R - A)[gy<=gx], control = A[gy<=gx] */
double orc_cholesky_residual(const double* A, const double* R, int64_t n) {
  double* W = (double*)malloc(sizeof(double) * n * n);
  for (int64_t j = 0; j < n; ++j) memcpy(W + j * n, A + j * n, sizeof(double) * n);
  double* Ru = (double*)malloc(sizeof(double) * n * n);
  for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < n; ++i) Ru[i + j * n] = i <= j ? R[i + j * n] : 0.0; /* util::remove_triangle */
  orc_dgemm(ORC_TRANS, ORC_NOTRANS, n, n, n, 1.0, Ru, n, Ru, n, -1.0, W, n);
  double err = 0, ctl = 0;
  for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i <= j; ++i) { err += W[i + j * n] * W[i + j * n]; ctl += A[i + j * n] * A[i + j * n]; }
  free(W); free(Ru);
  return sqrt(err) / sqrt(ctl);
}
/* test/qr/validate.hpp:34-52: ||Q R - A||_F / ||A||_F */
double orc_qr_residual(const double* A, const double* Q, const double* R, int64_t m, int64_t n) {
  double* W = (double*)malloc(sizeof(double) * m * n);
  memcpy(W, A, sizeof(double) * m * n);
  double* Ru = (double*)malloc(sizeof(double) * n * n);
  for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < n; ++i) Ru[i + j * n] = i <= j ? R[i + j * n] : 0.0;
  orc_dgemm(ORC_NOTRANS, ORC_NOTRANS, m, n, n, 1.0, Q, m, Ru, n, -1.0, W, m);
  double err = 0, ctl = 0;
  for (int64_t t = 0; t < m * n; ++t) { err += W[t] * W[t]; ctl += A[t] * A[t]; }
  free(W); free(Ru);
  return sqrt(err) / sqrt(ctl);
}
/* test/qr/validate.hpp:4-32: err = Q^T Q - I entrywise, control = 1 per entry -> ||.||_F / n */
double orc_qr_orthogonality(const double* Q, int64_t m, int64_t n) {
  double* W = (double*)malloc(sizeof(double) * n * n);
  orc_dgemm(ORC_TRANS, ORC_NOTRANS, n, n, m, 1.0, Q, m, Q, m, 0.0, W, n);
  double err = 0, ctl = 0;
  for (int64_t j = 0; j < n; ++j) for (int64_t i = 0; i < n; ++i) {
    double v = i == j ? fabs(1.0 - W[i + j * n]) : W[i + j * n];
    err += v * v; ctl += 1.0;
  }
  free(W);
  return sqrt(err) / sqrt(ctl);
}
