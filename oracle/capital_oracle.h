/*
 * capital_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the arithmetic and schedules of huttered40/capital's
 * hot path (recursive Cholesky-with-inverse "cholinv", 1-D CA-CholeskyQR2
 * "cacqr", the BLAS/LAPACK calls they bottom out in, the matrix generators and
 * the residual validators).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (libcapital_hip.so) never links, imports or falls back to it.
 *
 * PARITY STATUS: "parity unpinned" against reference *outputs*: the reference
 * ships no golden vectors / known-answer tests and its
 * sources cannot be built in this image (src/util/shared.h:24 includes "mkl.h",
 * which the image lacks; writing a stand-in header is not allowed).  What IS
 * pinned (tests/test_oracle_*.py):
 *   - generators: bit-exact against glibc srand48/drand48, the very functions
 *     the reference calls (src/matrix/structure.hpp:68-129);
 *   - K1..K9 (dgemm/dtrmm/dsyrk/dpotrf/dtrtri): against the LAPACK/BLAS the
 *     image does have (scipy's OpenBLAS, and libmkl_rt.so -- the reference's
 *     own third-party dependency -- when present);
 *   - whole schedules: orc_host_blas_bind() routes the five BLAS/LAPACK routines
 *     below to the cblas_* / LAPACKE_* entry points of a host library at run time;
 *     tests/golden/make_mkl_golden.py runs cholinv and cacqr that way on Intel MKL
 *     (the reference's arithmetic provider) and commits R, R^-1, Q as fixtures
 *     (tests/golden/capital_mkl_schedules.npz) that this file's OWN kernels must
 *     reproduce to 1e-12 (tests/test_golden.py).  The same binding is bench.py's
 *     cpu_baseline (oracle/host_baseline.py);
 *   - schedules: through the reference's own validator metrics
 *     (test/cholesky/validate.hpp, test/qr/validate.hpp) and the values the
 *     survey recorded for the unmodified reference.
 *
 * All matrices are column-major.  Enum codes follow the reference's
 * blas::/lapack:: enums (src/blas/engine.h:23-46, src/lapack/engine.h:23-36).
 */
#ifndef CAPITAL_ORACLE_H_
#define CAPITAL_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* blas::Transpose / Side / UpLo / Diag (src/blas/engine.h:28-46) */
enum { ORC_NOTRANS = 0, ORC_TRANS = 1 };
enum { ORC_LEFT = 0, ORC_RIGHT = 1 };
enum { ORC_LOWER = 0, ORC_UPPER = 1 };
enum { ORC_NONUNIT = 0, ORC_UNIT = 1 };

/* host-BLAS backend: bind cblas_dgemm/dtrmm/dsyrk + LAPACKE_dpotrf/dtrtri of `path` (symbol names get prefix/suffix, e.g.
 * "scipy_" / "64_" for numpy's bundled OpenBLAS; ilp64: 64-bit integer arguments); 0 on success.  Once bound, orc_dgemm ..
 * orc_dtrtri forward to it (column-major, the constants of src/blas/interface.hpp:49-52) until orc_host_blas_enable(0). */
int  orc_host_blas_bind(const char* path, const char* prefix, const char* suffix, int ilp64);
void orc_host_blas_enable(int on);
int  orc_host_blas_active(void);

void orc_set_threads(int nthreads);
int  orc_get_threads(void);

/* K1-K3: cblas_dgemm as called at src/blas/interface.hpp:54 (column-major). */
void orc_dgemm(int transA, int transB, int64_t m, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, const double* B, int64_t ldb,
               double beta, double* C, int64_t ldc);
/* K4-K6: cblas_dtrmm, src/blas/interface.hpp:74.  B <- alpha*op(T)*B or alpha*B*op(T). */
void orc_dtrmm(int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb);
/* K7: cblas_dsyrk, src/blas/interface.hpp:92.  Only the `uplo` triangle of C is touched. */
void orc_dsyrk(int uplo, int trans, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, double beta, double* C, int64_t ldc);
/* K8: LAPACKE_dpotrf, src/lapack/interface.hpp:39.  Returns LAPACK info. */
int  orc_dpotrf(int uplo, int64_t n, double* A, int64_t lda);
/* K9: LAPACKE_dtrtri, src/lapack/interface.hpp:54.  Returns LAPACK info. */
int  orc_dtrtri(int uplo, int diag, int64_t n, double* A, int64_t lda);
/* LAPACKE_dgeqrf / LAPACKE_dorgqr (lapack/interface.hpp:60-88): unblocked dgeqr2 / dorg2r */
int  orc_dgeqrf(int64_t m, int64_t n, double* A, int64_t lda, double* tau);
int  orc_dorgqr(int64_t m, int64_t n, int64_t k, double* A, int64_t lda, const double* tau);
/* Not in the reference (it has no TRSM, SURVEY.md quick facts): checker for the
 * product's extra capi_dtrsm.  B <- alpha*op(T)^-1*B or alpha*B*op(T)^-1. */
void orc_dtrsm(int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb);

/* Generators, src/matrix/structure.hpp:36-129 (rect::_distribute_*).
 * data is dimX columns x dimY rows, column-major, ld = dimY. */
void orc_distribute_symmetric(double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                              int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key, int diag_dominant);
void orc_distribute_random(double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                           int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key);
void orc_distribute_identity(double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                             int64_t px, int64_t py, int64_t PX, int64_t PY, double val);
/* raw POSIX stream helpers (to pin the closed forms the device generators use) */
double orc_drand48_after_seed(int64_t seed);          /* srand48(seed); return drand48(); */
void   orc_drand48_stream(int64_t seed, int64_t count, double* out);

/* Packed-triangular layouts, src/matrix/structure.h:39,59 and serialize.hpp:12-150. */
int64_t orc_offset(int structure /*0 rect,1 uppertri,2 lowertri*/, int64_t x, int64_t y, int64_t dimX, int64_t dimY);
void orc_serialize(int src_struct, int dst_struct, const double* src, int64_t sdimX, int64_t sdimY,
                   double* dst, int64_t ddimX, int64_t ddimY,
                   int64_t ssx, int64_t sex, int64_t ssy, int64_t sey,
                   int64_t dsx, int64_t dex, int64_t dsy, int64_t dey);

/* Base-case re-index helpers, src/util/util.hpp:56-230 (restated index maps). */
void orc_block_to_cyclic_rect(const double* blocked, double* cyclic, int64_t rows_local, int64_t cols_local, int64_t d);
void orc_cyclic_to_block_rect(double* blocked, const double* cyclic, int64_t rows_local, int64_t cols_local, int64_t d);
void orc_block_to_cyclic_triangle(const double* blocked, double* cyclic, int64_t num_elems, int64_t rows_local, int64_t cols_local, int64_t d);
void orc_cyclic_to_block_triangle(double* blocked, const double* cyclic, int64_t num_elems, int64_t rows_local, int64_t cols_local, int64_t d);
void orc_cyclic_to_local(double* T, double* TI, int64_t local_dim, int64_t bc_dim, int64_t d, int64_t slice_rank);
/* element-cyclic ownership (src/matrix/matrix.hpp:8-11): extract / scatter one rank's block */
void orc_cyclic_extract(const double* global, int64_t gcols, int64_t grows, int64_t ldg,
                        double* local, int64_t px, int64_t py, int64_t PX, int64_t PY);
void orc_cyclic_insert(double* global, int64_t gcols, int64_t grows, int64_t ldg,
                       const double* local, int64_t px, int64_t py, int64_t PX, int64_t PY);

/* cholesky::cholinv<...>::factor, src/alg/cholesky/cholinv/cholinv.hpp:6-183, on a
 * c x d x d grid collapsed to one address space (the factor R of an SPD matrix is
 * unique, so this is the elementwise reference for any grid; c,d only enter the
 * base-case size rule cholinv.hpp:15-18).  A, R, Rinv: n x n full column-major;
 * R/Rinv strictly-lower parts are returned as zeros.  Returns 0, or potrf info. */
int orc_cholinv_factor(const double* A, int64_t n, int complete_inv, int split, int bc_mult_dim,
                       int c, int d, double* R, double* Rinv);
/* number of base cases / recursion depth the rule above yields (host-logic check) */
int64_t orc_cholinv_bc_dimension(int64_t n_local, int c, int d, int bc_mult_dim);

/* qr::cacqr<...>::factor with c==1 (invoke_1d / sweep_1d), src/alg/qr/cacqr/cacqr.hpp:7-29,174-193.
 * P simulated ranks own row-cyclic blocks of A (m x n, ld m); A is overwritten by Q
 * (global, same layout), R is n x n upper (strictly lower zero).  num_iter = 1|2. */
int orc_cacqr_factor_1d(double* A_to_Q, int64_t m, int64_t n, int P, int num_iter, double* R);

/* Validators, test/cholesky/validate.hpp:7-49, test/qr/validate.hpp:4-52,
 * util::residual_local src/util/util.hpp:25-53. */
double orc_cholesky_residual(const double* A, const double* R, int64_t n);
double orc_qr_residual(const double* A, const double* Q, const double* R, int64_t m, int64_t n);
double orc_qr_orthogonality(const double* Q, int64_t m, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
