// qr_f64.hip -- Householder QR for gfx950: capi_dgeqrf / capi_dorgqr behind lapack::engine::_geqrf / _orgqr
// (reference src/lapack/interface.hpp:60-88, LAPACKE_dgeqrf / LAPACKE_dorgqr; the reference has the engine slots but no
// caller -- CholeskyQR2 is its QR -- so this is the numerically robust fallback of SURVEY.md 8f-2, not a hot path).
//
// Blocked right-looking with compact-WY block reflectors of width 32:
//   panel      dgeqr2 column by column, two launches per column: a one-workgroup kernel turns the column's partial sums
//              (x^T x and x^T A_c, fixed slots added in a fixed order -> bit-reproducible) into the reflector (dlarfg:
//              beta = -sign(alpha) ||x||, tau = (beta-alpha)/beta, v = x / (alpha - beta)) and w = tau v^T A; ONE pass over the
//              rest of the panel then scales x, applies the rank-1 update and accumulates the next column's partial sums;
//   T          from G = V^T V (MFMA tile kernel, split-K over the rows) by the dlarft recurrence in one wave;
//   trailing   A2 -= V (T^T (V^T A2)) as three products on the tile kernel.
// dorgqr applies the block reflectors, last panel first, to [I; 0] in workspace and copies the result over A.
// HBM-bound per column (the rest of the panel is read and written once per column); dlarfg's rescaling loop for subnormal
// norms is omitted.
// Tall panels (m >= 64 n, n <= 2048) do not go column by column: geqrf runs CholeskyQR2 on the MFMA tall-skinny kernels and
// RECONSTRUCTS the LAPACK output (reflectors, R, tau) from its Q and R (geqrf_tall_reconstruct below; Householder panels as the
// fallback when CholeskyQR2 is not safe), dorgqr applies all n reflectors as ONE block reflector (orgqr_tall_one_block).
#include <stdlib.h>
#include <vector>
#include "capi_internal.h"

namespace {

__global__ void set_info_kernel(int* info, int v) { *info = v; }

constexpr int QNB = 32;      // block reflector width
constexpr int QPART = 1024;  // partial-sum slots of the column reductions (= workgroups of a panel pass: 4 per CU)

__device__ __forceinline__ double block_sum(double v, double* red) {   // sum over a 256-thread workgroup, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// part[b] = sum of x_r^2 over this workgroup's rows of x (len entries)
__global__ void col_sumsq_kernel(const double* __restrict__ x, int64_t len, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < len; r += (int64_t)gridDim.x * 256) s += x[r] * x[r];
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// dlarfg on column j: a = &A[j][j], x = a + 1 (len entries below the diagonal)
__global__ void make_reflector_kernel(double* __restrict__ a, int64_t len, const double* __restrict__ part, int npart,
                                      double* __restrict__ tau) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < npart; i += 256) s += part[i];
  const double xnorm2 = block_sum(s, red);
  const double alpha = a[0];
  if (xnorm2 == 0.0) {                                       // H = I
    if (blockIdx.x == 0 && threadIdx.x == 0) *tau = 0.0;
    return;
  }
  const double nrm = sqrt(alpha * alpha + xnorm2);
  const double beta = alpha >= 0.0 ? -nrm : nrm;
  const double scal = 1.0 / (alpha - beta);
  __syncthreads();                                           // every thread has read a[0] before it is overwritten
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < len; r += (int64_t)gridDim.x * 256) a[1 + r] *= scal;
  if (blockIdx.x == 0 && threadIdx.x == 0) { *tau = (beta - alpha) / beta; }
}
// (the diagonal entry is set to beta by a separate one-thread kernel AFTER every workgroup has read alpha)
__global__ void set_beta_kernel(double* __restrict__ a, const double* __restrict__ part, int npart) {
  double s = 0.0;
  for (int i = 0; i < npart; ++i) s += part[i];
  if (s == 0.0) return;
  const double alpha = a[0], nrm = sqrt(alpha * alpha + s);
  a[0] = alpha >= 0.0 ? -nrm : nrm;
}

// part2[b][c] = sum over this workgroup's rows of v_r * A[r][c+1], c = 0..nc-1; v_0 = 1, v_r = a[r] (column j below the diagonal)
__global__ void panel_dot_kernel(const double* __restrict__ a, int64_t lda, int64_t rows, int nc, double* __restrict__ part2) {
  __shared__ double red[4];
  double acc[QNB - 1];
#pragma unroll
  for (int c = 0; c < QNB - 1; ++c) acc[c] = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const double v = r == 0 ? 1.0 : a[r];
#pragma unroll
    for (int c = 0; c < QNB - 1; ++c)
      if (c < nc) acc[c] += v * a[r + (int64_t)(c + 1) * lda];
  }
#pragma unroll
  for (int c = 0; c < QNB - 1; ++c) {
    if (c < nc) {
      const double s = block_sum(acc[c], red);
      if (threadIdx.x == 0) part2[(int64_t)blockIdx.x * QNB + c] = s;
    }
  }
}

// w[c] = tau * sum_b part2[b][c]
__global__ void reduce_w_kernel(const double* __restrict__ part2, int npart, int nc, const double* __restrict__ tau, double* __restrict__ w) {
  const int c = threadIdx.x;
  if (c >= nc) return;
  double s = 0.0;
  for (int b = 0; b < npart; ++b) s += part2[(int64_t)b * QNB + c];
  w[c] = (*tau) * s;
}

// A[r][c+1] -= v_r * w[c]
__global__ void panel_apply_kernel(double* __restrict__ a, int64_t lda, int64_t rows, int nc, const double* __restrict__ w) {
  double wl[QNB - 1];
#pragma unroll
  for (int c = 0; c < QNB - 1; ++c) wl[c] = c < nc ? w[c] : 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const double v = r == 0 ? 1.0 : a[r];
#pragma unroll
    for (int c = 0; c < QNB - 1; ++c)
      if (c < nc) a[r + (int64_t)(c + 1) * lda] -= v * wl[c];
  }
}

// ---- fused panel: ONE pass over the rest of the panel per column ---------------------------------------------------
// State between columns: part[b][0] = partial sum of x^2, part[b][1 + t] = partial sum of x * A[:, t-th column right of x],
// for the UNSCALED x = A[j+1:, j] of the column about to be reduced.  reflector_w_kernel turns them into beta, tau, the
// scale of x and w = tau v^T A (v = [1; x * scal]: v^T A_c = A[j][c] + scal * x^T A_c[j+1:]); panel_pass_kernel scales x,
// applies the rank-1 update to the remaining columns and, on the updated values, accumulates the same partial sums for the
// next column.  Partial sums live in fixed slots and are added in a fixed order: bit-reproducible.
template <int DUMMY = 0>
__device__ __forceinline__ void store_block_sums(const double (&acc)[QNB], int count, double* __restrict__ out, double* red) {
#pragma unroll
  for (int c = 0; c < QNB; ++c) {
    if (c < count) {
      const double t = block_sum(acc[c], red);
      if (threadIdx.x == 0) out[c] = t;
    }
  }
}

// a = &A[j][j]; nc columns right of it; sw[0] = scal, sw[1 + c] = w[c]; tau_j = &tau[j]
__global__ void reflector_w_kernel(double* __restrict__ a, int64_t lda, int64_t len, int nc, const double* __restrict__ part, int npart,
                                   double* __restrict__ tau_j, double* __restrict__ sw) {
  // 256 threads = 32 values x 8 slot-lanes: lane g sums slots g, g + 8, ... (loads independent of each other), the eight
  // partial sums are added in a fixed order
  __shared__ double ps[8][QNB];
  __shared__ double sums[QNB];
  __shared__ double sc[2];                                   // scal, tau
  const int t = threadIdx.x & (QNB - 1), g = threadIdx.x / QNB;
  {
    double s = 0.0;
    if (t <= nc && len > 0)
      for (int b = g; b < npart; b += 8) s += part[(int64_t)b * QNB + t];
    ps[g][t] = s;
  }
  __syncthreads();
  if (g == 0) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += ps[q][t];
    sums[t] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double xnorm2 = sums[0], alpha = a[0];
    double scal = 0.0, tv = 0.0;
    if (xnorm2 != 0.0) {
      const double nrm = sqrt(alpha * alpha + xnorm2);
      const double beta = alpha >= 0.0 ? -nrm : nrm;
      scal = 1.0 / (alpha - beta);
      tv = (beta - alpha) / beta;
      a[0] = beta;
    }
    *tau_j = tv;
    sc[0] = scal; sc[1] = tv;
    sw[0] = scal;
  }
  __syncthreads();
  if (g == 0 && t >= 1 && t <= nc) sw[t] = sc[1] * (a[(int64_t)t * lda] + sc[0] * sums[t]);   // w[t-1] = tau (A[j][c] + scal x^T A_c)
}

// APPLY: scale x (rows 1.. of column 0 of `a`), update columns 1..nc, accumulate the next column's partial sums (its x starts
// two rows below a's first row).  !APPLY: only accumulate, for column 0 itself (first column of a panel; x starts at row 1).
template <bool APPLY>
__global__ void panel_pass_kernel(double* __restrict__ a, int64_t lda, int64_t rows, int nc, const double* __restrict__ sw,
                                  double* __restrict__ part) {
  __shared__ double red[4];
  double acc[QNB];
#pragma unroll
  for (int c = 0; c < QNB; ++c) acc[c] = 0.0;
  double wl[QNB - 1];
  double scal = 0.0;
  if (APPLY) {
    scal = sw[0];
#pragma unroll
    for (int c = 0; c < QNB - 1; ++c) wl[c] = c < nc ? sw[1 + c] : 0.0;
  }
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    if (APPLY) {
      double v = 1.0;
      if (r > 0) { v = a[r] * scal; a[r] = v; }
      double x = 0.0;
#pragma unroll
      for (int c = 0; c < QNB - 1; ++c) {
        if (c < nc) {
          const double val = a[r + (int64_t)(c + 1) * lda] - v * wl[c];
          a[r + (int64_t)(c + 1) * lda] = val;
          if (c == 0) { x = r >= 2 ? val : 0.0; acc[0] += x * x; }
          else acc[c] += x * val;
        }
      }
    } else {
      const double x = r >= 1 ? a[r] : 0.0;
      acc[0] += x * x;
#pragma unroll
      for (int c = 0; c < QNB - 1; ++c)
        if (c < nc) acc[c + 1] += x * a[r + (int64_t)(c + 1) * lda];
    }
  }
  // APPLY: the next column has nc - 1 columns to its right -> nc values (norm + nc - 1 dots); !APPLY: nc + 1 values
  store_block_sums(acc, APPLY ? nc : nc + 1, part + (int64_t)blockIdx.x * QNB, red);
}

// Vw (rows x nb, ld rows) <- the panel's reflectors: unit diagonal, zeros above, A's entries below
__global__ void form_v_kernel(const double* __restrict__ a, int64_t lda, int64_t rows, int nb, double* __restrict__ vw) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  for (int c = blockIdx.y; c < nb; c += gridDim.y) vw[r + (int64_t)c * rows] = r < c ? 0.0 : (r == c ? 1.0 : a[r + (int64_t)c * lda]);
}

// dlarft (forward, columnwise): T upper triangular nb x nb (full storage, zeros below) from G = V^T V and tau
__global__ void larft_kernel(const double* __restrict__ G, const double* __restrict__ tau, int nb, double* __restrict__ T) {
  __shared__ double t[QNB][QNB + 1];
  const int i0 = threadIdx.x;
  for (int c = 0; c < nb; ++c) if (i0 < nb) t[i0][c] = 0.0;
  __syncthreads();
  for (int i = 0; i < nb; ++i) {
    const double ti = tau[i];
    // T(0:i, i) = -tau_i * T(0:i, 0:i) * G(0:i, i)
    double s = 0.0;
    if (i0 < i)
      for (int l = i0; l < i; ++l) s += t[i0][l] * G[l + (int64_t)i * nb];
    __syncthreads();
    if (i0 < i) t[i0][i] = -ti * s;
    if (i0 == i) t[i][i] = ti;
    __syncthreads();
  }
  for (int c = 0; c < nb; ++c) if (i0 < nb) T[i0 + (int64_t)c * nb] = t[i0][c];
}

__global__ void identity_kernel(double* __restrict__ Q, int64_t m, int64_t n) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  for (int64_t c = blockIdx.y; c < n; c += gridDim.y) Q[r + c * m] = r == c ? 1.0 : 0.0;
}

#define RC(x) do { int rc__ = (x); if (rc__ != CAPI_OK) return rc__; } while (0)

int nslots(int64_t rows) {
  int64_t b = cdiv(rows, 1024);
  return (int)(b < 1 ? 1 : (b > QPART ? QPART : b));
}

struct qr_ws {
  double *Vw, *W, *W2, *G, *T, *part, *wv, *Q;
};

int qr_workspace(capi_handle_t h, int64_t m, int64_t n, bool with_q, qr_ws& w) {
  const size_t nV = (size_t)m * QNB, nW = (size_t)QNB * (size_t)n, nQ = with_q ? (size_t)m * (size_t)n : 0;
  const size_t total = nV + 2 * nW + 2 * QNB * QNB + (size_t)QPART * QNB + QNB + nQ + 64;
  void* p;
  RC(capi_ws2_get(h, sizeof(double) * total, &p));
  double* d = (double*)p;
  w.Vw = d; d += nV;
  w.W = d; d += nW;
  w.W2 = d; d += nW;
  w.G = d; d += QNB * QNB;
  w.T = d; d += QNB * QNB;
  w.part = d; d += (size_t)QPART * QNB;
  w.wv = d; d += QNB;
  w.Q = d;
  return CAPI_OK;
}

// Vw, G = Vw^T Vw and T of the panel at (j0, j0) of width nb
int block_reflector(capi_handle_t h, const double* Apanel, int64_t lda, int64_t rows, int nb, const double* tau, qr_ws& w) {
  hipStream_t s = h->stream;
  hipLaunchKernelGGL(form_v_kernel, dim3((unsigned)cdiv(rows, 256), (unsigned)nb), dim3(256), 0, s, Apanel, lda, rows, nb, w.Vw);
  CAPI_HIP_CHECK(h, hipGetLastError());
  RC(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, nb, nb, rows, 1.0, w.Vw, rows, w.Vw, rows, 0.0, w.G, nb));
  hipLaunchKernelGGL(larft_kernel, dim3(1), dim3(64), 0, s, w.G, tau, nb, w.T);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}


// ---- tall panels: CholeskyQR2 + Householder reconstruction ------------------------------------------------------------------
// For m >= 64 n the column-by-column panel above makes n passes over the panel (135 ms at 2^22 x 256) while CholeskyQR2 -- this
// library's hot path -- factors the same matrix in five passes on the MFMA kernels (19 ms).  Its Q and R determine the LAPACK
// output uniquely (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik: "Reconstructing Householder vectors from TSQR", 2014):
// with S = diag(s_i), the thin Householder factor is Q S and
//        Q - [S; 0] = Y U,      Y unit lower trapezoidal (the reflectors), U upper triangular,
// an LU factorisation WITHOUT pivoting in which s_i = -sign of the i-th diagonal entry at the moment it becomes the pivot -- the
// sign dlarfg chooses, and the choice that keeps every pivot >= 1 in magnitude.  Then R_out = S R, tau_i = T_ii = -s_i U_ii
// (T = -U S Y1^-T), and the rows below the top block are Y2 = Q2 U^-1: one more tall product.
// Used only when CholeskyQR2 is safe: both Gram matrices factor (device info) and the first sweep's R has a diagonal ratio below 1e6
// (kappa(A) well inside the u^-1/2 limit); otherwise A is untouched and the Householder panels above run.  n <= 2048.
constexpr int HRNB = 32;

// unblocked LU without pivoting of an (rows x nb) panel whose pivot block starts at P; sgn[c] = the sign subtracted from pivot c
__global__ __launch_bounds__(1024) void hr_lu_panel_kernel(double* __restrict__ P, int64_t ld, int rows, int nb, double* __restrict__ sgn) {
  __shared__ double piv;
  for (int c = 0; c < nb; ++c) {
    if (threadIdx.x == 0) {
      const double d = P[c + (int64_t)c * ld];
      const double sg = d >= 0.0 ? -1.0 : 1.0;             // -sign(d), sign(0) = +1 as in dlarfg (beta = -sign(alpha) norm)
      P[c + (int64_t)c * ld] = d - sg;
      sgn[c] = sg;
      piv = d - sg;
    }
    __syncthreads();
    const double inv = 1.0 / piv;                            // dlarfg scales x by 1 / (alpha - beta)
    for (int r = c + 1 + (int)threadIdx.x; r < rows; r += (int)blockDim.x) {
      const double l = P[r + (int64_t)c * ld] * inv;
      P[r + (int64_t)c * ld] = l;
      for (int cc = c + 1; cc < nb; ++cc) P[r + (int64_t)cc * ld] -= l * P[c + (int64_t)cc * ld];
    }
    __syncthreads();
  }
}

// top n x n block of the output: reflectors (strictly lower part of the LU image W), R_out = S R above, tau_i = -s_i U_ii
__global__ void hr_assemble_kernel(double* __restrict__ A, int64_t lda, const double* __restrict__ W, const double* __restrict__ R,
                                   const double* __restrict__ sgn, int n, double* __restrict__ tau) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y) A[i + (int64_t)j * lda] = i > j ? W[i + (int64_t)j * n] : sgn[i] * R[i + (int64_t)j * n];
  if (blockIdx.y == 0) tau[i] = -sgn[i] * W[i + (int64_t)i * n];
}

__global__ void hr_diag_ratio_kernel(const double* __restrict__ R, int n, double* __restrict__ out2) {   // one workgroup: min and max |r_ii|
  __shared__ double lo[256], hi[256];
  double a = 1e300, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { const double v = fabs(R[i + (int64_t)i * n]); a = v < a ? v : a; b = v > b ? v : b; }
  lo[threadIdx.x] = a; hi[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { lo[threadIdx.x] = fmin(lo[threadIdx.x], lo[threadIdx.x + o]); hi[threadIdx.x] = fmax(hi[threadIdx.x], hi[threadIdx.x + o]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = lo[0]; out2[1] = hi[0]; }
}

// returns CAPI_OK with *done = 1 when A and tau hold the factorisation, *done = 0 when the caller must take the Householder panels
int geqrf_tall_reconstruct(capi_handle_t h, int64_t m, int64_t n, double* A, int64_t lda, double* tau, int* done) {
  *done = 0;
  hipStream_t s = h->stream;
  const size_t nn = (size_t)n * (size_t)n;
  double* buf = nullptr;                                     // the handle's QR block: the BLAS / LAPACK calls below use ws / ws2 / ws3 themselves
  {
    void* pv = nullptr;
    if (capi_ws4_get(h, sizeof(double) * (2 * (size_t)m * (size_t)n + 8 * nn + 2 * (size_t)n + 16), &pv) != CAPI_OK) return CAPI_OK;   // no room: slow path
    buf = (double*)pv;
  }
  double *Q1 = buf, *Q2 = Q1 + (size_t)m * n, *G = Q2 + (size_t)m * n, *Gi = G + nn, *R1 = Gi + nn, *Rf = R1 + nn, *W = Rf + nn, *Ui = W + nn,
         *sgn = Ui + nn + 2 * nn, *stat = sgn + n;
  auto fail = [&](int rc) { return rc; };
  int saved_info = 0;
#define HR(x) do { int rc__ = (x); if (rc__ != CAPI_OK) return fail(rc__); } while (0)
  HR(capi_get_info(h, &saved_info));                         // the handle's LAPACK info word is borrowed for the two Gram factorisations
  HR(capi_reset_info(h));
  // sweep 1 (cacqr.hpp:7-29)
  HR(capi_dsyrk(h, CAPI_UPPER, CAPI_TRANS, n, m, 1.0, A, lda, 0.0, G, n));
  HR(capi_dpotrf_trtri(h, n, G, n, Gi, n));
  HR(capi_memcpy_d2d_async(h, R1, G, sizeof(double) * nn));
  hipLaunchKernelGGL(hr_diag_ratio_kernel, dim3(1), dim3(256), 0, s, G, (int)n, stat);
  HR(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, m, n, 1.0, Gi, n, A, lda, Q1, m));
  // sweep 2 and R = R2 R1 (cacqr.hpp:181-189)
  HR(capi_dsyrk(h, CAPI_UPPER, CAPI_TRANS, n, m, 1.0, Q1, m, 0.0, G, n));
  HR(capi_dpotrf_trtri(h, n, G, n, Gi, n));
  HR(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, m, n, 1.0, Gi, n, Q1, m, Q2, m));
  HR(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n, n, 1.0, R1, n, G, n, Rf, n));
  int info = 0;
  HR(capi_get_info(h, &info));                               // synchronises
  double st[2];
  HR(capi_memcpy_d2h(h, st, stat, sizeof(st)));
  if (saved_info) {                                          // give the caller's pending info back (first failure wins, as on the device)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(set_info_kernel), dim3(1), dim3(1), 0, s, h->d_info, saved_info);
  } else {
    HR(capi_reset_info(h));
  }
  if (info != 0 || !(st[0] > 0.0) || st[1] / st[0] > 1e6) return fail(CAPI_OK);      // CholeskyQR2 is not safe here: Householder panels
  // LU without pivoting of the top block of Q (blocked right-looking: panel, U12 = L11^-1 A12, A22 -= L21 U12)
  HR(capi_dlacpy(h, 0, n, n, Q2, m, W, n));
  for (int64_t j0 = 0; j0 < n; j0 += HRNB) {
    const int nb = (int)(n - j0 < HRNB ? n - j0 : HRNB);
    double* P = W + j0 + j0 * n;
    hipLaunchKernelGGL(hr_lu_panel_kernel, dim3(1), dim3(1024), 0, s, P, n, (int)(n - j0), nb, sgn + j0);
    const int64_t rest = n - j0 - nb;
    if (rest > 0) {
      HR(capi_dtrsm(h, CAPI_LEFT, CAPI_LOWER, CAPI_NOTRANS, CAPI_UNIT, nb, rest, 1.0, P, n, P + (int64_t)nb * n, n));
      HR(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, rest, rest, nb, -1.0, P + nb, n, P + (int64_t)nb * n, n, 1.0, P + nb + (int64_t)nb * n, n));
    }
  }
  // Y2 = Q(n:m, :) U^-1 straight into the output; then the top block
  HR(capi_memset_async(h, Ui, 0, sizeof(double) * nn));
  HR(capi_dlacpy(h, 1, n, n, W, n, Ui, n));
  HR(capi_dtrtri(h, CAPI_UPPER, CAPI_NONUNIT, n, Ui, n));
  if (m > n) HR(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, m - n, n, 1.0, Ui, n, Q2 + n, m, A + n, lda));
  hipLaunchKernelGGL(hr_assemble_kernel, dim3((unsigned)cdiv(n, 256), (unsigned)(n < 65535 ? n : 65535)), dim3(256), 0, s, A, lda, W, Rf, sgn, (int)n, tau);
  if (hipGetLastError() != hipSuccess) return fail(CAPI_EHIP);
#undef HR
  *done = 1;
  return CAPI_OK;
}


// ---- dorgqr on tall panels: ONE block reflector of width n --------------------------------------------------------------------
// Q(:, 0:n) = (I - Y T Y^T)(:, 0:n) = E - Y (T Y1^T) with T^-1 = strictly-upper(Y^T Y) + diag(1 / tau)  (from T^-1 + T^-T = Y^T Y and
// ||v_i||^2 = 2 / tau_i): a tall Gram matrix, an n x n triangular inverse and one tall triangular product -- three passes over
// the panel on the MFMA kernels instead of n / 32 block reflectors applied one after the other.  Needs k == n and every tau_i != 0
// (a reflector with tau = 0 is the identity and has no 1 / tau); otherwise the panel-by-panel path below runs.
__global__ void hr_tinv_kernel(double* __restrict__ G, int n, const double* __restrict__ tau) {   // G (upper, Y^T Y) -> strictly-upper(G) + diag(1/tau), zeros below
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y) {
    double v = G[i + (int64_t)j * n];
    if (i == j) v = 1.0 / tau[i];
    if (i > j) v = 0.0;
    G[i + (int64_t)j * n] = v;
  }
}
__global__ void hr_transpose_top_kernel(const double* __restrict__ Y, int64_t ldy, int n, double* __restrict__ Yt) {   // Yt = Y(0:n, 0:n)^T
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y) Yt[i + (int64_t)j * n] = Y[j + (int64_t)i * ldy];
}
__global__ void hr_add_identity_kernel(double* __restrict__ A, int64_t lda, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) A[i + (int64_t)i * lda] += 1.0;
}

int orgqr_tall_one_block(capi_handle_t h, int64_t m, int64_t n, double* A, int64_t lda, const double* tau, int* done) {
  *done = 0;
  hipStream_t s = h->stream;
  const size_t nn = (size_t)n * (size_t)n;
  {
    std::vector<double> ht((size_t)n);
    int rc = capi_memcpy_d2h(h, ht.data(), tau, sizeof(double) * (size_t)n);
    if (rc != CAPI_OK) return rc;
    for (double t : ht) if (t == 0.0) return CAPI_OK;
  }
  double* buf = nullptr;
  {
    void* pv = nullptr;
    if (capi_ws4_get(h, sizeof(double) * ((size_t)m * (size_t)n + 4 * nn), &pv) != CAPI_OK) return CAPI_OK;
    buf = (double*)pv;
  }
  double *Y = buf, *G = Y + (size_t)m * n, *Yt = G + nn, *M = Yt + nn;
  auto fin = [&](int rc) { return rc; };
#define OQ(x) do { int rc__ = (x); if (rc__ != CAPI_OK) return fin(rc__); } while (0)
  const dim3 gnn((unsigned)cdiv(n, 256), (unsigned)(n < 65535 ? n : 65535));
  hipLaunchKernelGGL(form_v_kernel, dim3((unsigned)cdiv(m, 256), (unsigned)(n < 65535 ? n : 65535)), dim3(256), 0, s, A, lda, m, (int)n, Y);
  OQ(capi_dsyrk(h, CAPI_UPPER, CAPI_TRANS, n, m, 1.0, Y, m, 0.0, G, n));
  hipLaunchKernelGGL(hr_tinv_kernel, gnn, dim3(256), 0, s, G, (int)n, tau);
  OQ(capi_dtrtri(h, CAPI_UPPER, CAPI_NONUNIT, n, G, n));                                              // G = T
  hipLaunchKernelGGL(hr_transpose_top_kernel, gnn, dim3(256), 0, s, Y, m, (int)n, Yt);
  OQ(capi_dtrmm_oop(h, CAPI_LEFT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n, n, 1.0, G, n, Yt, n, M, n));  // M = T Y1^T (upper)
  OQ(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, m, n, -1.0, M, n, Y, m, A, lda));
  hipLaunchKernelGGL(hr_add_identity_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, A, lda, (int)n);
  if (hipGetLastError() != hipSuccess) return fin(CAPI_EHIP);
#undef OQ
  *done = 1;
  return fin(CAPI_OK);
}

}  // namespace

extern "C" {

int capi_dgeqrf(capi_handle_t h, int64_t m, int64_t n, double* A, int64_t lda, double* tau) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, m >= 0 && n >= 0 && m < (1LL << 31) && n < (1LL << 31), "dims");
  if (m == 0 || n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, A && tau && lda >= m, "operands");
  const int64_t k = m < n ? m : n;
  static const bool no_hr = getenv("CAPI_GEQRF_NO_RECONSTRUCT") != nullptr;
  if (!no_hr && n >= 32 && n <= 2048 && m >= 64 * n) {
    int done = 0;
    RC(geqrf_tall_reconstruct(h, m, n, A, lda, tau, &done));
    if (done) return CAPI_OK;
  }
  qr_ws w;
  RC(qr_workspace(h, m, n, false, w));
  hipStream_t s = h->stream;
  for (int64_t j0 = 0; j0 < k; j0 += QNB) {
    const int nb = (int)(k - j0 < QNB ? k - j0 : QNB);
    // panel: columns j0 .. j0+nb-1, one reflector at a time, one pass over the rest of the panel per column
    {
      double* a0 = A + j0 + j0 * lda;
      const int np0 = nslots(m - j0);
      hipLaunchKernelGGL(panel_pass_kernel<false>, dim3(np0), dim3(256), 0, s, a0, lda, m - j0, nb - 1, w.wv, w.part);
      int np_prev = np0;                                      // slots the current column's partial sums occupy
      for (int c = 0; c < nb; ++c) {
        const int64_t j = j0 + c, rows = m - j, len = rows - 1;
        double* a = A + j + j * lda;
        const int nc = nb - 1 - c;                            // columns of the panel right of j
        hipLaunchKernelGGL(reflector_w_kernel, dim3(1), dim3(256), 0, s, a, lda, len, nc, w.part, np_prev, tau + j, w.wv);
        if (len > 0) {
          const int np2 = nslots(rows);
          hipLaunchKernelGGL(panel_pass_kernel<true>, dim3(np2), dim3(256), 0, s, a, lda, rows, nc, w.wv, w.part);
          np_prev = np2;
        }
        CAPI_HIP_CHECK(h, hipGetLastError());
      }
    }
    // trailing block A2 = A[j0:m, j0+nb:n]:  A2 <- (I - V T^T V^T) A2
    const int64_t n2 = n - j0 - nb, rows = m - j0;
    if (n2 > 0) {
      double* Ap = A + j0 + j0 * lda;
      double* A2 = A + j0 + (j0 + nb) * lda;
      RC(block_reflector(h, Ap, lda, rows, nb, tau + j0, w));
      RC(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, nb, n2, rows, 1.0, w.Vw, rows, A2, lda, 0.0, w.W, nb));
      RC(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, nb, n2, nb, 1.0, w.T, nb, w.W, nb, 0.0, w.W2, nb));
      RC(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, rows, n2, nb, -1.0, w.Vw, rows, w.W2, nb, 1.0, A2, lda));
    }
  }
  return CAPI_OK;
}

int capi_dorgqr(capi_handle_t h, int64_t m, int64_t n, int64_t k, double* A, int64_t lda, const double* tau) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, m >= 0 && n >= 0 && n <= m && k >= 0 && k <= n && m < (1LL << 31), "dims (m >= n >= k >= 0)");
  if (m == 0 || n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, A && lda >= m && (tau || k == 0), "operands");
  static const bool no_hr = getenv("CAPI_GEQRF_NO_RECONSTRUCT") != nullptr;
  if (!no_hr && k == n && n >= 32 && n <= 2048 && m >= 64 * n) {
    int done = 0;
    RC(orgqr_tall_one_block(h, m, n, A, lda, tau, &done));
    if (done) return CAPI_OK;
  }
  qr_ws w;
  RC(qr_workspace(h, m, n, true, w));
  hipStream_t s = h->stream;
  hipLaunchKernelGGL(identity_kernel, dim3((unsigned)cdiv(m, 256), (unsigned)(n < 65535 ? n : 65535)), dim3(256), 0, s, w.Q, m, n);
  CAPI_HIP_CHECK(h, hipGetLastError());
  // Q = H_1 ... H_k [I; 0]: block reflectors applied last panel first; panel p only touches Q[j0:m, j0:n]
  const int64_t last = k > 0 ? ((k - 1) / QNB) * QNB : -1;
  for (int64_t j0 = last; j0 >= 0; j0 -= QNB) {
    const int nb = (int)(k - j0 < QNB ? k - j0 : QNB);
    const int64_t rows = m - j0, nq = n - j0;
    RC(block_reflector(h, A + j0 + j0 * lda, lda, rows, nb, tau + j0, w));
    double* Qs = w.Q + j0 + j0 * m;
    RC(capi_dgemm(h, CAPI_TRANS, CAPI_NOTRANS, nb, nq, rows, 1.0, w.Vw, rows, Qs, m, 0.0, w.W, nb));
    RC(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, nb, nq, nb, 1.0, w.T, nb, w.W, nb, 0.0, w.W2, nb));
    RC(capi_dgemm(h, CAPI_NOTRANS, CAPI_NOTRANS, rows, nq, nb, -1.0, w.Vw, rows, w.W2, nb, 1.0, Qs, m));
  }
  return capi_dlacpy(h, 0, m, n, w.Q, m, A, lda);
}

}  // extern "C"
