// movement.hip -- HBM-bound data-movement kernels and the matrix generators.
//
// Replaces, with the data resident in HBM:
//   serialize<S1,S2>::invoke            src/matrix/serialize.hpp:12-150      (capi_serialize)
//   summa's pack/unpack, zero and axpy   src/alg/matmult/summa/summa.hpp:33,135,147-153,216-217
//   util::remove_triangle               src/util/util.hpp:266-291           (capi_remove_triangle)
//   util::block_to_cyclic_rect etc.     src/util/util.hpp:105-128,203-217   (capi_block_to_cyclic, capi_cyclic_to_block)
//   rect::_distribute_*                 src/matrix/structure.hpp:36-129     (capi_distribute_*)
//   util::residual_local's local sums   src/util/util.hpp:25-53             (capi_diff_norms)
// All are one-pass streaming kernels: threads run along the contiguous (row) index of a column.
#include "capi_internal.h"

namespace {

__device__ __host__ inline int64_t st_offset(int st, int64_t x, int64_t y, int64_t dimY) {
  // src/matrix/structure.h:13 (rect), :39 (uppertri), :59 (lowertri)
  if (st == CAPI_RECT) return x * dimY + y;
  if (st == CAPI_UPPERTRI) return ((x * (x + 1)) >> 1) + y;
  return x * dimY + y - (x * (x + 1) / 2);
}

// each thread moves up to four elements of a column, 256 apart, with all its loads in flight before the first store
__global__ void serialize_kernel(int shape, int ss, int ds, const double* __restrict__ src, int64_t sdimY, double* __restrict__ dst,
                                 int64_t ddimY, int64_t ssx, int64_t ssy, int64_t dsx, int64_t dsy, int64_t rangeX,
                                 int64_t rangeY) {
  const int64_t y = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const bool lower = shape == CAPI_LOWERTRI, upper = shape == CAPI_UPPERTRI;
  for (int64_t i = blockIdx.y; i < rangeX; i += gridDim.y) {
    int64_t so, d_o, cnt;
    if (lower) {
      so = st_offset(ss, ssx + i, ssy + i, sdimY);
      d_o = st_offset(ds, dsx + i, dsy + i, ddimY);
      cnt = rangeY - i;
    } else {
      so = st_offset(ss, ssx + i, ssy, sdimY);
      d_o = st_offset(ds, dsx + i, dsy, ddimY);
      cnt = upper ? i + 1 : rangeY;
    }
    if (y >= cnt) continue;                               // (whole workgroups above a triangle's diagonal leave here)
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = src[so + (y + 256 * q < cnt ? y + 256 * q : y)];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (y + 256 * q < cnt) dst[d_o + y + 256 * q] = v[q];
  }
}

__global__ void lacpy_kernel(int part, int64_t m, int64_t n, const double* __restrict__ A, int64_t lda, double* __restrict__ B,
                             int64_t ldb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.y * 8;
  if (i >= m) return;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int64_t j = j0 + q;
    if (j >= n) break;
    if (part == 1 && i > j) continue;
    if (part == 2 && i < j) continue;
    B[i + j * ldb] = A[i + j * lda];
  }
}

// Whole m x n block, B <- A: the product's own copy kernel (round 4) instead of hipMemcpy2DAsync.  Every thread moves the same row pair of
// eight consecutive columns, the eight 16-byte loads in flight before the first store; column groups are walked grid-stride in y.
// Why not the runtime's copy: hipMemcpy2DAsync / hipMemcpyAsync device-to-device are blit kernels that the HIP runtime dispatches on its own
// (4000 of them per bench process, nearly all behind the in-place TRMMs of the TRSM mode).  With its own kernel every dispatch the product
// makes is a plain kernel launch -- one code path through the runtime and through a profiler's queue interception (profiles/README.md: the
// host SIGSEGV of `rocprofv3 --pmc` sits in librocprofiler-sdk's packet interceptor) -- and the small copies lose the blit path's set-up cost.
typedef double d2m_t __attribute__((ext_vector_type(2)));
template <bool VEC>
__global__ __launch_bounds__(256) void copy2d_kernel(int64_t m, int64_t n, const double* __restrict__ A, int64_t lda, double* __restrict__ B, int64_t ldb) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * (VEC ? 2 : 1);
  if (i >= m) return;
  for (int64_t j0 = (int64_t)blockIdx.y * 8; j0 < n; j0 += (int64_t)gridDim.y * 8) {
    if (VEC) {
      d2m_t v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *(const d2m_t*)(A + i + (j0 + q < n ? j0 + q : j0) * lda);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j0 + q < n) *(d2m_t*)(B + i + (j0 + q) * ldb) = v[q];
    } else {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = A[i + (j0 + q < n ? j0 + q : j0) * lda];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j0 + q < n) B[i + (j0 + q) * ldb] = v[q];
    }
  }
}

// Y(part) <- alpha*X + beta*Y on an m x n block (summa.hpp:33,153 generalised to strided blocks)
__global__ void geadd_kernel(int part, int64_t m, int64_t n, double alpha, const double* __restrict__ X, int64_t ldx, double beta,
                             double* __restrict__ Y, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
    if (part == 1 && i > j) continue;
    if (part == 2 && i < j) continue;
    const double x = alpha == 0.0 ? 0.0 : alpha * X[i + j * ldx];
    Y[i + j * ldy] = beta == 0.0 ? x : x + beta * Y[i + j * ldy];
  }
}

__global__ void trizero_kernel(int keep_uplo, int64_t n, double* __restrict__ A, int64_t lda) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int64_t j = blockIdx.y; j < n; j += gridDim.y)
    if (keep_uplo == CAPI_UPPER ? (i > j) : (i < j)) A[i + j * lda] = 0.0;
}

__global__ void axpby_kernel(int64_t count, double beta, const double* __restrict__ x, double* __restrict__ y) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) y[i] = beta * y[i] + x[i];
}

__global__ void remove_triangle_kernel(int dirU, double* __restrict__ A, int64_t dimX, int64_t dimY, int64_t px, int64_t py,
                                       int64_t P) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // local row
  if (j >= dimY) return;
  for (int64_t i = blockIdx.y; i < dimX; i += gridDim.y) {          // local column
    const int64_t gx = px + i * P, gy = py + j * P;
    if (dirU ? (gy > gx) : (gy < gx)) A[i * dimY + j] = 0.0;
  }
}

// cyclic[(i*d + x)*rg + (k*d + y)] = blocked[(y*d + x)*rl*cl + i*rl + k], strictly-lower part zeroed (to_cyclic);
// piece index = x + d*y = rank inside the reference's `slice` communicator (topology.h:85,93-94).
__global__ void block_cyclic_kernel(int to_cyclic, double* __restrict__ blocked, double* __restrict__ cyclic, int64_t rl,
                                    int64_t cl, int64_t d) {
  const int64_t grow = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t rg = rl * d, cg = cl * d;
  if (grow >= rg) return;
  for (int64_t gcol = blockIdx.y; gcol < cg; gcol += gridDim.y) {
    const int64_t i = gcol / d, x = gcol % d, k = grow / d, y = grow % d;
    const int64_t b = (y * d + x) * rl * cl + i * rl + k;
    if (to_cyclic == 2) cyclic[gcol * rg + grow] = blocked[b];                     // an off-diagonal aggregate: nothing is zeroed
    else if (to_cyclic) cyclic[gcol * rg + grow] = (grow > gcol) ? 0.0 : blocked[b];
    else blocked[b] = cyclic[gcol * rg + grow];
  }
}

// Packed-triangle form (util.hpp:57-102, 167-201): the d*d pieces are the ranks' local blocks of an UPPER triangular aggregate,
// each stored packed upper (local column i holds rows 0..i at offset i(i+1)/2: structure.h:39) -- rl(rl+1)/2 doubles per piece
// instead of rl^2, which is what travels in the base-case gather / scatter with the Serialize policy (policy.h:176,322-377).
// Local (column i, row k <= i) of piece (x, y) is global (i d + x, k d + y); on the local diagonal (k == i) only pieces with
// y <= x lie in the aggregate's upper triangle.  to_cyclic: the aggregate's strictly lower part is zeroed, as in the rect form.
// to_blocked: packed entries that lie BELOW the aggregate's diagonal (k == i, y > x) are written as zeros -- the reference's
// cyclic_to_block_triangle leaves them untouched (stale), one reason its Serialize x NoReplication combination returns
// wrong factors.
__global__ void block_cyclic_tri_kernel(int to_cyclic, double* __restrict__ blocked, double* __restrict__ cyclic, int64_t rl, int64_t d) {
  const int64_t grow = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t rg = rl * d, psz = rl * (rl + 1) / 2;
  if (grow >= rg) return;
  for (int64_t gcol = blockIdx.y; gcol < rg; gcol += gridDim.y) {
    const int64_t i = gcol / d, x = gcol % d, k = grow / d, y = grow % d;
    if (k > i) { if (to_cyclic) cyclic[gcol * rg + grow] = 0.0; continue; }      // below every piece's packed triangle
    const int64_t b = (y * d + x) * psz + i * (i + 1) / 2 + k;
    const bool below = grow > gcol;                                            // k == i && y > x
    if (to_cyclic) cyclic[gcol * rg + grow] = below ? 0.0 : blocked[b];
    else blocked[b] = below ? 0.0 : cyclic[gcol * rg + grow];
  }
}

// util::cyclic_to_local (util.hpp:131-164): this rank's element-cyclic piece (slice rank sr: row offset sr / d, column offset sr % d)
// of the aggregated factor T and of its inverse TI, each bc x bc, moved into the leading L x L corner (leading dimension stays bc),
// entries below the GLOBAL diagonal zeroed.  The reference does it in place, front to back; in parallel the piece goes through a
// compact scratch image first (pass 0: gather, pass 1: scatter back).
__global__ void cyclic_to_local_kernel(int pass, double* __restrict__ T, double* __restrict__ TI, double* __restrict__ tmp, int64_t L, int64_t bc,
                                       int64_t d, int64_t ro, int64_t co) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= L) return;
  for (int64_t i = blockIdx.y; i < L; i += gridDim.y) {
    const int64_t rc = i * d + co, rr = j * d + ro;
    if (pass == 0) {
      tmp[i * L + j] = rc >= rr ? T[rc * bc + rr] : 0.0;
      tmp[L * L + i * L + j] = rc >= rr ? TI[rc * bc + rr] : 0.0;
    } else {
      T[i * bc + j] = tmp[i * L + j];
      TI[i * bc + j] = tmp[L * L + i * L + j];
    }
  }
}

// ---- POSIX drand48 in closed form (the reference calls srand48/drand48: structure.hpp:68-129) ----
constexpr uint64_t LCG_A = 0x5DEECE66DULL, LCG_C = 0xBULL, LCG_MASK = (1ULL << 48) - 1;

__device__ __forceinline__ uint64_t seed48(int64_t seed) { return ((((uint64_t)seed) & 0xffffffffULL) << 16) | 0x330EULL; }
__device__ __forceinline__ double to_unit(uint64_t x) { return (double)x * 0x1p-48; }

__global__ void gen_symmetric_kernel(double* __restrict__ data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                                     int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t padX, int64_t padY, int dd) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dimY) return;
  for (int64_t i = blockIdx.y; i < dimX; i += gridDim.y) {
    double v = 0.0;
    if (i < padX && j < padY) {
      const int64_t gx = px + i * PX, gy = py + j * PY;
      const int64_t seed = gx > gy ? gx + gdimY * gy : gy + gdimY * gx;
      const uint64_t x1 = (LCG_A * seed48(seed) + LCG_C) & LCG_MASK;
      v = to_unit(x1);
      if (dd && gx == gy && i == j) v += (double)gdimX;
    }
    data[i * dimY + j] = v;
  }
}

constexpr int GEN_CHUNK = 32;
// element (i,j), i<padX, j<padY is draw number i*padY + j + 1 of the stream seeded with `key`
__global__ void gen_random_kernel(double* __restrict__ data, int64_t dimX, int64_t dimY, int64_t padX, int64_t padY, int64_t key) {
  const int64_t chunk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = padX * padY;
  int64_t t0 = chunk * GEN_CHUNK;
  if (t0 >= total) return;
  // affine jump-ahead: f^n(x) = An*x + Cn, n = t0
  uint64_t An = 1, Cn = 0, Ab = LCG_A, Cb = LCG_C;
  for (uint64_t n = (uint64_t)t0; n; n >>= 1) {
    if (n & 1) { An = (Ab * An) & LCG_MASK; Cn = (Ab * Cn + Cb) & LCG_MASK; }
    Cb = (Ab * Cb + Cb) & LCG_MASK;
    Ab = (Ab * Ab) & LCG_MASK;
  }
  uint64_t x = (An * seed48(key) + Cn) & LCG_MASK;
  const int64_t t1 = (t0 + GEN_CHUNK < total) ? t0 + GEN_CHUNK : total;
  for (int64_t t = t0; t < t1; ++t) {
    x = (LCG_A * x + LCG_C) & LCG_MASK;
    const int64_t i = t / padY, j = t - i * padY;
    data[i * dimY + j] = to_unit(x);
  }
}

__global__ void gen_pad_zero_kernel(double* __restrict__ data, int64_t dimX, int64_t dimY, int64_t padX, int64_t padY) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dimY) return;
  for (int64_t i = blockIdx.y; i < dimX; i += gridDim.y)
    if (i >= padX || j >= padY) data[i * dimY + j] = 0.0;
}

__global__ void gen_identity_kernel(double* __restrict__ data, int64_t dimX, int64_t dimY, int64_t px, int64_t py, int64_t PX,
                                    int64_t PY, int64_t padX, int64_t padY, double val) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dimY) return;
  for (int64_t i = blockIdx.y; i < dimX; i += gridDim.y) {
    double v = 0.0;
    if (i < padX && j < padY && (px + i * PX == py + j * PY) && i == j) v = val;
    data[i * dimY + j] = v;
  }
}

// per-block partial sums of (X-Y)^2 and Y^2 over the selected part; fixed tree order -> reproducible
__global__ void diff_norms_kernel(int part, int64_t m, int64_t n, const double* __restrict__ X, int64_t ldx,
                                  const double* __restrict__ Y, int64_t ldy, double* __restrict__ partial) {
  __shared__ double se[256], sc[256];
  const int64_t j = blockIdx.x;
  double e = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < m; i += 256) {
    if (part == 1 && i > j) break;
    if (part == 2 && i < j) continue;
    const double y = Y[i + j * ldy], dlt = X[i + j * ldx] - y;
    e += dlt * dlt;
    c += y * y;
  }
  se[threadIdx.x] = e;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { se[threadIdx.x] += se[threadIdx.x + s]; sc[threadIdx.x] += sc[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * j] = se[0]; partial[2 * j + 1] = sc[0]; }
}

void pad_lens(int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY, int64_t px, int64_t py, int64_t PX, int64_t PY,
              int64_t* padX, int64_t* padY) {
  // structure.hpp:74-75: at most one trailing local row/column of padding
  *padX = (((gdimX % PX != 0) && ((dimX - 1) * PX + px >= gdimX)) ? dimX - 1 : dimX);
  *padY = (((gdimY % PY != 0) && ((dimY - 1) * PY + py >= gdimY)) ? dimY - 1 : dimY);
}

inline dim3 grid2(int64_t rows, int64_t cols) { return dim3((unsigned)cdiv(rows, 256), (unsigned)(cols < 65535 ? cols : 65535)); }

}  // namespace

extern "C" {

int capi_serialize_shape(capi_handle_t h, int shape, int ss, int ds, const double* src, int64_t sdimX, int64_t sdimY, double* dst,
                         int64_t ddimX, int64_t ddimY, int64_t ssx, int64_t sex, int64_t ssy, int64_t sey, int64_t dsx, int64_t dex,
                         int64_t dsy, int64_t dey) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ss >= 0 && ss <= 2 && ds >= 0 && ds <= 2 && shape >= 0 && shape <= 2, "structure code");
  CAPI_REQUIRE(h, (sex - ssx) == (dex - dsx) && (sey - ssy) == (dey - dsy), "source and destination ranges differ");  // serialize.hpp:19
  (void)sdimX; (void)ddimX;
  const int64_t rangeX = sex - ssx, rangeY = sey - ssy;
  if (rangeX <= 0 || rangeY <= 0) return CAPI_OK;
  CAPI_REQUIRE(h, src && dst, "null matrix");
  hipLaunchKernelGGL(serialize_kernel, dim3((unsigned)cdiv(rangeY, 1024), (unsigned)(rangeX < 65535 ? rangeX : 65535)), dim3(256), 0, h->stream, shape, ss, ds, src, sdimY, dst, ddimY, ssx, ssy,
                     dsx, dsy, rangeX, rangeY);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_serialize(capi_handle_t h, int ss, int ds, const double* src, int64_t sdimX, int64_t sdimY, double* dst, int64_t ddimX,
                   int64_t ddimY, int64_t ssx, int64_t sex, int64_t ssy, int64_t sey, int64_t dsx, int64_t dex, int64_t dsy,
                   int64_t dey) {
  // copy shape follows the structures, as the reference's seven specialisations do (serialize.hpp:12-150)
  const int shape = (ss == CAPI_LOWERTRI || ds == CAPI_LOWERTRI) ? CAPI_LOWERTRI : ((ss == CAPI_UPPERTRI || ds == CAPI_UPPERTRI) ? CAPI_UPPERTRI : CAPI_RECT);
  return capi_serialize_shape(h, shape, ss, ds, src, sdimX, sdimY, dst, ddimX, ddimY, ssx, sex, ssy, sey, dsx, dex, dsy, dey);
}

// B(m x n, ldb) <- A(m x n, lda) on the handle's stream (the blocks may not overlap unless they coincide)
__attribute__((visibility("hidden"))) int capi_internal_copy2d(capi_handle_t h, int64_t m, int64_t n, const double* A, int64_t lda, double* B, int64_t ldb) {
  if (m <= 0 || n <= 0 || (A == B && lda == ldb)) return CAPI_OK;
  static const bool runtime_copy = getenv("CAPI_RUNTIME_COPY") != nullptr;      // A/B: the HIP runtime's blit kernels, as before round 4
  if (runtime_copy) {
    CAPI_HIP_CHECK(h, hipMemcpy2DAsync(B, sizeof(double) * ldb, A, sizeof(double) * lda, sizeof(double) * m, n, hipMemcpyDeviceToDevice, h->stream));
    return CAPI_OK;
  }
  if (lda == m && ldb == m && m * n >= (1 << 16) && (m > 8192 || n == 1)) {
    // contiguous on both sides: re-cut into columns of 8192 (eight 16-byte pieces in flight per thread), the rest as a short column behind them
    const int64_t total = m * n, cols = total / 8192, rest = total - cols * 8192;
    int rc = capi_internal_copy2d(h, 8192, cols, A, 8192, B, 8192);
    if (rc != CAPI_OK || rest == 0) return rc;
    return capi_internal_copy2d(h, rest, 1, A + cols * 8192, rest, B + cols * 8192, rest);
  }
  const bool vec = (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && ((lda | ldb | m) & 1) == 0;
  const int64_t gx = cdiv(vec ? m / 2 : m, 256), gy = cdiv(n, 8);
  CAPI_REQUIRE(h, gx < (1LL << 31), "copy too tall");
  const dim3 grid((unsigned)gx, (unsigned)(gy < 65535 ? gy : 65535));
  if (vec) hipLaunchKernelGGL(copy2d_kernel<true>, grid, dim3(256), 0, h->stream, m, n, A, lda, B, ldb);
  else hipLaunchKernelGGL(copy2d_kernel<false>, grid, dim3(256), 0, h->stream, m, n, A, lda, B, ldb);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_dlacpy(capi_handle_t h, int part, int64_t m, int64_t n, const double* A, int64_t lda, double* B, int64_t ldb) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, part >= 0 && part <= 2 && m >= 0 && n >= 0, "args");
  if (m == 0 || n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, A && B && lda >= m && ldb >= m, "operands");
  if (part == 0) return capi_internal_copy2d(h, m, n, A, lda, B, ldb);
  const int64_t gy = cdiv(n, 8);
  CAPI_REQUIRE(h, gy <= 65535, "n too large");
  hipLaunchKernelGGL(lacpy_kernel, dim3((unsigned)cdiv(m, 256), (unsigned)gy), dim3(256), 0, h->stream, part, m, n, A, lda, B, ldb);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_dgeadd(capi_handle_t h, int part, int64_t m, int64_t n, double alpha, const double* X, int64_t ldx, double beta, double* Y,
                int64_t ldy) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, part >= 0 && part <= 2 && m >= 0 && n >= 0, "args");
  if (m == 0 || n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, X && Y && ldx >= m && ldy >= m, "operands");
  hipLaunchKernelGGL(geadd_kernel, grid2(m, n), dim3(256), 0, h->stream, part, m, n, alpha, X, ldx, beta, Y, ldy);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_dtrizero(capi_handle_t h, int keep_uplo, int64_t n, double* A, int64_t lda) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, (keep_uplo == 0 || keep_uplo == 1) && n >= 0, "args");
  if (n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, A && lda >= n, "A/lda");
  hipLaunchKernelGGL(trizero_kernel, grid2(n, n), dim3(256), 0, h->stream, keep_uplo, n, A, lda);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_daxpby(capi_handle_t h, int64_t count, double beta, const double* x, double* y) {
  CAPI_REQUIRE(h, h, "null handle");
  if (count <= 0) return count < 0 ? CAPI_EINVAL : CAPI_OK;
  CAPI_REQUIRE(h, x && y, "null vector");
  int64_t blocks = cdiv(count, 256);
  if (blocks > h->num_cu * 16) blocks = h->num_cu * 16;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, count, beta, x, y);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_remove_triangle(capi_handle_t h, char dir, double* A, int64_t dimX, int64_t dimY, int64_t px, int64_t py, int64_t P) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, (dir == 'U' || dir == 'L') && dimX >= 0 && dimY >= 0 && P > 0, "args");
  if (dimX == 0 || dimY == 0) return CAPI_OK;
  hipLaunchKernelGGL(remove_triangle_kernel, grid2(dimY, dimX), dim3(256), 0, h->stream, dir == 'U', A, dimX, dimY, px, py, P);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_block_to_cyclic(capi_handle_t h, const double* blocked, double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  CAPI_REQUIRE(h, h && blocked && cyclic && rl > 0 && cl > 0 && d > 0, "args");
  hipLaunchKernelGGL(block_cyclic_kernel, grid2(rl * d, cl * d), dim3(256), 0, h->stream, 1, (double*)blocked, cyclic, rl, cl, d);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}
int capi_block_to_cyclic_full(capi_handle_t h, const double* blocked, double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  CAPI_REQUIRE(h, h && blocked && cyclic && rl > 0 && cl > 0 && d > 0, "args");
  hipLaunchKernelGGL(block_cyclic_kernel, grid2(rl * d, cl * d), dim3(256), 0, h->stream, 2, (double*)blocked, cyclic, rl, cl, d);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_cyclic_to_block(capi_handle_t h, double* blocked, const double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  CAPI_REQUIRE(h, h && blocked && cyclic && rl > 0 && cl > 0 && d > 0, "args");
  hipLaunchKernelGGL(block_cyclic_kernel, grid2(rl * d, cl * d), dim3(256), 0, h->stream, 0, blocked, (double*)cyclic, rl, cl, d);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_block_to_cyclic_tri(capi_handle_t h, const double* blocked, double* cyclic, int64_t rl, int64_t d) {
  CAPI_REQUIRE(h, h && blocked && cyclic && rl > 0 && d > 0, "args");
  hipLaunchKernelGGL(block_cyclic_tri_kernel, grid2(rl * d, rl * d), dim3(256), 0, h->stream, 1, (double*)blocked, cyclic, rl, d);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}
int capi_cyclic_to_block_tri(capi_handle_t h, double* blocked, const double* cyclic, int64_t rl, int64_t d) {
  CAPI_REQUIRE(h, h && blocked && cyclic && rl > 0 && d > 0, "args");
  hipLaunchKernelGGL(block_cyclic_tri_kernel, grid2(rl * d, rl * d), dim3(256), 0, h->stream, 0, blocked, (double*)cyclic, rl, d);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_cyclic_to_local(capi_handle_t h, double* T, double* TI, int64_t local_dim, int64_t bc_dim, int64_t d, int64_t slice_rank) {
  CAPI_REQUIRE(h, h && T && TI && local_dim > 0 && d > 0 && bc_dim >= local_dim * d && slice_rank >= 0 && slice_rank < d * d, "args");
  void* w;
  int rc = capi_ws2_get(h, sizeof(double) * 2 * (size_t)local_dim * (size_t)local_dim, &w);
  if (rc != CAPI_OK) return rc;
  for (int pass = 0; pass < 2; ++pass)
    hipLaunchKernelGGL(cyclic_to_local_kernel, grid2(local_dim, local_dim), dim3(256), 0, h->stream, pass, T, TI, (double*)w, local_dim, bc_dim, d,
                       slice_rank / d, slice_rank % d);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_distribute_symmetric(capi_handle_t h, double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY, int64_t px,
                              int64_t py, int64_t PX, int64_t PY, int64_t key, int dd) {
  (void)key;  // the reference reseeds per element (structure.hpp:80-85), so `key` never reaches the output
  CAPI_REQUIRE(h, h && data && dimX > 0 && dimY > 0 && PX > 0 && PY > 0, "args");
  int64_t padX, padY;
  pad_lens(dimX, dimY, gdimX, gdimY, px, py, PX, PY, &padX, &padY);
  hipLaunchKernelGGL(gen_symmetric_kernel, grid2(dimY, dimX), dim3(256), 0, h->stream, data, dimX, dimY, gdimX, gdimY, px, py, PX, PY,
                     padX, padY, dd);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_distribute_random(capi_handle_t h, double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY, int64_t px,
                           int64_t py, int64_t PX, int64_t PY, int64_t key) {
  CAPI_REQUIRE(h, h && data && dimX > 0 && dimY > 0 && PX > 0 && PY > 0, "args");
  int64_t padX, padY;
  pad_lens(dimX, dimY, gdimX, gdimY, px, py, PX, PY, &padX, &padY);
  if (padX != dimX || padY != dimY) {
    hipLaunchKernelGGL(gen_pad_zero_kernel, grid2(dimY, dimX), dim3(256), 0, h->stream, data, dimX, dimY, padX, padY);
    CAPI_HIP_CHECK(h, hipGetLastError());
  }
  const int64_t chunks = cdiv(padX * padY, GEN_CHUNK);
  if (chunks > 0) {
    hipLaunchKernelGGL(gen_random_kernel, dim3((unsigned)cdiv(chunks, 256)), dim3(256), 0, h->stream, data, dimX, dimY, padX, padY, key);
    CAPI_HIP_CHECK(h, hipGetLastError());
  }
  return CAPI_OK;
}

int capi_distribute_identity(capi_handle_t h, double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY, int64_t px,
                             int64_t py, int64_t PX, int64_t PY, double val) {
  CAPI_REQUIRE(h, h && data && dimX > 0 && dimY > 0 && PX > 0 && PY > 0, "args");
  int64_t padX, padY;
  pad_lens(dimX, dimY, gdimX, gdimY, px, py, PX, PY, &padX, &padY);
  hipLaunchKernelGGL(gen_identity_kernel, grid2(dimY, dimX), dim3(256), 0, h->stream, data, dimX, dimY, px, py, PX, PY, padX, padY, val);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_diff_norms(capi_handle_t h, int part, int64_t m, int64_t n, const double* X, int64_t ldx, const double* Y, int64_t ldy,
                    double* out2) {
  CAPI_REQUIRE(h, h && out2 && part >= 0 && part <= 2 && m >= 0 && n >= 0, "args");
  out2[0] = out2[1] = 0.0;
  if (m == 0 || n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, X && Y && ldx >= m && ldy >= m, "operands");
  void* w;
  int rc = capi_ws_get(h, sizeof(double) * 2 * (size_t)n, &w);
  if (rc != CAPI_OK) return rc;
  hipLaunchKernelGGL(diff_norms_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, part, m, n, X, ldx, Y, ldy, (double*)w);
  CAPI_HIP_CHECK(h, hipGetLastError());
  double* host = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  if (!host) return CAPI_ENOMEM;
  rc = capi_memcpy_d2h(h, host, w, sizeof(double) * 2 * (size_t)n);
  if (rc == CAPI_OK)
    for (int64_t j = 0; j < n; ++j) { out2[0] += host[2 * j]; out2[1] += host[2 * j + 1]; }
  free(host);
  return rc;
}

}  // extern "C"
