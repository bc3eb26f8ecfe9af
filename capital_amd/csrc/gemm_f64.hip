// gemm_f64.hip -- fp64 MFMA (v_mfma_f64_16x16x4_f64) tile kernel for gfx950 and the BLAS-level
// C-ABI entry points built on it: capi_dgemm / capi_dgemmt / capi_dsyrk / capi_dtrmm[_oop].
//
// Replaces cblas_dgemm / cblas_dsyrk / cblas_dtrmm behind the reference's blas::engine
// (src/blas/interface.hpp:43-97).  One templated kernel serves all of them:
//   * 128x128 output tile per 256-thread workgroup (4 waves = one per SIMD, 2 workgroups per CU),
//     each wave owns a 64x64 block = 4x4 MFMA tiles of 16x16 (64 fp64 accumulators per lane);
//   * K is consumed in 16-deep panels, double-buffered in LDS (one barrier per panel), staged
//     global -> registers (16-byte coalesced loads) -> LDS while the MFMAs of the current panel run;
//   * every operand panel sits in LDS k-contiguous ([row][k] with a 2-double pad): a k-contiguous operand is staged
//     by a pure 16-byte copy, a row-contiguous one is transposed on the way in (16-byte loads of two rows at one k,
//     two 8-byte LDS stores), so all four variants read 16-byte fragments and share one pinned inner loop;
//   * the MFMA's A operand is fed from the N-side panel and its B operand from the M-side panel, so the
//     accumulator's lane index runs along C's column-major rows and the epilogue stores 128-byte runs;
//   * triangular work is skipped at tile granularity: GEMMT/SYRK launch only the tiles of the wanted
//     triangle, TRMM shortens each tile's k-range to the non-zero band and masks the diagonal panels;
//   * split-K (tall-skinny Gram matrices of CholeskyQR2) writes per-slice partial tiles to a slab that a
//     second kernel reduces in a fixed order (bit-reproducible, no atomics);
//   * workgroup ids are re-dealt so that the 64 tiles an XCD runs concurrently form a compact block
//     of the output (8 XCDs x private 4 MiB L2): bands of 8 tile rows, for full and for triangular outputs alike.
// Around it, in this file: dtrmm_pair_kernel (a TRMM whose launch is whole resident rounds: two tiles of complementary k-range per
// workgroup, the second walked backwards in k), dgemm_small_kernel (order <= 512: 32 x 32 tiles, K in 256-deep bursts -- the recursion's
// latency-bound levels), gram_ts_kernel and trmm_right_ts32_kernel (CholeskyQR2's tall-skinny Gram matrix and Q = A R^-1,
// full-width workgroups that read the tall operand once), and launch_gemm, which picks between them with a makespan
// model in CU-cycles.
// Diagnostic environment switches (read once): CAPI_DEBUG_GEMM (print every choice), CAPI_FORCE_TS=64|128, CAPI_SMALL=0|1,
// CAPI_NO_TS, CAPI_TS_ROWS16, CAPI_NO_SHARE, CAPI_NO_SKIP, CAPI_NO_ROTATE, CAPI_NO_TAIL, CAPI_THIN_ROUNDS, CAPI_PEAK_BLOCKS_PER_CU,
// CAPI_TILE_ORDER=0, CAPI_TRMM_PAIR=0|2, CAPI_ROUNDS=1, CAPI_PROF_DUMP.
#include "capi_internal.h"
#include <type_traits>

extern "C" int capi_internal_copy2d(capi_handle_t h, int64_t m, int64_t n, const double* A, int64_t lda, double* B, int64_t ldb);   // movement.hip

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 16, NTHREADS = 256;
constexpr int SK = BK + 2;            // [row][k] layout: row stride in doubles (144 B, 16-B aligned, odd multiple of 16 B)
// Two tile sizes share the code: TS = 128 (4x4 MFMA tiles per wave) for large outputs, TS = 64 (2x2 per wave, 4
// workgroups per CU) when a 128-tiling would leave CUs idle (the small levels of the recursion).
template <int TS> struct tile_cfg {
  static constexpr int SR = TS + 8;                 // [k][row] layout: k-row stride in doubles
  static constexpr int TILE_LDS = TS * SK;          // >= BK*SR
  static constexpr int STAGE_LDS = 2 * TILE_LDS;
  static constexpr int NQ = TS / 32;                // 16-byte pieces per thread per operand panel
  static constexpr int SUB = TS / 32;               // MFMA tiles per wave per dimension
};
constexpr int GROUP_M = 8;
#ifndef CAPI_STORE_IN_SHADOW
#define CAPI_STORE_IN_SHADOW 1
#endif
constexpr bool STORE_IN_SHADOW = CAPI_STORE_IN_SHADOW;
#ifndef CAPI_LOADS_IN_SHADOW
#define CAPI_LOADS_IN_SHADOW 1
#endif
constexpr bool LOADS_IN_SHADOW_ON = CAPI_LOADS_IN_SHADOW;
// A row-contiguous operand ([k][row] in HBM) is transposed while it is staged, so that LDS holds every operand [row][k] and all
// four variants run the k-contiguous variant's inner loop (16-byte fragment reads, the pinned schedule).
#ifndef CAPI_TRANSPOSE_STAGE
#define CAPI_TRANSPOSE_STAGE 1
#endif
constexpr bool TRANSPOSE_STAGE = CAPI_TRANSPOSE_STAGE;
// thread -> element map of a row-contiguous operand's panel when it is staged transposed: piece q of thread tid is rows
// rc_r, rc_r + 1 at k = rc_k.  A wave instruction covers 16 rows x 8 k: eight whole 128-byte lines on the way in, and
// (rows 18 doubles apart in LDS) bank pairs (8 rp + 2 k) mod 64 on the way out -- two-way conflicts at most.
template <int TS> __device__ __forceinline__ int rc_r(int tid, int q) { return 16 * ((tid >> 6) * (TS / 64) + (q >> 1)) + 2 * (tid & 7); }
__device__ __forceinline__ int rc_k(int tid, int q) { return 8 * (q & 1) + ((tid >> 3) & 7); }

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  double alpha, beta;
  int out_uplo;       // -1: full output; CAPI_UPPER / CAPI_LOWER: only that triangle of C (M == N)
  int tri_side;       // -1: none; CAPI_LEFT: op(A) is triangular (M == K); CAPI_RIGHT: op(B) is triangular (N == K)
  int tri_eff_upper;  // op(T) is upper triangular
  int tri_unit;
  int a_vec, b_vec;   // 16-byte loads legal for A / B
  int splitk, k_per_split;
  int k_rotate;       // split-K slice z starts its k-loop z/splitk of the way through its range and wraps around
  double* slab;       // split-K partial sums: slab[z*slab_stride + i + j*slab_ld]
  int64_t slab_ld, slab_stride;
  int tiles_m, tiles_n, ntiles;
  int batch;          // burst-load kernel only: blockIdx.y indexes `batch` independent products, operands sa/sb/sc elements apart
  int64_t sa, sb, sc;
  int tail_base, tail_tm, tail_tn;   // > 0: this 64-tile launch covers the four quarters of the 128-tiles [tail_base, ...) of a tail_tm x tail_tn tiling
  int ts;             // tile size chosen by the launcher
  int no_skip;        // diagnostics: never skip zero sub-tiles
  int share_ab;       // A and B are the same matrix in the same orientation (syrk): diagonal tiles stage ONE panel
  int tri_dense;      // the triangular operand is a clean copy whose other triangle holds zeros: the panels that cross its diagonal
                      // need neither masking nor sub-tile bookkeeping and run in the FAST loop (tall right-TRMM, see trmm_launch)
  int tri_koff;       // tri_dense, right side, upper: this launch holds columns [tri_koff, tri_koff + N) of op(T): column tile tj's k-range
                      // ends at tri_koff + (tj + 1) TS.  Tiles are then dealt like a plain product's (tri_block)
  int tri_block;
  int pid_base;       // this launch covers pids [pid_base, pid_base + gridDim.x) of the tile enumeration (launch in resident rounds)
  int order;          // bit 0: a triangular output's tiles run in bands of 8 tile rows (tile_of_dims)
  int a_tiled, c_tiled;   // tall-skinny kernels (n == 256): the tall operand / the output is a "panel32" image -- tiles of 32 rows x 256 columns,
                      // each column-major with ld 32, tile t at 32 * 256 * t: every pass over the panel is ONE contiguous stream
  unsigned long long* stamp;   // capi_prof_*: this launch's record {min over workgroups of the start time, min of ~(end time)} in 100 MHz wall-clock
                               // ticks, both initialised to ~0 -- the launch's EXECUTION interval, whatever else shares the device with it
};

// One atomic per workgroup at either end (512 per resident round): the interval in which the launch really had workgroups on the device.
// HIP events around a launch are stream-ordered: with launches of several streams interleaved round by round, an event bracket also
// contains the neighbours' rounds (round 3: the brackets summed to 1.11 x the step) -- these stamps do not.
__device__ __forceinline__ void stamp_begin(const GemmArgs& p) {
  if (p.stamp && threadIdx.x == 0) atomicMin(p.stamp, (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void stamp_end(const GemmArgs& p) {
  if (p.stamp && threadIdx.x == 0) atomicMin(p.stamp + 1, ~(unsigned long long)wall_clock64());
}

// ---- global -> registers: this thread's 4 x 16 bytes of a 128 x 16 operand panel -----------------
template <int TS, bool KC>
__device__ __forceinline__ void panel_load(const double* __restrict__ X, int64_t ld, int r0, int R, int k0, int kend,
                                           int tid, bool vec_ok, d2_t (&v)[TS / 32]) {
  constexpr int NQ = TS / 32;
  if (KC) {  // element (r,k) at X[k + r*ld]
    const int kp = tid & 7, rb = tid >> 3;
    const int k = k0 + 2 * kp;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = r0 + rb + 32 * q;
      d2_t val = {0.0, 0.0};
      if (r < R) {
        const double* p = X + (int64_t)r * ld + k;
        if (vec_ok && k + 1 < kend) {
          val = *(const d2_t*)p;
        } else {
          if (k < kend) val.x = p[0];
          if (k + 1 < kend) val.y = p[1];
        }
      }
      v[q] = val;
    }
  } else {  // element (r,k) at X[r + k*ld]
    const int rp = tid & (TS / 2 - 1), kb = tid / (TS / 2);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int r = r0 + (TRANSPOSE_STAGE ? rc_r<TS>(tid, q) : 2 * rp);
      const int k = k0 + (TRANSPOSE_STAGE ? rc_k(tid, q) : kb + (512 / TS) * q);
      d2_t val = {0.0, 0.0};
      if (k < kend) {
        const double* p = X + (int64_t)k * ld + r;
        if (vec_ok && r + 1 < R) {
          val = *(const d2_t*)p;
        } else {
          if (r < R) val.x = p[0];
          if (r + 1 < R) val.y = p[1];
        }
      }
      v[q] = val;
    }
  }
}

// interior fast path: the whole panel is in range and 16-byte aligned -- four straight-line loads, no predicates.
// `base` already points at this thread's first piece of the panel; piece q lies (q & 1) s1 + (q >> 1) s2 elements further.
template <int TS>
__device__ __forceinline__ void panel_load_fast(const double* __restrict__ base, int64_t s1, int64_t s2, d2_t (&v)[TS / 32]) {
#pragma unroll
  for (int q = 0; q < TS / 32; ++q) v[q] = *(const d2_t*)(base + (q & 1) * s1 + (q >> 1) * s2);
}

// zero what lies outside the triangle of op(T) (and force a unit diagonal) on a staged panel.
// keep_ge: keep k >= r, else keep k <= r  (r = the panel's row index in op(T) coordinates)
template <int TS, bool KC>
__device__ __forceinline__ void panel_mask(int r0, int k0, int tid, bool keep_ge, bool unit, d2_t (&v)[TS / 32]) {
#pragma unroll
  for (int q = 0; q < TS / 32; ++q) {
    int r[2], k[2];
    if (KC) {
      r[0] = r[1] = r0 + (tid >> 3) + 32 * q;
      k[0] = k0 + 2 * (tid & 7);
      k[1] = k[0] + 1;
    } else {
      r[0] = r0 + (TRANSPOSE_STAGE ? rc_r<TS>(tid, q) : 2 * (tid & (TS / 2 - 1)));
      r[1] = r[0] + 1;
      k[0] = k[1] = k0 + (TRANSPOSE_STAGE ? rc_k(tid, q) : tid / (TS / 2) + (512 / TS) * q);
    }
    double e0 = v[q].x, e1 = v[q].y;
    if (keep_ge ? (k[0] < r[0]) : (k[0] > r[0])) e0 = 0.0;
    if (keep_ge ? (k[1] < r[1]) : (k[1] > r[1])) e1 = 0.0;
    if (unit) {
      if (k[0] == r[0]) e0 = 1.0;
      if (k[1] == r[1]) e1 = 1.0;
    }
    v[q].x = e0;
    v[q].y = e1;
  }
}

template <int TS, bool KC>
__device__ __forceinline__ void panel_store(double* __restrict__ L, int tid, const d2_t (&v)[TS / 32]) {
  constexpr int SR = tile_cfg<TS>::SR;
  if (KC) {
    const int kp = tid & 7, rb = tid >> 3;
#pragma unroll
    for (int q = 0; q < TS / 32; ++q) *(d2_t*)&L[(rb + 32 * q) * SK + 2 * kp] = v[q];
  } else if (TRANSPOSE_STAGE) {
#pragma unroll
    for (int q = 0; q < TS / 32; ++q) {
      double* d = &L[rc_r<TS>(tid, q) * SK + rc_k(tid, q)];
      d[0] = v[q].x;
      d[SK] = v[q].y;
    }
  } else {
    const int rp = tid & (TS / 2 - 1), kb = tid / (TS / 2);
#pragma unroll
    for (int q = 0; q < TS / 32; ++q) *(d2_t*)&L[(kb + (512 / TS) * q) * SR + 2 * rp] = v[q];
  }
}

// fragment for MFMA k-steps 2u and 2u+1 of 16-row sub-block `row` (panel row index of this lane).
// k-step s = 2u+e of lane group g = lane>>4 consumes physical k = 8u + 2g + e in BOTH operands.
template <int TS, bool KC>
__device__ __forceinline__ d2_t frag_read(const double* __restrict__ L, int row, int u, int g) {
  constexpr int SR = tile_cfg<TS>::SR;
  if (KC || TRANSPOSE_STAGE) {
    return *(const d2_t*)&L[row * SK + 8 * u + 2 * g];
  } else {
    d2_t f;
    f.x = L[(8 * u + 2 * g) * SR + row];
    f.y = L[(8 * u + 2 * g + 1) * SR + row];
    return f;
  }
}

// One k-step (4 deep) of the wave's SUB x SUB MFMA tiles.  `keep` has one bit per (a,b) sub-tile: sub-tiles that are
// provably all-zero products (below the diagonal of a triangular output tile, or outside the band of a triangular
// operand in the diagonal zone of a TRMM tile) are skipped at MFMA granularity (16 x 16) instead of tile granularity.
template <int SUB, int COMP, bool ALL>
__device__ __forceinline__ void mfma_step(d4_t (&acc)[SUB][SUB], const d2_t (&af)[SUB], const d2_t (&bf)[SUB], unsigned keep) {
  constexpr unsigned FULL = (1u << (SUB * SUB)) - 1u;
  if (ALL || keep == FULL) {
#pragma unroll
    for (int a = 0; a < SUB; ++a)
#pragma unroll
      for (int b = 0; b < SUB; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(COMP ? bf[b].y : bf[b].x, COMP ? af[a].y : af[a].x, acc[a][b], 0, 0, 0);
  } else {
#pragma unroll
    for (int a = 0; a < SUB; ++a)
#pragma unroll
      for (int b = 0; b < SUB; ++b)
        if (keep & (1u << (a * SUB + b)))
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(COMP ? bf[b].y : bf[b].x, COMP ? af[a].y : af[a].x, acc[a][b], 0, 0, 0);
  }
}

__device__ __forceinline__ void tile_of_dims(int out_uplo, int nm, int nn, int t, int& ti, int& tj, int order = 0) {
  if (out_uplo >= 0 && (order & 1)) {
    // pairs (lo <= hi) in bands of GROUP_M values of lo: the band's diagonal triangle first, then hi outwards with the band's
    // GROUP_M values of lo under each -- an XCD's 64 resident tiles then form an 8 x 8 block of the triangle (16 panels
    // instead of 65 behind 64 tiles), as a plain product's do
    int first = 0, gsz = GROUP_M;
    for (;; first += GROUP_M) {
      gsz = min(nn - first, GROUP_M);
      const int cnt = gsz * (gsz + 1) / 2 + gsz * (nn - first - gsz);
      if (t < cnt || first + GROUP_M >= nn) break;
      t -= cnt;
    }
    const int ntri = gsz * (gsz + 1) / 2;
    int lo, hi;
    if (t < ntri) {
      hi = 0;
      while ((hi + 1) * (hi + 2) / 2 <= t) ++hi;
      lo = first + t - hi * (hi + 1) / 2;
      hi += first;
    } else {
      t -= ntri;
      hi = first + gsz + t / gsz;
      lo = first + t % gsz;
    }
    if (out_uplo == CAPI_UPPER) { ti = lo; tj = hi; } else { ti = hi; tj = lo; }
  } else if (out_uplo < 0) {
    // bands of GROUP_M tile-rows (all tile-columns) are dealt to XCDs in order
    const int in_group = GROUP_M * nn;
    const int group = t / in_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(nm - first_m, GROUP_M);
    const int loc = t - group * in_group;
    ti = first_m + loc % gsz;
    tj = loc / gsz;
  } else {
    // t enumerates pairs (lo <= hi): t = hi*(hi+1)/2 + lo
    int hi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)hi * (hi + 1) / 2 > t) --hi;
    while ((int64_t)(hi + 1) * (hi + 2) / 2 <= t) ++hi;
    const int lo = t - (int)((int64_t)hi * (hi + 1) / 2);
    if (out_uplo == CAPI_UPPER) { ti = lo; tj = hi; } else { ti = hi; tj = lo; }
  }
}

__device__ __forceinline__ void tile_of(const GemmArgs& p, int t, int& ti, int& tj) {
  if (p.tail_base > 0) {
    // the last, partially filled round of a 128-tiling is re-cut into 64-tiles: quarter t & 3 of 128-tile tail_base + t / 4
    int ti128, tj128;
    tile_of_dims(p.out_uplo, p.tail_tm, p.tail_tn, p.tail_base + (t >> 2), ti128, tj128, p.order);
    ti = 2 * ti128 + (t & 1);
    tj = 2 * tj128 + ((t >> 1) & 1);
  } else {
    tile_of_dims(p.out_uplo, p.tiles_m, p.tiles_n, t, ti, tj, p.order);
  }
}

// TRMM: a tile's k-range grows linearly along the triangular dimension, so the tiles are ranked longest first and
// rank r goes to workgroup r.  The dispatcher deals workgroups round-robin over the 8 XCDs: every XCD receives every
// 8th entry of the sorted list -- equal shares of long and short tiles, each share still longest-first (LPT), and with
// a multiple of 8 tiles across the free dimension an XCD keeps seeing the same columns of the dense operand in its L2.
__device__ __forceinline__ void trmm_tile_of(const GemmArgs& p, int r, int& ti, int& tj) {
  const bool left = p.tri_side == CAPI_LEFT;
  const int nfree = left ? p.tiles_n : p.tiles_m, ntri = left ? p.tiles_m : p.tiles_n;
  int b = r / nfree;                       // position along the triangular dimension, 0 = longest k-range
  const int a = r - b * nfree;
  const bool longest_last = left ? !p.tri_eff_upper : (p.tri_eff_upper != 0);
  if (longest_last) b = ntri - 1 - b;
  ti = left ? b : a;
  tj = left ? a : b;
}

template <int TS, bool AK, bool BKC>
__global__ __launch_bounds__(NTHREADS, TS == 128 ? 2 : 4) void dgemm_tile_kernel(const GemmArgs p) {
  constexpr int BM = TS, BN = TS, SUB = tile_cfg<TS>::SUB, NQ = tile_cfg<TS>::NQ;
  // the fully pinned iteration schedule counts 16-byte fragment reads: both operands k-contiguous (measured: +1.5 % there,
  // -3 % on the variants whose row-contiguous operand is read in 8-byte pieces)
  constexpr bool LOADS_IN_SHADOW = LOADS_IN_SHADOW_ON && ((AK && BKC) || TRANSPOSE_STAGE);
  constexpr int NW = NQ * ((AK || !TRANSPOSE_STAGE ? 1 : 2) + (BKC || !TRANSPOSE_STAGE ? 1 : 2));   // LDS stores of one staging pass
  constexpr int TILE_LDS = tile_cfg<TS>::TILE_LDS, STAGE_LDS = tile_cfg<TS>::STAGE_LDS;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  stamp_begin(p);

  // XCD-aware re-deal: consecutive pids share an XCD (dispatcher deals blockIdx round-robin over 8 XCDs)
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
  const int pid = p.pid_base + (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  int z = pid / p.ntiles;
  int ti, tj;
  if (p.tri_side >= 0 && p.splitk == 1 && !p.tri_block) { z = 0; trmm_tile_of(p, bid, ti, tj); }
  else tile_of(p, pid - z * p.ntiles, ti, tj);
  if (p.tail_base > 0) {                      // quarters outside the matrix or wholly in the unwanted triangle (uniform per workgroup)
    if (ti >= p.tiles_m || tj >= p.tiles_n) return;
    if (p.out_uplo == CAPI_UPPER && ti > tj) return;
    if (p.out_uplo == CAPI_LOWER && tj > ti) return;
  }
  const int i0 = ti * BM, j0 = tj * BN;

  // k-range of this tile
  int klo = 0, khi = p.K;
  if (p.tri_side == CAPI_LEFT) {
    if (p.tri_eff_upper) klo = i0; else khi = min(p.K, i0 + BM);
  } else if (p.tri_side == CAPI_RIGHT) {
    if (p.tri_eff_upper) khi = min(p.K, p.tri_koff + j0 + BN); else klo = j0;
  }
  constexpr int kstep = BK;
  if (p.splitk > 1) {
    klo = max(klo, z * p.k_per_split);
    khi = min(khi, (z + 1) * p.k_per_split);
  }
  const int ntk = khi > klo ? (khi - klo + kstep - 1) / kstep : 0;
  const bool maskA = p.tri_side == CAPI_LEFT && !p.tri_dense, maskB = p.tri_side == CAPI_RIGHT && !p.tri_dense;
  const bool shareB = AK == BKC && p.share_ab && ti == tj;   // B panel == A panel: load and stage it once
  const bool keep_ge = (p.tri_side == CAPI_LEFT) == (p.tri_eff_upper != 0);

  d4_t acc[SUB][SUB];
#pragma unroll
  for (int a = 0; a < SUB; ++a)
#pragma unroll
    for (int b = 0; b < SUB; ++b) acc[a][b] = (d4_t){0.0, 0.0, 0.0, 0.0};

  // per-thread source of the fast path (see panel_load for the thread -> element map)
  const bool interior = (i0 + BM <= p.M) && (j0 + BN <= p.N) && p.a_vec && p.b_vec;
  const double* fa = AK ? p.A + (int64_t)(i0 + (tid >> 3)) * p.lda + klo + 2 * (tid & 7)
                     : TRANSPOSE_STAGE ? p.A + (int64_t)(klo + rc_k(tid, 0)) * p.lda + i0 + rc_r<TS>(tid, 0)
                                       : p.A + (int64_t)(klo + tid / (TS / 2)) * p.lda + i0 + 2 * (tid & (TS / 2 - 1));
  const double* fb = BKC ? p.B + (int64_t)(j0 + (tid >> 3)) * p.ldb + klo + 2 * (tid & 7)
                     : TRANSPOSE_STAGE ? p.B + (int64_t)(klo + rc_k(tid, 0)) * p.ldb + j0 + rc_r<TS>(tid, 0)
                                       : p.B + (int64_t)(klo + tid / (TS / 2)) * p.ldb + j0 + 2 * (tid & (TS / 2 - 1));
  // piece q of the fast path lies (q & 1) qsa + (q >> 1) qsa2 elements behind piece 0
  const int64_t qsa = AK ? 32 * p.lda : TRANSPOSE_STAGE ? 8 * p.lda : (512 / TS) * p.lda;
  const int64_t qsa2 = AK ? 64 * p.lda : TRANSPOSE_STAGE ? 16 : 2 * (512 / TS) * p.lda;
  const int64_t qsb = BKC ? 32 * p.ldb : TRANSPOSE_STAGE ? 8 * p.ldb : (512 / TS) * p.ldb;
  const int64_t qsb2 = BKC ? 64 * p.ldb : TRANSPOSE_STAGE ? 16 : 2 * (512 / TS) * p.ldb;
  const int64_t ksa = AK ? kstep : (int64_t)kstep * p.lda, ksb = BKC ? kstep : (int64_t)kstep * p.ldb;

  // triangular output: on a diagonal tile the sub-tiles lying entirely in the unwanted triangle are never computed
  unsigned out_keep = (1u << (SUB * SUB)) - 1u;
  if (p.out_uplo >= 0 && ti == tj && !p.no_skip) {
#pragma unroll
    for (int a = 0; a < SUB; ++a)
#pragma unroll
      for (int b = 0; b < SUB; ++b) {
        const int rlo = wm * (TS / 2) + 16 * a, clo = wn * (TS / 2) + 16 * b;
        const bool on = p.out_uplo == CAPI_UPPER ? (rlo <= clo + 15) : (rlo + 15 >= clo);
        if (!on) out_keep &= ~(1u << (a * SUB + b));
      }
  }

  // Rotation (split-K without a triangular operand): every slice of a tall operand walks rows that sit a multiple of
  // the slice length apart -- a power of two times 128 bytes in practice, so all slices would sweep the SAME memory
  // channel at the same moment and move on together.  Slice z starts z/splitk of the way through and wraps.
  const int rot = (p.k_rotate && p.splitk > 1 && ntk > 1) ? (int)((int64_t)z * ntk / p.splitk) : 0;
  const int kfirst = klo + rot * kstep;
  d2_t ra[NQ], rb[NQ];
  if (ntk > 0) {
    panel_load<TS, AK>(p.A, p.lda, i0, p.M, kfirst, khi, tid, p.a_vec, ra);
    if (!shareB) panel_load<TS, BKC>(p.B, p.ldb, j0, p.N, kfirst, khi, tid, p.b_vec, rb);
    if (maskA && klo < i0 + BM && klo + BK > i0) panel_mask<TS, AK>(i0, klo, tid, keep_ge, p.tri_unit, ra);
    if (maskB && klo < j0 + BN && klo + BK > j0) panel_mask<TS, BKC>(j0, klo, tid, keep_ge, p.tri_unit, rb);
    panel_store<TS, AK>(lds, tid, ra);
    if (!shareB) panel_store<TS, BKC>(lds + TILE_LDS, tid, rb);
  }
  __syncthreads();

  // One iteration = prefetch panel t+1 into registers, MFMA over panel t from LDS, stage panel t+1, barrier.
  // FAST iterations (interior tile, successor panel entirely in range) are a loop of their own: sharing one loop with
  // the predicated loads made the register allocator keep two copies of the staging registers and wait for the
  // prefetch (s_waitcnt vmcnt) BEFORE the MFMA phase to reconcile them -- the load latency it was meant to hide.
  int par = 0;                              // LDS stage holding the panel being multiplied
  auto iterate = [&](const int t, const int tnext, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const double* La = lds + par * STAGE_LDS;
    const double* Lb = shareB ? La : La + TILE_LDS;
    const int kn = klo + tnext * kstep;
    const bool more = FAST || (tnext >= 0);
    if (FAST) {
      fa += ksa;
      fb += ksb;
      if (LOADS_IN_SHADOW) __builtin_amdgcn_s_setprio(1);     // one scheduling region from here to the end of the MFMAs
      panel_load_fast<TS>(fa, qsa, qsa2, ra);
      panel_load_fast<TS>(fb, qsb, qsb2, rb);       // (on a shared diagonal tile this re-reads A's panel: cache hit, never staged)
    } else if (more) {
      panel_load<TS, AK>(p.A, p.lda, i0, p.M, kn, khi, tid, p.a_vec, ra);
      if (!shareB) panel_load<TS, BKC>(p.B, p.ldb, j0, p.N, kn, khi, tid, p.b_vec, rb);
    }
    // sub-tile activity of this wave for this panel (wave-uniform).  FAST iterations are chosen so that every sub-tile
    // is live and neither this panel nor the next crosses the diagonal band of a triangular operand.
    unsigned keep = (1u << (SUB * SUB)) - 1u;
    if (!FAST) {
      keep = out_keep;
      const int kk = klo + t * kstep;
      // only panels that cross the diagonal band of this tile can have dead sub-tiles
      const bool band = p.tri_side == CAPI_LEFT ? (kk < i0 + BM && kk + BK > i0) : (kk < j0 + BN && kk + BK > j0);
      if (p.tri_side >= 0 && band && !p.no_skip && !p.tri_dense) {
#pragma unroll
        for (int a = 0; a < SUB; ++a)
#pragma unroll
          for (int b = 0; b < SUB; ++b) {
            bool on;
            if (p.tri_side == CAPI_LEFT) {
              const int rlo = i0 + wm * (TS / 2) + 16 * a;
              on = p.tri_eff_upper ? (kk + BK - 1 >= rlo) : (kk <= rlo + 15);
            } else {
              const int clo = j0 + wn * (TS / 2) + 16 * b;
              on = p.tri_eff_upper ? (kk <= clo + 15) : (kk + BK - 1 >= clo);
            }
            if (!on) keep &= ~(1u << (a * SUB + b));
          }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      d2_t af[SUB], bf[SUB];
#pragma unroll
      for (int a = 0; a < SUB; ++a) af[a] = frag_read<TS, AK>(La, wm * (TS / 2) + a * 16 + r16, u, g);
#pragma unroll
      for (int b = 0; b < SUB; ++b) bf[b] = frag_read<TS, BKC>(Lb, wn * (TS / 2) + b * 16 + r16, u, g);
      if (!(FAST && LOADS_IN_SHADOW)) __builtin_amdgcn_s_setprio(1);      // keeps the cluster contiguous (+1 % measured)
      mfma_step<SUB, 0, FAST>(acc, af, bf, keep);
      mfma_step<SUB, 1, FAST>(acc, af, bf, keep);
      if (FAST && STORE_IN_SHADOW && u == 1) {
        // the staging stores of the next panel are issued in the shadow of this half's MFMAs (one LDS write per few
        // MFMAs) instead of behind them: the prefetch landed thousands of cycles ago, and the pipe stays fed while the
        // wave issues them.  (B is staged even on a shared diagonal tile, where nobody reads it: no branch in the block.)
        double* Na = lds + (par ^ 1) * STAGE_LDS;
        panel_store<TS, AK>(Na, tid, ra);
        panel_store<TS, BKC>(Na + TILE_LDS, tid, rb);
        if (LOADS_IN_SHADOW) {
          // the whole iteration is one scheduling region; its order is pinned here.  First half: the fragment reads go out
          // first, the next panel's global loads follow in the shadow of the first MFMAs (instead of standing between the
          // barrier and the first MFMA), then the second half's fragment reads in the shadow of the rest.
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * SUB, 0);                     // DS read: fragments of half 0
#pragma unroll
          for (int i = 0; i < 2 * NQ; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, (SUB * SUB) / (2 * NQ), 0);     // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                         // VMEM read
          }
#pragma unroll
          for (int i = 0; i < 2 * SUB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, (SUB * SUB) / (2 * SUB), 0);    // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                         // DS read: fragments of half 1
          }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, (2 * SUB * SUB) / NW > 0 ? (2 * SUB * SUB) / NW : 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                           // DS write
        }
      }
      if (!(FAST && LOADS_IN_SHADOW) || u == 1) __builtin_amdgcn_s_setprio(0);
    }
    if (more && !(FAST && STORE_IN_SHADOW)) {
      if (!FAST) {
        if (maskA && kn < i0 + BM && kn + BK > i0) panel_mask<TS, AK>(i0, kn, tid, keep_ge, p.tri_unit, ra);
        if (maskB && kn < j0 + BN && kn + BK > j0) panel_mask<TS, BKC>(j0, kn, tid, keep_ge, p.tri_unit, rb);
      }
      double* Na = lds + (par ^ 1) * STAGE_LDS;
      panel_store<TS, AK>(Na, tid, ra);
      if (!shareB) panel_store<TS, BKC>(Na + TILE_LDS, tid, rb);
    }
    par ^= 1;
    __syncthreads();
  };
  // Iterations t in [0, nfast) have an interior successor panel.  Of those, FAST ones must also have every sub-tile live
  // (not a diagonal tile of a triangular output) and keep clear of the band [tb0, tb1) of panels that cross the diagonal
  // of a triangular operand -- both as the panel multiplied (t) and as the panel staged (t + 1).
  // FAST iteration t needs its successor panel [klo + (t+1) kstep, + BK) inside [klo, khi)
  // (diagonal tiles of a triangular output take it too: computing their dead sub-tiles in the lean loop is cheaper than
  //  skipping them in the generic one -- the wave holding 16 live sub-tiles sets the pace either way)
  const int nfast = (interior && khi - klo >= BK) ? (khi - klo - BK) / kstep : 0;
  int tb0 = ntk, tb1 = ntk;                 // no band
  if (p.tri_side >= 0 && !p.tri_dense) {
    const int d0 = p.tri_side == CAPI_LEFT ? i0 : j0;
    tb0 = d0 > klo ? (d0 - klo) / BK : 0;
    tb1 = d0 + TS > klo ? (d0 + TS - klo + BK - 1) / BK : 0;
  }
  // two segments of the panel sequence: [rot, ntk) then (after the wrap) [0, rot); without rotation the second is empty
  const double* const fa0 = fa;
  const double* const fb0 = fb;
  for (int seg = 0; seg < 2; ++seg) {
    const int tb = seg == 0 ? rot : 0, te = seg == 0 ? ntk : rot;
    const int after = (seg == 0 && rot > 0) ? 0 : -1;     // successor of the segment's last iteration
    fa = fa0 + (int64_t)tb * ksa;
    fb = fb0 + (int64_t)tb * ksb;
    const int lim = min(nfast, te - 1);                   // a FAST iteration's successor is t + 1, inside the segment
    int t = tb;
    while (t < te) {
      const int fe = t + 1 < tb0 ? min(tb0 - 1, lim) : (t >= tb1 ? lim : t);
      for (; t < fe; ++t) iterate(t, t + 1, std::true_type{});
      // fa/fb advance only in FAST iterations; the generic ones derive their addresses from (i0, j0, kn) -- resynchronise
      const int ge = (t < tb1 && tb1 < lim) ? tb1 : te;
      const int t_in = t;
      for (; t < ge; ++t) iterate(t, t + 1 < te ? t + 1 : after, std::false_type{});
      fa += (int64_t)(t - t_in) * ksa;
      fb += (int64_t)(t - t_in) * ksb;
    }
  }

  // epilogue: lane holds C[i = ..+r16][j = ..+g+4*reg]; 16 lanes -> 128 contiguous bytes of one column.
  // beta is tested once, and with beta != 0 the old values of a whole row of sub-tiles are requested together before any
  // store: a per-element `if (beta != 0) r += beta * *c` is a load in a branch in front of every store, i.e. 64 serialised
  // memory round trips per thread (it dominated the K = 128 trailing updates of the blocked factorization).
  auto element_ok = [&](int i, int j) {
    bool ok = (i < p.M) && (j < p.N);
    if (p.out_uplo == CAPI_UPPER) ok = ok && (i <= j);
    if (p.out_uplo == CAPI_LOWER) ok = ok && (i >= j);
    return ok;
  };
  if (p.splitk > 1) {
#pragma unroll
    for (int a = 0; a < SUB; ++a) {
      const int i = i0 + wm * (TS / 2) + a * 16 + r16;
#pragma unroll
      for (int b = 0; b < SUB; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int j = j0 + wn * (TS / 2) + b * 16 + g + 4 * reg;
          if (element_ok(i, j)) p.slab[(int64_t)z * p.slab_stride + i + (int64_t)j * p.slab_ld] = acc[a][b][reg];
        }
    }
  } else if (p.beta == 0.0) {
#pragma unroll
    for (int a = 0; a < SUB; ++a) {
      const int i = i0 + wm * (TS / 2) + a * 16 + r16;
#pragma unroll
      for (int b = 0; b < SUB; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int j = j0 + wn * (TS / 2) + b * 16 + g + 4 * reg;
          if (element_ok(i, j)) p.C[i + (int64_t)j * p.ldc] = p.alpha * acc[a][b][reg];
        }
    }
  } else {
#pragma unroll
    for (int a = 0; a < SUB; ++a) {
      const int i = i0 + wm * (TS / 2) + a * 16 + r16;
      double old[SUB][4];
#pragma unroll
      for (int b = 0; b < SUB; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int j = j0 + wn * (TS / 2) + b * 16 + g + 4 * reg;
          const bool ok = element_ok(i, j);
          old[b][reg] = *(ok ? p.C + i + (int64_t)j * p.ldc : p.C);      // unconditional load from a clamped address
        }
#pragma unroll
      for (int b = 0; b < SUB; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int j = j0 + wn * (TS / 2) + b * 16 + g + 4 * reg;
          if (element_ok(i, j)) p.C[i + (int64_t)j * p.ldc] = p.alpha * acc[a][b][reg] + p.beta * old[b][reg];
        }
    }
  }
  stamp_end(p);
}

// ---- TRMM in tile PAIRS ------------------------------------------------------------------------------------------
// The tiles of a triangular product have k-ranges of 1 .. ntri panels-of-128, so (dgemm_tile_kernel, trmm_tile_of) they are
// dealt longest first and run free: the tiles resident on an XCD are at unrelated depths of k, every one streams its own
// panel of the dense operand (measured at order 32768: 2 x FETCH_SIZE = 1.19 TB, 1.9-fold sharing), and below order ~8192
// the unequal lengths no longer pack into the 512 slots (order 4096: 50 TFLOP/s).
// Here one workgroup computes tile b AND tile ntri - 1 - b of the same panel of the dense operand: every workgroup does
// (ntri + 1) x 128 of k, the launch is a plain product's (bands of 8 pair-rows x 8 columns per XCD, equal work).  All k-ranges of a triangular operand share one end (the anchor: 0 or K).  Phase 0 walks the SHORT tile away
// from the anchor, phase 1 walks the LONG tile back towards it: a workgroup that entered phase 1 after s panels is at
// depth Ktot - s exactly when its neighbours -- still in phase 0 or already in phase 1 -- are, whatever their own
// lengths.  The 64 workgroups of an XCD therefore walk k in step through both phases and share 8 + 8 panels per step.
// Requirements (launch_gemm): M, N, K multiples of 128, 16-byte aligned operands, beta == 0, an even number of tile rows.
template <bool AK, bool BKC>
__global__ __launch_bounds__(NTHREADS, 2) void dtrmm_pair_kernel(const GemmArgs p) {
  constexpr int TS = 128, BM = TS, BN = TS, SUB = tile_cfg<TS>::SUB, NQ = tile_cfg<TS>::NQ;
  constexpr bool LOADS_IN_SHADOW = LOADS_IN_SHADOW_ON && ((AK && BKC) || TRANSPOSE_STAGE);
  constexpr int NW = NQ * ((AK || !TRANSPOSE_STAGE ? 1 : 2) + (BKC || !TRANSPOSE_STAGE ? 1 : 2));
  constexpr int TILE_LDS = tile_cfg<TS>::TILE_LDS, STAGE_LDS = tile_cfg<TS>::STAGE_LDS;
  constexpr int kstep = BK;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wm = wave & 1, wn = wave >> 1;
  stamp_begin(p);

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
  const int pid = p.pid_base + (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);      // (pid_base: launch in resident rounds)
  const bool left = p.tri_side == CAPI_LEFT;
  const int ntri = left ? p.tiles_m : p.tiles_n, nfree = left ? p.tiles_n : p.tiles_m, npair = ntri >> 1;
  int pb, fr;
  if (left) tile_of_dims(-1, npair, nfree, pid, pb, fr); else tile_of_dims(-1, nfree, npair, pid, fr, pb);
  const bool anchor0 = left ? !p.tri_eff_upper : (p.tri_eff_upper != 0);      // k-ranges [0, (b + 1) 128) ; otherwise [b 128, K)
  const bool maskA = left, maskB = !left;
  const bool keep_ge = left == (p.tri_eff_upper != 0);
  const int64_t ksa = AK ? kstep : (int64_t)kstep * p.lda, ksb = BKC ? kstep : (int64_t)kstep * p.ldb;
  const int64_t qsa = AK ? 32 * p.lda : TRANSPOSE_STAGE ? 8 * p.lda : (512 / TS) * p.lda;
  const int64_t qsa2 = AK ? 64 * p.lda : TRANSPOSE_STAGE ? 16 : 2 * (512 / TS) * p.lda;
  const int64_t qsb = BKC ? 32 * p.ldb : TRANSPOSE_STAGE ? 8 * p.ldb : (512 / TS) * p.ldb;
  const int64_t qsb2 = BKC ? 64 * p.ldb : TRANSPOSE_STAGE ? 16 : 2 * (512 / TS) * p.ldb;

  for (int ph = 0; ph < 2; ++ph) {
    const bool asc = (ph == 0) == anchor0;                 // phase 0 leaves the anchor, phase 1 returns to it
    const int bt = asc ? pb : ntri - 1 - pb;               // anchor 0: short tile pb, long tile ntri - 1 - pb; anchor K: the reverse
    const int ti = left ? bt : fr, tj = left ? fr : bt;
    const int i0 = ti * BM, j0 = tj * BN, d0 = left ? i0 : j0;
    const int klo = anchor0 ? 0 : d0, khi = anchor0 ? d0 + TS : p.K;
    const int ntk = (khi - klo) / kstep;
    const int tb0 = (d0 - klo) / BK, tb1 = tb0 + TS / BK;  // panels [tb0, tb1) cross the diagonal of the triangular operand

    d4_t acc[SUB][SUB];
#pragma unroll
    for (int a = 0; a < SUB; ++a)
#pragma unroll
      for (int c = 0; c < SUB; ++c) acc[a][c] = (d4_t){0.0, 0.0, 0.0, 0.0};

    // per-thread source of panel 0 (see panel_load for the thread -> element map)
    const double* const fa0 = AK ? p.A + (int64_t)(i0 + (tid >> 3)) * p.lda + klo + 2 * (tid & 7)
                              : TRANSPOSE_STAGE ? p.A + (int64_t)(klo + rc_k(tid, 0)) * p.lda + i0 + rc_r<TS>(tid, 0)
                                                : p.A + (int64_t)(klo + tid / (TS / 2)) * p.lda + i0 + 2 * (tid & (TS / 2 - 1));
    const double* const fb0 = BKC ? p.B + (int64_t)(j0 + (tid >> 3)) * p.ldb + klo + 2 * (tid & 7)
                              : TRANSPOSE_STAGE ? p.B + (int64_t)(klo + rc_k(tid, 0)) * p.ldb + j0 + rc_r<TS>(tid, 0)
                                                : p.B + (int64_t)(klo + tid / (TS / 2)) * p.ldb + j0 + 2 * (tid & (TS / 2 - 1));
    const double* fa = fa0;
    const double* fb = fb0;
    const int64_t sa = asc ? ksa : -ksa, sb = asc ? ksb : -ksb;
    const int dir = asc ? 1 : -1;
    int t = asc ? 0 : ntk - 1;

    d2_t ra[NQ], rb[NQ];
    {
      const int kf = klo + t * kstep;
      panel_load<TS, AK>(p.A, p.lda, i0, p.M, kf, khi, tid, 1, ra);
      panel_load<TS, BKC>(p.B, p.ldb, j0, p.N, kf, khi, tid, 1, rb);
      if (maskA && t >= tb0 && t < tb1) panel_mask<TS, AK>(i0, kf, tid, keep_ge, p.tri_unit, ra);
      if (maskB && t >= tb0 && t < tb1) panel_mask<TS, BKC>(j0, kf, tid, keep_ge, p.tri_unit, rb);
      panel_store<TS, AK>(lds, tid, ra);
      panel_store<TS, BKC>(lds + TILE_LDS, tid, rb);
    }
    __syncthreads();

    int par = 0;
    auto iterate = [&](const int tc, const int tnext, auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      const double* La = lds + par * STAGE_LDS;
      const double* Lb = La + TILE_LDS;
      const int kn = klo + tnext * kstep;
      const bool more = FAST || (tnext >= 0);
      if (FAST) {
        fa += sa;
        fb += sb;
        if (LOADS_IN_SHADOW) __builtin_amdgcn_s_setprio(1);
        panel_load_fast<TS>(fa, qsa, qsa2, ra);
        panel_load_fast<TS>(fb, qsb, qsb2, rb);
      } else if (more) {
        panel_load<TS, AK>(p.A, p.lda, i0, p.M, kn, khi, tid, 1, ra);
        panel_load<TS, BKC>(p.B, p.ldb, j0, p.N, kn, khi, tid, 1, rb);
      }
      unsigned keep = (1u << (SUB * SUB)) - 1u;
      if (!FAST) {
        const int kk = klo + tc * kstep;
        if (tc >= tb0 && tc < tb1 && !p.no_skip) {
#pragma unroll
          for (int a = 0; a < SUB; ++a)
#pragma unroll
            for (int c = 0; c < SUB; ++c) {
              bool on;
              if (left) {
                const int rlo = i0 + wm * (TS / 2) + 16 * a;
                on = p.tri_eff_upper ? (kk + BK - 1 >= rlo) : (kk <= rlo + 15);
              } else {
                const int clo = j0 + wn * (TS / 2) + 16 * c;
                on = p.tri_eff_upper ? (kk <= clo + 15) : (kk + BK - 1 >= clo);
              }
              if (!on) keep &= ~(1u << (a * SUB + c));
            }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        d2_t af[SUB], bf[SUB];
#pragma unroll
        for (int a = 0; a < SUB; ++a) af[a] = frag_read<TS, AK>(La, wm * (TS / 2) + a * 16 + r16, u, g);
#pragma unroll
        for (int c = 0; c < SUB; ++c) bf[c] = frag_read<TS, BKC>(Lb, wn * (TS / 2) + c * 16 + r16, u, g);
        if (!(FAST && LOADS_IN_SHADOW)) __builtin_amdgcn_s_setprio(1);
        mfma_step<SUB, 0, FAST>(acc, af, bf, keep);
        mfma_step<SUB, 1, FAST>(acc, af, bf, keep);
        if (FAST && STORE_IN_SHADOW && u == 1) {
          double* Na = lds + (par ^ 1) * STAGE_LDS;
          panel_store<TS, AK>(Na, tid, ra);
          panel_store<TS, BKC>(Na + TILE_LDS, tid, rb);
          if (LOADS_IN_SHADOW) {                                   // the pinned schedule of dgemm_tile_kernel's FAST iteration
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * SUB, 0);
#pragma unroll
            for (int i = 0; i < 2 * NQ; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, (SUB * SUB) / (2 * NQ), 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 2 * SUB; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, (SUB * SUB) / (2 * SUB), 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
#pragma unroll
          for (int i = 0; i < NW; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, (2 * SUB * SUB) / NW > 0 ? (2 * SUB * SUB) / NW : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
        if (!(FAST && LOADS_IN_SHADOW) || u == 1) __builtin_amdgcn_s_setprio(0);
      }
      if (more && !(FAST && STORE_IN_SHADOW)) {
        if (!FAST) {
          if (maskA && tnext >= tb0 && tnext < tb1) panel_mask<TS, AK>(i0, kn, tid, keep_ge, p.tri_unit, ra);
          if (maskB && tnext >= tb0 && tnext < tb1) panel_mask<TS, BKC>(j0, kn, tid, keep_ge, p.tri_unit, rb);
        }
        double* Na = lds + (par ^ 1) * STAGE_LDS;
        panel_store<TS, AK>(Na, tid, ra);
        panel_store<TS, BKC>(Na + TILE_LDS, tid, rb);
      }
      par ^= 1;
      __syncthreads();
    };
    // number of FAST iterations that can start at panel tc: the panel multiplied AND its successor clear of the diagonal band
    auto fast_run = [&](int tc) -> int {
      if (asc) return tc < tb0 ? max(0, min(tb0, ntk) - 1 - tc) : (tc >= tb1 ? max(0, ntk - 1 - tc) : 0);
      return tc >= tb1 ? tc - tb1 : (tc < tb0 ? tc : 0);
    };
    while (t >= 0 && t < ntk) {
      const int nf = fast_run(t);
      fa = fa0 + (int64_t)t * ksa;
      fb = fb0 + (int64_t)t * ksb;
      for (int q = 0; q < nf; ++q, t += dir) iterate(t, t + dir, std::true_type{});
      while (t >= 0 && t < ntk) {
        const int tn = t + dir;
        iterate(t, (tn >= 0 && tn < ntk) ? tn : -1, std::false_type{});
        t = tn;
        if (t >= 0 && t < ntk && fast_run(t) > 0) break;
      }
    }

    // epilogue (beta == 0): lane holds C[i = ..+r16][j = ..+g+4*reg]; 16 lanes -> 128 contiguous bytes of one column
#pragma unroll
    for (int a = 0; a < SUB; ++a) {
      const int i = i0 + wm * (TS / 2) + a * 16 + r16;
#pragma unroll
      for (int c = 0; c < SUB; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int j = j0 + wn * (TS / 2) + c * 16 + g + 4 * reg;
          p.C[i + (int64_t)j * p.ldc] = p.alpha * acc[a][c][reg];
        }
    }
  }
  stamp_end(p);
}

// C(part) <- alpha * sum_z slab[z] + beta*C   (fixed summation order: bit-reproducible)
__global__ void splitk_reduce_kernel(const double* __restrict__ slab, int64_t slab_ld, int64_t slab_stride, int splitk,
                                     double* __restrict__ C, int64_t ldc, int M, int N, double alpha, double beta, int out_uplo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  for (int j = blockIdx.y; j < N; j += gridDim.y) {
    if (out_uplo == CAPI_UPPER && i > j) continue;
    if (out_uplo == CAPI_LOWER && i < j) continue;
    double s = 0.0;
    for (int z = 0; z < splitk; ++z) s += slab[(int64_t)z * slab_stride + i + (int64_t)j * slab_ld];
    double* c = C + i + (int64_t)j * ldc;
    double r = alpha * s;
    if (beta != 0.0) r += beta * (*c);
    *c = r;
  }
}

// the same for many slices (the tall-skinny Gram matrix: one slice per CU): 64 rows x 4 slice-lanes per workgroup, each
// thread sums every 4th slice with 8 loads in flight, the four partial sums are combined through LDS in a fixed order
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const double* __restrict__ slab, int64_t slab_ld, int64_t slab_stride,
                                                                 int splitk, double* __restrict__ C, int64_t ldc, int M, int N,
                                                                 double alpha, double beta, int out_uplo) {
  __shared__ double part[4][64];
  const int il = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il, j = blockIdx.y;
  const bool live = i < M && !(out_uplo == CAPI_UPPER && i > j) && !(out_uplo == CAPI_LOWER && i < j);
  double s = 0.0;
  if (live) {
    const double* src = slab + i + (int64_t)j * slab_ld;
    int z = zl;
    for (; z + 28 < splitk; z += 32) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[(int64_t)(z + 4 * q) * slab_stride];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; z < splitk; z += 4) s += src[(int64_t)z * slab_stride];
  }
  part[zl][il] = s;
  __syncthreads();
  if (zl == 0 && live) {
    const double t = ((part[0][il] + part[1][il]) + part[2][il]) + part[3][il];
    double* c = C + i + (int64_t)j * ldc;
    double r = alpha * t;
    if (beta != 0.0) r += beta * (*c);
    *c = r;
  }
}

// scale-only path for alpha == 0 or K == 0:  C(part) <- beta*C
__global__ void scale_kernel(double* __restrict__ C, int64_t ldc, int M, int N, double beta, int out_uplo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  for (int j = blockIdx.y; j < N; j += gridDim.y) {
    if (out_uplo == CAPI_UPPER && i > j) continue;
    if (out_uplo == CAPI_LOWER && i < j) continue;
    double* c = C + i + (int64_t)j * ldc;
    *c = beta == 0.0 ? 0.0 : beta * (*c);
  }
}

typedef void (*gemm_kernel_t)(const GemmArgs);

// ---- latency-bound sizes: 32 x 32 output tile, K in chunks of 256 loaded in ONE burst ---------------------------
// The recursion's lower levels (orders 128 .. 1024) are chains of dependent products far too small to fill the chip; the
// tile kernel above then runs a handful of workgroups through K/16 iterations of (load latency + barrier) each.  Here a
// workgroup issues every load of a 256-deep chunk of both operand panels before anything else (32 x 16 bytes in flight
// per thread), stages them through LDS in four 64-deep quarters as they land, and its four waves multiply one 16 x 16
// MFMA tile each -- one exposed memory latency per 256 of K instead of one per 16, and 16x more workgroups per output.
constexpr int ST = 32, SKC = 256, SQK = 64, SLD = ST + 2;

// Issue only: unconditional 16-byte loads from clamped addresses.  Everything that depends on the loaded values (edge
// fix-ups, the triangle mask) lives in small_fix, called right before the quarter is staged -- a load inside a branch makes
// the compiler park an s_waitcnt vmcnt(0) at the join, and the "burst" of 32 loads then ran as 32 serialised round trips.
template <bool KC>
__device__ __forceinline__ void small_load(const double* __restrict__ X, int64_t ld, int r0, int R, int kc, int kend, int tid,
                                           bool vec_ok, int quarter, d2_t (&v)[16]) {
  // KC (k contiguous): row = tid>>3, k = kc + 2*(tid&7) + 16*q.   else (row contiguous): rows 2*(tid&15)+{0,1}, k = kc + (tid>>4) + 16*q
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int q = 4 * quarter + qq;
    int r[2], k[2];
    if (KC) { r[0] = r[1] = r0 + (tid >> 3); k[0] = kc + 2 * (tid & 7) + 16 * q; k[1] = k[0] + 1; }
    else { r[0] = r0 + 2 * (tid & 15); r[1] = r[0] + 1; k[0] = k[1] = kc + (tid >> 4) + 16 * q; }
    const double* ptr = KC ? X + (int64_t)r[0] * ld + k[0] : X + (int64_t)k[0] * ld + r[0];
    const bool in = r[1] < R && k[1] < kend && vec_ok;
    v[q] = *(const d2_t*)(in ? ptr : X);
  }
}

// interior, aligned chunk: one pointer, a constant stride between the 16 pieces, a piece-uniform range test -- nothing
// between the 32 load instructions but address increments
template <bool KC>
__device__ __forceinline__ void small_load_fast(const double* __restrict__ X, int64_t ld, int r0, int kc, int npieces, int tid,
                                                int quarter, d2_t (&v)[16]) {
  const double* ptr = KC ? X + (int64_t)(r0 + (tid >> 3)) * ld + kc + 2 * (tid & 7) : X + (int64_t)(kc + (tid >> 4)) * ld + r0 + 2 * (tid & 15);
  const int64_t sq = KC ? 16 : 16 * ld;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int q = 4 * quarter + qq;
    v[q] = q < npieces ? *(const d2_t*)(ptr + q * sq) : (d2_t){0.0, 0.0};
  }
}

template <bool KC>
__device__ __forceinline__ void small_fix(const double* __restrict__ X, int64_t ld, int r0, int R, int kc, int kend, int tid,
                                          bool vec_ok, bool tri, bool keep_ge, bool unit, int quarter, d2_t (&v)[16]) {
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int q = 4 * quarter + qq;
    int r[2], k[2];
    if (KC) { r[0] = r[1] = r0 + (tid >> 3); k[0] = kc + 2 * (tid & 7) + 16 * q; k[1] = k[0] + 1; }
    else { r[0] = r0 + 2 * (tid & 15); r[1] = r[0] + 1; k[0] = k[1] = kc + (tid >> 4) + 16 * q; }
    const bool in = r[1] < R && k[1] < kend && vec_ok;
    d2_t val = v[q];
    if (!in) {                                             // ragged edge / unaligned operand: scalar reads
      const double* ptr = KC ? X + (int64_t)r[0] * ld + k[0] : X + (int64_t)k[0] * ld + r[0];
      val = (d2_t){0.0, 0.0};
      if (r[0] < R && k[0] < kend) val.x = ptr[0];
      if (r[1] < R && k[1] < kend) val.y = ptr[1];
    }
    // only the pieces whose 16 k's overlap the tile's own 32 rows can straddle the diagonal (the tile's k-range was cut at
    // the triangle's edge by the caller); the test is uniform per piece
    const int kq = kc + 16 * q;
    if (tri && kq < r0 + ST && kq + 16 > r0) {
      if (keep_ge ? (k[0] < r[0]) : (k[0] > r[0])) val.x = 0.0;
      if (keep_ge ? (k[1] < r[1]) : (k[1] > r[1])) val.y = 0.0;
      if (unit) {
        if (k[0] == r[0]) val.x = 1.0;
        if (k[1] == r[1]) val.y = 1.0;
      }
    }
    v[q] = val;
  }
}

template <bool KC>
__device__ __forceinline__ void small_store(double* __restrict__ L, int tid, int quarter, const d2_t (&v)[16]) {
  // L[k][r], k relative to the chunk
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int q = 4 * quarter + qq;
    if (KC) {
      const int k = 2 * (tid & 7) + 16 * q, r = tid >> 3;
      L[k * SLD + r] = v[q].x;
      L[(k + 1) * SLD + r] = v[q].y;
    } else {
      const int k = (tid >> 4) + 16 * q, r = 2 * (tid & 15);
      *(d2_t*)&L[k * SLD + r] = v[q];
    }
  }
}

template <bool AK, bool BKC>
__global__ __launch_bounds__(256, 1) void dgemm_small_kernel(const GemmArgs pin) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  GemmArgs p = pin;
  if (p.batch > 1) {                           // strided batch: same shapes, operands a fixed stride apart
    const int64_t bz = blockIdx.y;
    p.A += bz * p.sa; p.B += bz * p.sb; p.C += bz * p.sc;
  }
  double* La = lds;
  double* Lb = lds + p.ts * SLD;               // p.ts carries the staged depth here: min(256, K rounded up to 64) rows per operand
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4, wm = wave & 1, wn = wave >> 1;
  const int ti = blockIdx.x % p.tiles_m, tj = blockIdx.x / p.tiles_m;
  const int i0 = ti * ST, j0 = tj * ST;
  if (p.out_uplo == CAPI_UPPER && i0 > j0 + ST - 1) return;
  if (p.out_uplo == CAPI_LOWER && j0 > i0 + ST - 1) return;
  int klo = 0, khi = p.K;
  if (p.tri_side == CAPI_LEFT) {
    if (p.tri_eff_upper) klo = i0; else khi = min(p.K, i0 + ST);
  } else if (p.tri_side == CAPI_RIGHT) {
    if (p.tri_eff_upper) khi = min(p.K, j0 + ST); else klo = j0;
  }
  klo &= ~1;                                               // keep 16-byte alignment of k-contiguous loads
  const bool keep_ge = (p.tri_side == CAPI_LEFT) == (p.tri_eff_upper != 0);
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int kc = klo; kc < khi; kc += SKC) {
    d2_t va[16], vb[16];
    // interior chunk (the recursion's case): straight-line loads; ragged or unaligned: clamped addresses + fix-ups
    const int span = khi - kc < SKC ? khi - kc : SKC;
    const bool fast = p.a_vec && p.b_vec && i0 + ST <= p.M && j0 + ST <= p.N && (span & 15) == 0;
    if (fast) {
#pragma unroll
      for (int quarter = 0; quarter < 4; ++quarter) {      // issue order = consumption order (vmcnt counts in order)
        small_load_fast<AK>(p.A, p.lda, i0, kc, span >> 4, tid, quarter, va);
        small_load_fast<BKC>(p.B, p.ldb, j0, kc, span >> 4, tid, quarter, vb);
      }
    } else {
#pragma unroll
      for (int quarter = 0; quarter < 4; ++quarter) {
        small_load<AK>(p.A, p.lda, i0, p.M, kc, khi, tid, p.a_vec, quarter, va);
        small_load<BKC>(p.B, p.ldb, j0, p.N, kc, khi, tid, p.b_vec, quarter, vb);
      }
    }
    if (kc > klo) __syncthreads();                         // the previous chunk's fragments have been consumed
    const int nq = min(4, (khi - kc + SQK - 1) / SQK);
#pragma unroll
    for (int quarter = 0; quarter < 4; ++quarter) {
      if (quarter >= nq) break;
      if (!fast || p.tri_side >= 0) {
        // (on the fast path vec_ok = true and full ranges make every piece "in": only the triangle mask acts)
        small_fix<AK>(p.A, p.lda, i0, fast ? i0 + ST : p.M, kc, fast ? kc + SKC : khi, tid, p.a_vec, p.tri_side == CAPI_LEFT, keep_ge, p.tri_unit, quarter, va);
        small_fix<BKC>(p.B, p.ldb, j0, fast ? j0 + ST : p.N, kc, fast ? kc + SKC : khi, tid, p.b_vec, p.tri_side == CAPI_RIGHT, keep_ge, p.tri_unit, quarter, vb);
      }
      small_store<AK>(La, tid, quarter, va);
      small_store<BKC>(Lb, tid, quarter, vb);
      __syncthreads();
#pragma unroll
      for (int s4 = 0; s4 < SQK / 4; ++s4) {
        const int k = quarter * SQK + 4 * s4 + g;
        const double a = La[k * SLD + wm * 16 + r16];
        const double b = Lb[k * SLD + wn * 16 + r16];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);
      }
    }
  }
  const int i = i0 + wm * 16 + r16;
  double old[4];
  bool okv[4];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int j = j0 + wn * 16 + g + 4 * reg;
    bool ok = (i < p.M) && (j < p.N);
    if (p.out_uplo == CAPI_UPPER) ok = ok && (i <= j);
    if (p.out_uplo == CAPI_LOWER) ok = ok && (i >= j);
    okv[reg] = ok;
    old[reg] = (p.beta != 0.0) ? *(ok ? p.C + i + (int64_t)j * p.ldc : p.C) : 0.0;   // beta uniform: loads issued together
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int j = j0 + wn * 16 + g + 4 * reg;
    if (okv[reg]) p.C[i + (int64_t)j * p.ldc] = p.alpha * acc[reg] + (p.beta != 0.0 ? p.beta * old[reg] : 0.0);
  }
}

gemm_kernel_t pick_small(bool ak, bool bkc) {
  return ak ? (bkc ? dgemm_small_kernel<true, true> : dgemm_small_kernel<true, false>)
            : (bkc ? dgemm_small_kernel<false, true> : dgemm_small_kernel<false, false>);
}



// ---- tall-skinny operands (CholeskyQR2: m x n with n <= 256 and m in the millions) --------------------------------
// With n this small a square tiling re-reads the tall operand once per tile column (Gram: 3 x 128-tiles = 3 passes over
// 8.6 GB, Q = A R^-1 with 64-tiles: 2.5 passes) and that, not the matrix pipe, sets the time.  Here one workgroup of 8
// waves covers the FULL width, so every element of the tall operand travels global -> LDS exactly once.
constexpr int TSK_THREADS = 512, TSK_W = 256;    // up to 16 column strips of 16
// the tall operands are read once and written once: non-temporal accesses (CAPI_TS_NT=0 compiles the plain ones, A/B)
#ifndef CAPI_TS_NT
#define CAPI_TS_NT 1
#endif
#if CAPI_TS_NT
#define TS_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define TS_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define TS_STORE(ptr, val) (*(ptr) = (val))
#define TS_LOAD(ptr) (*(ptr))
#endif

// Gram matrix, upper triangle:  slab[z] (or C) = sum over this workgroup's k-range of A[k][i] A[k][j], A k-contiguous.
// The 16 x 16 grid of output tiles has 136 upper tiles; wave w owns tile-rows w and 15-w (17 tiles): balanced, and per
// k-step it reads 2 row fragments and the <= 16 column fragments right of its shorter row.
// FULLW: all 16 column strips exist (n == 256): no per-tile liveness test, i.e. no branch between MFMAs
template <int W, bool FULLW>
__device__ __forceinline__ void gram_ts_body(const GemmArgs& p, double* __restrict__ lds) {
  constexpr int STAGE = TSK_W * SK;                        // [col][k], k-contiguous + 2 pad (as the tile kernel's Lk)
  constexpr int RA = W, RB = 15 - W, NA = 16 - RA;         // tile-rows of this wave; row RA holds NA tiles, row RB holds W + 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int N = p.N, nct = (N + 15) >> 4;
  const int z = blockIdx.x, S = gridDim.x;
  const int P = (p.K + BK - 1) / BK;                       // panels of the whole k-range
  const int pb = (int)((int64_t)z * P / S), pe = (int)((int64_t)(z + 1) * P / S), np = pe - pb;
  const int rot = np > 1 ? (int)((int64_t)z * np / S) : 0; // staggered start: resident slices do not sweep DRAM channels in step
  d4_t acc[17];
#pragma unroll
  for (int q = 0; q < 17; ++q) acc[q] = (d4_t){0.0, 0.0, 0.0, 0.0};

  const int kp = tid & 7, c0 = tid >> 3;                   // this thread's pieces: columns c0 + 64 q, k = 2 kp (+1)
  // operand addressing: column-major (k + c lda) or panel32 (32-row tiles: a 16-deep panel is half a tile)
  const int64_t acs = p.a_tiled ? 32 : p.lda;              // column stride
  auto poff = [&](int pnl) -> int64_t { return p.a_tiled ? (int64_t)(pnl >> 1) * (32 * TSK_W) + (pnl & 1) * 16 : (int64_t)pnl * BK; };
  // load() only ISSUES the loads (unconditional, from a clamped address); every select on their results waits until
  // stage(), after the MFMA phase -- a select right behind the load would make the wave wait for the prefetch up front
  auto load = [&](int pnl, d2_t (&st)[4]) {
    const int k = pnl * BK + 2 * kp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + 64 * q;
      const bool in = c < N && k + 1 < p.K && p.a_vec;
      st[q] = *(const d2_t*)(in ? p.A + (int64_t)c * acs + poff(pnl) + 2 * kp : p.A);
    }
  };
  auto stage = [&](double* L, int pnl, const d2_t (&st)[4]) {
    // workgroup-uniform fast path: no branch (and so no conservative s_waitcnt vmcnt(0) at a join) around the LDS stores
    if (FULLW && p.a_vec && (pnl + 1) * BK <= p.K) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *(d2_t*)&L[(c0 + 64 * q) * SK + 2 * kp] = st[q];
      return;
    }
    const int k = pnl * BK + 2 * kp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + 64 * q;
      const bool in = c < N && k + 1 < p.K && p.a_vec;
      d2_t v = st[q];
      if (!in) {                                            // ragged edge / unaligned operand: scalar reads, rare
        v = (d2_t){0.0, 0.0};
        if (c < N) {
          const double* ptr = p.A + (int64_t)c * acs + poff(pnl) + 2 * kp;
          if (k < p.K) v.x = ptr[0];
          if (k + 1 < p.K) v.y = ptr[1];
        }
      }
      *(d2_t*)&L[c * SK + 2 * kp] = v;
    }
  };
  auto pidx = [&](int i) { int q = rot + i; if (q >= np) q -= np; return pb + q; };
  // Two panels are in flight in registers beyond the one being multiplied: with one workgroup per CU the HBM latency
  // under this access pattern (256 streams 32 MB apart) exceeds one iteration, and one panel ahead left ~35 % exposed.
  d2_t stA[4], stB[4];
  int par = 0;
  // STEADY (all 16 column strips, aligned A, every panel whole, two more panels to come): the iteration has no branch.
  // With the prefetch inside `if (it + 2 < np)` the compiler put an s_waitcnt vmcnt(0) at the join right behind it, i.e.
  // every panel waited for the loads it had just issued -- the two-panel prefetch was synchronous in effect.
  auto step = [&](int it, d2_t (&cur)[4], d2_t (&nw)[4], auto steady_tag) {   // cur holds panel it+1, nw receives panel it+2
    constexpr bool STEADY = decltype(steady_tag)::value;
    const double* L = lds + par * STAGE;
    if (STEADY) {
      const double* src = p.A + (int64_t)c0 * acs + poff(pidx(it + 2)) + 2 * kp;
#pragma unroll
      for (int q = 0; q < 4; ++q) nw[q] = TS_LOAD((const d2_t*)(src + (int64_t)(64 * q) * acs));
    } else if (it + 2 < np) load(pidx(it + 2), nw);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      d2_t bf[16];
#pragma unroll
      for (int j = RA; j < 16; ++j) bf[j] = *(const d2_t*)&L[(16 * j + r16) * SK + 8 * u + 2 * g];
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // row RA: tiles (RA, j), j = RA..15 -> acc[j - RA];  row RB: tiles (RB, j), j = RB..15 -> acc[NA + j - RB]
#pragma unroll
        for (int j = RA; j < 16; ++j)
          if (FULLW || j < nct) acc[j - RA] = __builtin_amdgcn_mfma_f64_16x16x4f64(e ? bf[j].y : bf[j].x, e ? bf[RA].y : bf[RA].x, acc[j - RA], 0, 0, 0);
#pragma unroll
        for (int j = RB; j < 16; ++j)
          if (FULLW || j < nct) acc[NA + j - RB] = __builtin_amdgcn_mfma_f64_16x16x4f64(e ? bf[j].y : bf[j].x, e ? bf[RB].y : bf[RB].x, acc[NA + j - RB], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if (STEADY) {
      double* Ln = lds + (par ^ 1) * STAGE;
#pragma unroll
      for (int q = 0; q < 4; ++q) *(d2_t*)&Ln[(c0 + 64 * q) * SK + 2 * kp] = cur[q];
    } else if (it + 1 < np) stage(lds + (par ^ 1) * STAGE, pidx(it + 1), cur);
    par ^= 1;
    __syncthreads();
  };
  if (np > 0) {
    load(pidx(0), stA);
    if (np > 1) load(pidx(1), stB);
    stage(lds, pidx(0), stA);
  }
  __syncthreads();
  int it = 0;
  if (FULLW && p.a_vec && p.K % BK == 0) {
    for (; it + 3 < np; it += 2) {                          // both steps have panel it + 2 (+1) to prefetch
      step(it, stB, stA, std::true_type{});
      step(it + 1, stA, stB, std::true_type{});
    }
  }
  for (; it < np; it += 2) {
    step(it, stB, stA, std::false_type{});
    if (it + 1 < np) step(it + 1, stA, stB, std::false_type{});
  }
  // epilogue: lane holds (i = 16 row + r16, j = 16 col + g + 4 reg); partial sums go to slab z, or straight to C
  auto put = [&](int trow, int tcol, const d4_t& v) {
    const int i = 16 * trow + r16;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = 16 * tcol + g + 4 * reg;
      if (i < N && j < N && i <= j) {
        if (p.splitk > 1) p.slab[(int64_t)z * p.slab_stride + i + (int64_t)j * p.slab_ld] = v[reg];
        else {
          double* c = p.C + i + (int64_t)j * p.ldc;
          double r = p.alpha * v[reg];
          if (p.beta != 0.0) r += p.beta * (*c);
          *c = r;
        }
      }
    }
  };
#pragma unroll
  for (int j = RA; j < 16; ++j) if (j < nct) put(RA, j, acc[j - RA]);
#pragma unroll
  for (int j = RB; j < 16; ++j) if (j < nct) put(RB, j, acc[NA + j - RB]);
}

__global__ __launch_bounds__(TSK_THREADS, 1) void gram_ts_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // the tile-to-wave map is baked into eight instantiations (accumulators must be indexed at compile time); all of them
  // execute the same sequence of barriers
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (p.N == TSK_W) {
    switch (w) {
      case 0: gram_ts_body<0, true>(p, lds); break;
      case 1: gram_ts_body<1, true>(p, lds); break;
      case 2: gram_ts_body<2, true>(p, lds); break;
      case 3: gram_ts_body<3, true>(p, lds); break;
      case 4: gram_ts_body<4, true>(p, lds); break;
      case 5: gram_ts_body<5, true>(p, lds); break;
      case 6: gram_ts_body<6, true>(p, lds); break;
      default: gram_ts_body<7, true>(p, lds); break;
    }
  } else {
    switch (w) {
      case 0: gram_ts_body<0, false>(p, lds); break;
      case 1: gram_ts_body<1, false>(p, lds); break;
      case 2: gram_ts_body<2, false>(p, lds); break;
      case 3: gram_ts_body<3, false>(p, lds); break;
      case 4: gram_ts_body<4, false>(p, lds); break;
      case 5: gram_ts_body<5, false>(p, lds); break;
      case 6: gram_ts_body<6, false>(p, lds); break;
      default: gram_ts_body<7, false>(p, lds); break;
    }
  }
}

// Q = alpha * A * T (+ beta * C) for a tall A (M x 256, row-contiguous) and a 256 x 256 UPPER triangular T.
// T-stationary: wave w of the 8 keeps the MFMA fragments of output strips w and 15-w (16 columns each; 4 (w+1) + 4 (16-w)
// = 68 k-steps, the same for every wave) in registers for the whole kernel.  A streams through LDS in 16-row tiles
// (32 KB, read from HBM exactly once, two tiles in flight in registers beyond the one being multiplied); per tile a wave
// reads each A fragment once and issues its 68 MFMAs with no barrier in between; each wave stores its own two strips.
// Work per tile is uniform, so the one barrier per tile (8704 MFMA cycles) costs what it costs in the Gram kernel.
// (A row-split variant that staged T's panels through LDS instead spent its time on T reloads, on short late panels and
//  on `s_waitcnt vmcnt(0)` forced by mixing outstanding stores and loads: 8-9.6 ms against 7.6 ms for the tile kernel.)
template <int W>
__device__ __forceinline__ void trmm_ts_body(const GemmArgs& p, double* __restrict__ lds) {
  constexpr int SA = W, SB = 15 - W, NA = 4 * (SA + 1), NB = 4 * (SB + 1);   // strips and their k-step counts (NA <= NB)
  constexpr int TILE = 256 * 16;                           // LDS tile [k][16 rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntile = (p.M + 15) >> 4;
  // stationary T fragments: k-step s supplies T[4 s + g][16 strip + r16] (zero below the diagonal, unit diagonal on request)
  double ta[NA], tb[NB];
  auto tget = [&](int k, int j) {
    double v = p.B[k + (int64_t)j * p.ldb];
    if (k > j) v = 0.0;
    if (p.tri_unit && k == j) v = 1.0;
    return v;
  };
#pragma unroll
  for (int s_ = 0; s_ < NA; ++s_) ta[s_] = tget(4 * s_ + g, 16 * SA + r16);
#pragma unroll
  for (int s_ = 0; s_ < NB; ++s_) tb[s_] = tget(4 * s_ + g, 16 * SB + r16);

  // A tile -> registers: thread t owns rows 2 (t & 7), +1 of columns (t >> 3) + 64 q: the 8 lanes of a group fetch one
  // whole 128-byte line (the tile's 16 rows of one column), a wave instruction 8 whole lines.  (A first mapping gave each
  // thread 64 contiguous bytes: every instruction then touched 32 lines, 32 bytes of each, four times over.)
  const int lc = tid >> 3, lr = 2 * (tid & 7);
  auto load = [&](int tile, d2_t (&st)[4]) {
    const bool in = 16 * tile + 16 <= p.M && p.a_vec;
    const double* src = in ? p.A + 16 * tile + lr + (int64_t)lc * p.lda : p.A;
    const int64_t cs = in ? 64 * p.lda : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) st[q] = *(const d2_t*)(src + q * cs);
  };
  auto stage = [&](double* L, int tile, const d2_t (&st)[4]) {
    // workgroup-uniform fast path: no branch (and so no conservative s_waitcnt vmcnt(0) at a join) around the LDS stores
    if (p.a_vec && 16 * tile + 16 <= p.M) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *(d2_t*)&L[(lc + 64 * q) * 16 + lr] = st[q];
      return;
    }
    const int r = 16 * tile + lr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                           // ragged last tile / unaligned A: scalar reads
      const int64_t c = lc + 64 * q;
      d2_t v = {0.0, 0.0};
      if (r < p.M) v.x = p.A[r + c * p.lda];
      if (r + 1 < p.M) v.y = p.A[r + 1 + c * p.lda];
      *(d2_t*)&L[c * 16 + lr] = v;
    }
  };
  const int t0 = blockIdx.x, dt = gridDim.x;
  if (t0 >= ntile) return;
  d2_t stA[4], stB[4];
  int par = 0;
  // STEADY: the two tiles ahead exist and are full, A is 16-byte aligned and beta == 0 -- the iteration then has no branch
  // at all.  That matters beyond the branch itself: a load or store inside a branch makes the compiler wait with
  // s_waitcnt vmcnt(0) at the next use of ANY loaded value, i.e. each staging waited for the prefetch issued 68 MFMAs
  // earlier AND for the previous tile's eight stores (measured: stores 1.8 ms, loads 1.3 ms of a 6.3 ms kernel whose
  // MFMAs take 4.1 ms).  Straight-line, the wait before staging is vmcnt(12): stores and the newest prefetch stay in flight.
  auto step = [&](int tile, d2_t (&cur)[4], d2_t (&nw)[4], auto steady_tag) {   // cur holds tile + dt, nw receives tile + 2 dt
    constexpr bool STEADY = decltype(steady_tag)::value;
    const double* L = lds + par * TILE;
    if (STEADY) {
      const double* src = p.A + 16 * (tile + 2 * dt) + lr + (int64_t)lc * p.lda;
#pragma unroll
      for (int q = 0; q < 4; ++q) nw[q] = *(const d2_t*)(src + q * 64 * p.lda);
    } else if (tile + 2 * dt < ntile) load(tile + 2 * dt, nw);
    d4_t ca = {0.0, 0.0, 0.0, 0.0}, cb = {0.0, 0.0, 0.0, 0.0};
    const double* la = L + g * 16 + r16;                    // A[row r16][k = 4 s + g]
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s_ = 0; s_ < NB; ++s_) {
      const double af = la[64 * s_];
      if (s_ < NA) ca = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[s_], af, ca, 0, 0, 0);
      cb = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[s_], af, cb, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (STEADY) {
      double* Ln = lds + (par ^ 1) * TILE;
#pragma unroll
      for (int q = 0; q < 4; ++q) *(d2_t*)&Ln[(lc + 64 * q) * 16 + lr] = cur[q];
    } else if (tile + dt < ntile) stage(lds + (par ^ 1) * TILE, tile + dt, cur);
    par ^= 1;
    __syncthreads();
    // lane holds (i = row r16 of the tile, j = 16 strip + g + 4 reg): 16 lanes -> 128 contiguous bytes of one column
    // (beta is tested ONCE: a per-store `if (beta != 0) r += beta * *c` puts a load in a branch before every store, and
    //  the compiler then parks an s_waitcnt vmcnt(0) at each join -- eight full drains of the prefetch queue per tile)
    const int i = 16 * tile + r16;
    if (STEADY || i < p.M) {
      double* c0_ = p.C + i + (int64_t)(16 * SA + g) * p.ldc;
      double* c1_ = p.C + i + (int64_t)(16 * SB + g) * p.ldc;
      const int64_t s4 = 4 * p.ldc;
      if (STEADY || p.beta == 0.0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { c0_[reg * s4] = p.alpha * ca[reg]; c1_[reg * s4] = p.alpha * cb[reg]; }
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          c0_[reg * s4] = p.alpha * ca[reg] + p.beta * c0_[reg * s4];
          c1_[reg * s4] = p.alpha * cb[reg] + p.beta * c1_[reg * s4];
        }
      }
    }
  };
  load(t0, stA);
  if (t0 + dt < ntile) load(t0 + dt, stB);
  stage(lds, t0, stA);
  __syncthreads();
  int tile = t0;
  if (p.a_vec && p.beta == 0.0) {
    const int nfull = p.M >> 4;                             // tiles with all 16 rows
    for (; tile + 3 * dt < nfull; tile += 2 * dt) {         // both steps see tile + 2 dt full
      step(tile, stB, stA, std::true_type{});
      step(tile + dt, stA, stB, std::true_type{});
    }
  }
  for (; tile < ntile; tile += 2 * dt) {
    step(tile, stB, stA, std::false_type{});
    if (tile + dt < ntile) step(tile + dt, stA, stB, std::false_type{});
  }
}

// The same T-stationary scheme on 32-row tiles (two 16-row halves, each laid out [k][16 rows] in LDS, 2 x 64 KB): every T
// fragment feeds two MFMAs, a wave runs four accumulator chains instead of two, there is one barrier per 136 MFMAs, and a
// tile's 256 bytes of one column (= one address translation; the 256 columns are 256 pages 8 lda bytes apart) are fetched
// by one 16-lane group.  One tile is in flight in registers (its loads have the 136 MFMAs of the current tile to land).
template <int W>
__device__ __forceinline__ void trmm_ts32_body(const GemmArgs& p, double* __restrict__ lds) {
  constexpr int SA = W, SB = 15 - W, NA = 4 * (SA + 1), NB = 4 * (SB + 1);   // strips and their k-step counts (NA <= NB)
  constexpr int HALF = 256 * 16, TILE = 2 * HALF;          // LDS tile: [half][k][16 rows]; half 0 = the tile's EVEN rows, half 1 = its odd rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntile = (p.M + 31) >> 5;
  double ta[NA], tb[NB];
  // (alpha is folded into the stationary fragments: nothing is scaled on the way out; alpha == 1 -- CholeskyQR2 -- is exact either way)
  auto tget = [&](int k, int j) {
    double v = p.B[k + (int64_t)j * p.ldb];
    if (k > j) v = 0.0;
    if (p.tri_unit && k == j) v = 1.0;
    return p.beta == 0.0 ? p.alpha * v : v;
  };
#pragma unroll
  for (int s_ = 0; s_ < NA; ++s_) ta[s_] = tget(4 * s_ + g, 16 * SA + r16);
#pragma unroll
  for (int s_ = 0; s_ < NB; ++s_) tb[s_] = tget(4 * s_ + g, 16 * SB + r16);

  // thread t owns rows 2 (t & 15), +1 of columns (t >> 4) + 32 q: 16 lanes fetch the tile's 256 bytes of one column
  const int lc = tid >> 4, lr = 2 * (tid & 15);
  // operand / output addressing: column-major (row + col ld) or panel32 (tile t at 32 * 256 * t, column stride 32)
  const int64_t acs = p.a_tiled ? 32 : p.lda, ats = p.a_tiled ? 32 * 256 : 32;     // A: column stride, tile stride
  const int64_t ccs = p.c_tiled ? 32 : p.ldc, cts = p.c_tiled ? 32 * 256 : 32;     // C likewise
  // Rows 2i and 2i + 1 of a tile go to the two HALVES of its LDS image (index i in each): MFMA half h then computes rows 2 r16 + h, so a
  // lane ends up with two CONSECUTIVE rows of every column it holds -- 16-byte stores, whole 256-byte column segments per store
  // instruction, and no lane exchange on the way out (round 2 kept rows 0-15 / 16-31 in the halves and traded registers across lane
  // pairs with DPP: 16 moves, 48 selects and 16 multiplies per tile and wave between two MFMA phases).
  const int lofs = lc * 16 + (lr >> 1);                     // LDS offset (in either half) of row pair (lr, lr + 1) of column lc
  auto load = [&](int tile, d2_t (&st)[8]) {
    const bool in = 32 * tile + 32 <= p.M && p.a_vec;
    const double* src = in ? p.A + ats * tile + lr + (int64_t)lc * acs : p.A;
    const int64_t cs = in ? 32 * acs : 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) st[q] = *(const d2_t*)(src + q * cs);
  };
  auto stage = [&](double* L, int tile, const d2_t (&st)[8]) {
    if (p.a_vec && 32 * tile + 32 <= p.M) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { L[lofs + 512 * q] = st[q].x; L[HALF + lofs + 512 * q] = st[q].y; }
      return;
    }
    const int r = 32 * tile + lr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {                           // ragged last tile / unaligned A: scalar reads
      const int64_t c = lc + 32 * q;
      d2_t v = {0.0, 0.0};
      if (r < p.M) v.x = p.A[ats * tile + lr + c * acs];
      if (r + 1 < p.M) v.y = p.A[ats * tile + lr + 1 + c * acs];
      L[lofs + 512 * q] = v.x; L[HALF + lofs + 512 * q] = v.y;
    }
  };
  const int t0 = blockIdx.x, dt = gridDim.x;
  if (t0 >= ntile) return;
  d2_t st[8];
  int par = 0;
  // STEADY: the next tile exists and is full, A is 16-byte aligned, beta == 0: an iteration without a single branch
  // (a load or store inside a branch costs an s_waitcnt vmcnt(0) at the next use of any loaded value)
  auto step = [&](int tile, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    const double* L = lds + par * TILE;
    if (STEADY) {
      const double* src = p.A + ats * (tile + dt) + lr + (int64_t)lc * acs;
#pragma unroll
      for (int q = 0; q < 8; ++q) st[q] = TS_LOAD((const d2_t*)(src + q * 32 * acs));
    } else if (tile + dt < ntile) load(tile + dt, st);
    d4_t ca0 = {0.0, 0.0, 0.0, 0.0}, ca1 = ca0, cb0 = ca0, cb1 = ca0;
    const double* la = L + g * 16 + r16;                    // A[row 2 r16 (+1)][k = 4 s + g]
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s_ = 0; s_ < NB; ++s_) {
      const double a0 = la[64 * s_], a1 = la[HALF + 64 * s_];
      if (s_ < NA) {
        ca0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[s_], a0, ca0, 0, 0, 0);
        ca1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[s_], a1, ca1, 0, 0, 0);
      }
      cb0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[s_], a0, cb0, 0, 0, 0);
      cb1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[s_], a1, cb1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (STEADY) {
      // (spreading these eight LDS stores over the last eight k-steps' MFMAs instead -- the loads landed a whole phase earlier -- was
      //  measured, build against build: p32->p32 4.64 -> 4.54 ms, cm->cm 4.99 -> 5.07, CholeskyQR2 18.99 -> 18.95 ms: nothing; r3k)
      double* Ln = lds + (par ^ 1) * TILE;
#pragma unroll
      for (int q = 0; q < 8; ++q) { Ln[lofs + 512 * q] = st[q].x; Ln[HALF + lofs + 512 * q] = st[q].y; }
    } else if (tile + dt < ntile) stage(lds + (par ^ 1) * TILE, tile + dt, st);
    par ^= 1;
    // (The reverse order -- stage, fetch TWO tiles ahead, store, and only then the barrier, so that nothing but the first LDS reads stands
    //  between the barrier and the MFMAs -- was measured build against build: 5.40 / 5.17 / 5.31 / 5.02 ms against 4.92 / 4.66 / 4.87 / 4.55
    //  (cm->cm / cm->p32 / p32->cm / p32->p32), CholeskyQR2 19.44 against 19.11 ms: slower, like the LDS-DMA form, which shares the
    //  two-tile fetch distance.  profiles/r3z_*.)
    __syncthreads();
    // lane holds rows 2 r16, 2 r16 + 1 of columns j = 16 strip + g + 4 reg: one 16-byte store per column, 16 lanes -> the tile's whole
    // 256-byte segment of that column
    const int i = 32 * tile + 2 * r16;
    double* c0_ = p.C + cts * tile + 2 * r16 + (int64_t)(16 * SA + g) * ccs;
    double* c1_ = p.C + cts * tile + 2 * r16 + (int64_t)(16 * SB + g) * ccs;
    const int64_t s4 = 4 * ccs;
    if (STEADY) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        TS_STORE((d2_t*)(c0_ + reg * s4), ((d2_t){ca0[reg], ca1[reg]}));
        TS_STORE((d2_t*)(c1_ + reg * s4), ((d2_t){cb0[reg], cb1[reg]}));
      }
    } else {
      const bool ok0 = i < p.M, ok1 = i + 1 < p.M;
      if (p.beta == 0.0) {                                     // (alpha already sits in the fragments)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          if (ok0) { c0_[reg * s4] = ca0[reg]; c1_[reg * s4] = cb0[reg]; }
          if (ok1) { c0_[reg * s4 + 1] = ca1[reg]; c1_[reg * s4 + 1] = cb1[reg]; }
        }
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          if (ok0) { c0_[reg * s4] = p.alpha * ca0[reg] + p.beta * c0_[reg * s4]; c1_[reg * s4] = p.alpha * cb0[reg] + p.beta * c1_[reg * s4]; }
          if (ok1) {
            c0_[reg * s4 + 1] = p.alpha * ca1[reg] + p.beta * c0_[reg * s4 + 1];
            c1_[reg * s4 + 1] = p.alpha * cb1[reg] + p.beta * c1_[reg * s4 + 1];
          }
        }
      }
    }
  };
  load(t0, st);
  stage(lds, t0, st);
  __syncthreads();
  int tile = t0;
  if (p.a_vec && p.beta == 0.0 && (((uintptr_t)p.C & 15) == 0) && ((p.ldc & 1) == 0)) {   // (16-byte stores of row pairs)
    const int nfull = p.M >> 5;                             // tiles with all 32 rows
    for (; tile + dt < nfull; tile += dt) step(tile, std::true_type{});
  }
  for (; tile < ntile; tile += dt) step(tile, std::false_type{});
}

__global__ __launch_bounds__(TSK_THREADS, 1) void trmm_right_ts32_kernel(const GemmArgs p) {   // N == K == 256
  extern __shared__ __attribute__((aligned(16))) double lds[];
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: trmm_ts32_body<0>(p, lds); break;
    case 1: trmm_ts32_body<1>(p, lds); break;
    case 2: trmm_ts32_body<2>(p, lds); break;
    case 3: trmm_ts32_body<3>(p, lds); break;
    case 4: trmm_ts32_body<4>(p, lds); break;
    case 5: trmm_ts32_body<5>(p, lds); break;
    case 6: trmm_ts32_body<6>(p, lds); break;
    default: trmm_ts32_body<7>(p, lds); break;
  }
}

// (LDS-DMA forms of this kernel -- global_load_lds_dwordx4, no staging registers, no ds_write pass -- were built twice and measured slower
//  both times.  Round 1: DMA one tile ahead, the previous tile's stores spread over the MFMA loop: 5.95 against 5.21 ms.  Round 3: the fetch
//  issued from inline assembly (through the builtin the compiler, unable to tell the two buffers apart, waits with vmcnt(0) for the fetch it
//  has just issued), TWO tiles ahead on the two buffers -- MFMA(t) | vmcnt(0) | barrier | DMA(t + 2) | stores(t) --, natural [column][32 rows]
//  LDS image read with one ds_read_b128 per k-step; parity-green, 254 VGPRs, no spills, and 5.55 / 5.26 / 5.30 / 4.96 ms against 4.90 / 4.64 /
//  4.83 / 4.46 for the register-staged kernel above (cm->cm / cm->p32 / p32->cm / p32->p32, same process order, min = median to 1 %:
//  profiles/r3h_ts32_dma_ab.log).  With the stores compiled out the register kernel runs at 64.8 TFLOP/s, its MFMA work alone takes 4.1 ms.)
__global__ __launch_bounds__(TSK_THREADS, 1) void trmm_right_ts_kernel(const GemmArgs p) {   // N == K == 256
  extern __shared__ __attribute__((aligned(16))) double lds[];
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: trmm_ts_body<0>(p, lds); break;
    case 1: trmm_ts_body<1>(p, lds); break;
    case 2: trmm_ts_body<2>(p, lds); break;
    case 3: trmm_ts_body<3>(p, lds); break;
    case 4: trmm_ts_body<4>(p, lds); break;
    case 5: trmm_ts_body<5>(p, lds); break;
    case 6: trmm_ts_body<6>(p, lds); break;
    default: trmm_ts_body<7>(p, lds); break;
  }
}

template <int TS>
gemm_kernel_t pick_kernel(bool ak, bool bkc) {
  return ak ? (bkc ? dgemm_tile_kernel<TS, true, true> : dgemm_tile_kernel<TS, true, false>)
            : (bkc ? dgemm_tile_kernel<TS, false, true> : dgemm_tile_kernel<TS, false, false>);
}

int64_t count_tiles(const GemmArgs& p, int ts) {
  const int64_t tm = cdiv(p.M, ts), tn = cdiv(p.N, ts);
  return p.out_uplo < 0 ? tm * tn : tm * (tm + 1) / 2;
}

// ws_for_slab: split-K partials go to the handle's primary workspace; callers that already stage through it
// (in-place trmm) pass false.
int launch_gemm(capi_handle_t h, bool ak, bool bkc, GemmArgs& p, bool ws_for_slab) {
  if (p.M <= 0 || p.N <= 0) return CAPI_OK;
  hipStream_t s = h->stream;
  if (p.K <= 0 || p.alpha == 0.0) {
    if (p.beta == 1.0) return CAPI_OK;
    dim3 grid((unsigned)cdiv(p.M, 256), (unsigned)(p.N < 65535 ? p.N : 65535));
    hipLaunchKernelGGL(scale_kernel, grid, dim3(256), 0, s, p.C, p.ldc, p.M, p.N, p.beta, p.out_uplo);
    CAPI_HIP_CHECK(h, hipGetLastError());
    return CAPI_OK;
  }
  // tall-skinny right-TRMM (Q = A R^-1): persistent full-width workgroups, A read once
  {
    static const bool no_ts2 = getenv("CAPI_NO_TS") != nullptr;
    if (!no_ts2 && p.tri_side == CAPI_RIGHT && p.tri_eff_upper && !ak && bkc && p.N == TSK_W && p.K == p.N &&
        (int64_t)p.M >= 64 * (int64_t)p.N) {
      p.a_vec = (((uintptr_t)p.A & 15) == 0) && ((p.lda & 1) == 0);
      p.splitk = 1;
      static const bool rows16 = getenv("CAPI_TS_ROWS16") != nullptr;     // the 16-row-tile variant (A/B)
      CAPI_REQUIRE(h, !(rows16 && (p.a_tiled || p.c_tiled)), "panel32 images need the 32-row T-stationary kernel");
      const int rows = rows16 ? 16 : 32;
      const int ntile = (int)cdiv(p.M, rows);
      const int grid = ntile < h->num_cu ? ntile : h->num_cu;
      const size_t lds_bytes = sizeof(double) * 2 * 256 * rows;
      void (*k)(const GemmArgs) = rows16 ? trmm_right_ts_kernel : trmm_right_ts32_kernel;
      CAPI_RAISE_LDS_LIMIT(h, rows16 ? CAPI_ATTR_TRMM_TS16 : CAPI_ATTR_TRMM_TS32, k, lds_bytes);
      hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(TSK_THREADS), lds_bytes, s, p);
      CAPI_HIP_CHECK(h, hipGetLastError());
      return CAPI_OK;
    }
  }
  // Tall-skinny Gram matrix WIDER than the full-width kernel (CholeskyQR2 at n = 512..2048; config 5: n = 1024, K = 2^23): by 256-blocks.
  // The 128-tiling computes its diagonal tiles whole (36 tile-units for 32 at n = 1024: 11 % of the MFMAs produce the unwanted
  // triangle).  Here the diagonal 256-blocks go to the full-width kernel below, whose 136-of-256 tile map wastes 6 % of a quarter
  // of the work, and the off-diagonal blocks are plain 256 x 256 products A_I^T A_J on the tile kernel (split-K; the four tiles of a
  // slice are consecutive arrivals on one XCD and share their panels in its L2).  (CAPI_NO_TALL: the plain 128-tiling, A/B.)
  {
    static const bool no_tall_gram = getenv("CAPI_NO_TALL") != nullptr || getenv("CAPI_NO_TS") != nullptr;
    if (!no_tall_gram && p.out_uplo == CAPI_UPPER && p.tri_side < 0 && ak && bkc && p.A == p.B && p.lda == p.ldb && p.N > TSK_W &&
        p.N <= 2048 && p.N % TSK_W == 0 && (int64_t)p.K >= 64 * (int64_t)p.N && ws_for_slab && p.batch <= 1) {
      const int nb = p.N / TSK_W;
      for (int J = 0; J < nb; ++J)
        for (int I = 0; I <= J; ++I) {
          GemmArgs q = p;
          q.A = p.A + (int64_t)I * TSK_W * p.lda;
          q.B = p.A + (int64_t)J * TSK_W * p.lda;
          q.C = p.C + (int64_t)I * TSK_W + (int64_t)J * TSK_W * p.ldc;
          q.M = q.N = TSK_W;
          q.out_uplo = I == J ? CAPI_UPPER : -1;
          int rc = launch_gemm(h, true, true, q, true);
          if (rc != CAPI_OK) return rc;
        }
      return CAPI_OK;
    }
  }
  // tall-skinny Gram matrix: full-width workgroups, the tall operand is read once
  {
    static const bool no_ts = getenv("CAPI_NO_TS") != nullptr;
    if (!no_ts && p.out_uplo == CAPI_UPPER && p.tri_side < 0 && ak && bkc && p.A == p.B && p.lda == p.ldb && p.N <= TSK_W &&
        p.N >= 64 && (int64_t)p.K >= 64 * (int64_t)p.N && ws_for_slab) {
      const int64_t P = cdiv(p.K, BK);
      int S = (int)(P / 32 < h->num_cu ? (P / 32 > 0 ? P / 32 : 1) : h->num_cu);       // one resident workgroup per CU
      p.a_vec = (((uintptr_t)p.A & 15) == 0) && ((p.lda & 1) == 0);
      p.splitk = S;
      p.slab = nullptr; p.slab_ld = p.N; p.slab_stride = (int64_t)p.N * p.N;
      if (S > 1) {
        void* ws;
        int rc = capi_ws_get(h, sizeof(double) * (size_t)p.slab_stride * (size_t)S, &ws);
        if (rc != CAPI_OK) return rc;
        p.slab = (double*)ws;
      }
      const size_t lds_bytes = sizeof(double) * 2 * TSK_W * SK;
      CAPI_RAISE_LDS_LIMIT(h, CAPI_ATTR_GRAM_TS, gram_ts_kernel, lds_bytes);
      hipLaunchKernelGGL(gram_ts_kernel, dim3((unsigned)S), dim3(TSK_THREADS), lds_bytes, s, p);
      CAPI_HIP_CHECK(h, hipGetLastError());
      if (S >= 16 && p.N <= 65535) {
        hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)cdiv(p.M, 64), (unsigned)p.N), dim3(256), 0, s, p.slab, p.slab_ld,
                           p.slab_stride, p.splitk, p.C, p.ldc, p.M, p.N, p.alpha, p.beta, p.out_uplo);
        CAPI_HIP_CHECK(h, hipGetLastError());
      } else if (S > 1) {
        dim3 grid((unsigned)cdiv(p.M, 256), (unsigned)(p.N < 65535 ? p.N : 65535));
        hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, s, p.slab, p.slab_ld, p.slab_stride, p.splitk, p.C, p.ldc,
                           p.M, p.N, p.alpha, p.beta, p.out_uplo);
        CAPI_HIP_CHECK(h, hipGetLastError());
      }
      return CAPI_OK;
    }
  }
  // (panel32 images are understood by the two full-width tall-skinny kernels above and by nothing below: the entry points check the switches that
  //  could route such an operand here -- CAPI_NO_TS, CAPI_TS_ROWS16 -- per call, the dispatch above reads them once per process, so a process that
  //  changes them between calls must get a refusal, not a tile kernel that reads an image as column-major with ld 32)
  CAPI_REQUIRE(h, !p.a_tiled && !p.c_tiled, "panel32 operand reached a kernel that does not read it");
  // Choose tile size and split-K from a small cost model in CU-cycles.  One k-panel (16 deep) of a 128-tile keeps all
  // four MFMA pipes of a CU busy for 64 MFMAs x 64 cycles = 4096 cycles, of a 64-tile for 1024; co-resident workgroups
  // share the pipes, so a CU works through the tiles dealt to it at that rate whatever their number.  The makespan is
  // the busiest CU's queue: ceil(tiles / CUs) equal tiles, or for TRMM (k-range grows linearly along the triangular
  // dimension, longest-first dealing) the larger of the mean load and the single longest tile.
  const bool tri = p.tri_side >= 0;
  const double ghz = 2.35;
  double best = 1e300;
  int best_ts = 128, best_s = 1;
  static const char* force_ts = getenv("CAPI_FORCE_TS");
  for (int ts : {128, 64}) {
    if (force_ts && atoi(force_ts) != ts) continue;
    const double nt = (double)count_tiles(p, ts);
    const double cyc = ts == 128 ? 4096.0 : 1024.0;
    const double eff = ts == 128 ? 0.89 : 0.80;          // measured pipe utilisation of the two kernels (fast path)
    const int ncu = h->cu_of[h->cur] ? h->cu_of[h->cur] : h->num_cu;
    for (int sk = 1; sk <= 512; ++sk) {
      if (sk > 1 && (!ws_for_slab || p.K / sk < 256)) break;
      double busiest;                                     // cycles of work queued on the busiest CU
      if (!tri) {
        // a CU keeps `res` workgroups resident; the last, partially filled group of its queue runs without partners to
        // cover its barrier and load stalls (measured ~0.7x the paired rate for a lone 128-tile workgroup)
        const int64_t q = cdiv((int64_t)nt * sk, ncu);
        const int res = ts == 128 ? 2 : 4;
        const int64_t lone = q % res;
        const double per = ((double)p.K / sk / 16.0) * cyc;
        busiest = (double)(q - lone) * per + (double)lone * per / (lone == 0 ? 1.0 : (0.62 + 0.38 * (double)lone / res));
      } else {
        const double kmax = (double)p.K / sk, kavg = (0.5 * p.K + 0.5 * ts) / sk;
        const double mean = nt * sk * (kavg / 16.0) * cyc / ncu, longest = (kmax / 16.0) * cyc;
        busiest = mean * 1.08 > longest ? mean * 1.08 : longest;
      }
      double t = busiest / (ghz * 1e3 * eff) + 7.0;
      // operand panels stream from L2/MALL: ~4 TB/s effective when every tile re-reads its two panels
      // (a syrk's diagonal tiles stage one panel; a TRMM's triangular operand is small and stays cache resident)
      const double keff = tri ? 0.5 * p.K + 0.5 * ts : (double)p.K;
      const double panels = (p.out_uplo >= 0 && p.A == p.B) ? 2.0 * nt - (double)cdiv(p.N, ts) : (tri ? 1.0 * nt : 2.0 * nt);
      // (operands that fit the 256 MB Infinity Cache are re-read from there at roughly twice the HBM-side rate)
      const double footprint = ((p.A == p.B ? 0.0 : (double)p.M) + (double)p.N) * (double)p.K * 8.0;
      double t_mem = panels * ts * keff * 8.0 / (footprint <= 192.0e6 ? 8.0e6 : 4.0e6);
      // a split-K slice whose tiles all fit one XCD's resident set (<= 32 tiles: consecutive pids, started together, walking the
      // same panels in step) shares those panels in that L2: a tall product then streams each operand about once
      if (!tri && nt <= 32.0 && (double)p.K >= 64.0 * (double)(p.M > p.N ? p.M : p.N)) t_mem = footprint / 4.0e6;
      if (t_mem > t) t = t_mem;
      if (sk > 1) t = 1.12 * t + 6.0 + (double)(sk + 2) * (double)p.M * (double)p.N * (p.out_uplo >= 0 ? 0.5 : 1.0) * 8.0 / 2.5e6;
      if (t < best) { best = t; best_ts = ts; best_s = sk; }
    }
  }
  static const bool dbg = getenv("CAPI_DEBUG_GEMM") != nullptr;
  if (dbg) fprintf(stderr, "[capi gemm] M=%d N=%d K=%d uplo=%d tri=%d -> ts=%d splitk=%d est=%.1f us\n", p.M, p.N, p.K, p.out_uplo, p.tri_side, best_ts, best_s, best);
  // latency-bound sizes go to the burst-load 32-tile kernel: one workgroup per CU (139 KB of LDS), per 256-deep chunk
  // ~2 us of exposed load latency + 64 MFMAs per wave
  {
    static const char* force_small = getenv("CAPI_SMALL");
    const double nt32 = (double)cdiv(p.M, ST) * (double)cdiv(p.N, ST) * (p.out_uplo >= 0 ? 0.5 : 1.0);
    const double keff = tri ? 0.5 * p.K + 16.0 : (double)p.K;
    const double chunks = keff / SKC < 1.0 ? 1.0 : keff / SKC;
    const double t_small = (double)cdiv((int64_t)nt32, h->num_cu) * (chunks * 2.0 + keff * (16.0 / 2200.0) * 4.0 / 4.0) + 3.0;
    bool use_small = p.M <= 512 && p.N <= 512 && p.K <= 2048;      // measured: 1.5-2x faster up to order 512, slower from 1024
    // thin products of the blocked factorization (K <= 256: a 128-row panel against up to ~2000 columns, its trailing
    // update): one staged chunk, LDS sized by K, so two workgroups share a CU at K <= 128
    static const int thin_rounds = getenv("CAPI_THIN_ROUNDS") ? atoi(getenv("CAPI_THIN_ROUNDS")) : 3;
    if (!use_small && p.K <= SKC && p.batch <= 1) {
      const double slots = (double)h->num_cu * (p.K <= 128 ? 2.0 : 1.0);
      use_small = nt32 <= thin_rounds * slots;
    }
    if (force_small) use_small = atoi(force_small) != 0 && p.M <= 4096 && p.N <= 4096;
    if (dbg) fprintf(stderr, "[capi gemm]   small-kernel estimate %.1f us -> %s\n", t_small, use_small ? "small" : "tile");
    if (use_small) {
      const int kcap = p.K >= SKC ? SKC : (int)(cdiv(p.K, SQK) * SQK);      // staged depth: LDS holds 2 x kcap x SLD doubles
      p.ts = kcap;
      p.tiles_m = (int)cdiv(p.M, ST);
      p.tiles_n = (int)cdiv(p.N, ST);
      p.ntiles = p.tiles_m * p.tiles_n;
      p.a_vec = (((uintptr_t)p.A & 15) == 0) && ((p.lda & 1) == 0);
      p.b_vec = (((uintptr_t)p.B & 15) == 0) && ((p.ldb & 1) == 0);
      p.splitk = 1;
      gemm_kernel_t k = pick_small(ak, bkc);
      const size_t lds_bytes = sizeof(double) * 2 * (size_t)kcap * SLD;
      const int vi = (ak ? 2 : 0) + (bkc ? 1 : 0);
      CAPI_RAISE_LDS_LIMIT(h, CAPI_ATTR_SMALL0 + vi, k, sizeof(double) * 2 * SKC * SLD);
      if (dbg) {
        static long launch_no = 0;
        fprintf(stderr, "[capi gemm]   small launch #%ld v=%d grid=(%d,%d) lds=%zu stream=%p A=%p lda=%ld B=%p ldb=%ld C=%p ldc=%ld batch=%d sa=%ld sb=%ld sc=%ld\n",
                ++launch_no, vi, p.ntiles, p.batch > 1 ? p.batch : 1, lds_bytes, (void*)s, (const void*)p.A, (long)p.lda, (const void*)p.B, (long)p.ldb,
                (void*)p.C, (long)p.ldc, p.batch, (long)p.sa, (long)p.sb, (long)p.sc);
      }
      hipLaunchKernelGGL(k, dim3((unsigned)p.ntiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(256), lds_bytes, s, p);
      CAPI_HIP_CHECK(h, hipGetLastError());
      return CAPI_OK;
    }
  }
  p.ts = best_ts;
  p.tiles_m = (int)cdiv(p.M, p.ts);
  p.tiles_n = (int)cdiv(p.N, p.ts);
  p.ntiles = (int)count_tiles(p, p.ts);
  p.a_vec = (((uintptr_t)p.A & 15) == 0) && ((p.lda & 1) == 0);
  p.b_vec = (((uintptr_t)p.B & 15) == 0) && ((p.ldb & 1) == 0);
  p.no_skip = getenv("CAPI_NO_SKIP") ? 1 : 0;
  p.share_ab = (p.out_uplo >= 0 && p.A == p.B && p.lda == p.ldb && ak == bkc && !getenv("CAPI_NO_SHARE")) ? 1 : 0;
  {
    static const int order_mode = getenv("CAPI_TILE_ORDER") ? atoi(getenv("CAPI_TILE_ORDER")) : 1;
    static const int order_min = getenv("CAPI_TILE_ORDER_MIN") ? atoi(getenv("CAPI_TILE_ORDER_MIN")) : 16;
    p.order = 0;
    if (p.out_uplo >= 0 && (order_mode & 1) && p.tiles_n >= order_min) p.order |= 1;
  }
  p.splitk = 1;
  p.k_per_split = p.K;
  p.k_rotate = 0;
  p.slab = nullptr;
  p.slab_ld = p.slab_stride = 0;
  if (best_s > 1) {
    const int64_t kps = cdiv(cdiv(p.K, best_s), BK) * BK;
    const int64_t sk = cdiv(p.K, kps);
    if (sk > 1) {
      p.splitk = (int)sk;
      p.k_per_split = (int)kps;
      p.k_rotate = (!tri && !getenv("CAPI_NO_ROTATE")) ? 1 : 0;
      p.slab_ld = p.M;
      p.slab_stride = (int64_t)p.M * p.N;
      void* ws;
      int rc = capi_ws_get(h, sizeof(double) * (size_t)p.slab_stride * (size_t)sk, &ws);
      if (rc != CAPI_OK) return rc;
      p.slab = (double*)ws;
    }
  }
  // A 128-tiling fills the chip in rounds of 2 x CUs workgroups; the last round is usually partial and its lone
  // workgroups run at ~0.6 of the paired rate (8256 tiles = 16 rounds + 64: those 64 cost almost another round).  The
  // tail is re-cut into 64-tiles (4x the workgroups, a quarter of the length) and launched right behind the full rounds.
  // (CAPI_ROUNDS_MIN_K: products shallower than this keep the handle's environment defaults -- one launch, pairs up to four whole rounds --
  //  even while capi_set_launch_rounds is on: a round of a shallow product is short, and every boundary drains the chip once; A/B knob)
  static const int rounds_min_k = getenv("CAPI_ROUNDS_MIN_K") ? atoi(getenv("CAPI_ROUNDS_MIN_K")) : 0;
  const bool deep = p.K >= rounds_min_k;
  const int rounds_mode = deep ? h->rounds_mode : h->rounds_env[0];
  const int64_t per_round = 2 * (int64_t)(h->cu_of[h->cur] ? h->cu_of[h->cur] : h->num_cu);
  // (bit 0: plain products; bit 1: triangular outputs in the banded order -- an XCD's 64 tiles of a round are an 8 x 8 block of the triangle)
  // (the 256-column block launches of a tall right-TRMM were tried the same way: a round there is 512 tiles of K <= 1024, ~0.15 ms, and the
  //  launch boundaries cost 11 %: 35.1 -> 39.2 ms at m = 2^21, n = 1024; r3s)
  const bool use_rounds = p.ts == 128 && p.splitk == 1 && !tri && per_round % 16 == 0 && p.batch <= 1 && (int64_t)p.ntiles >= 2 * per_round &&
                          (p.out_uplo < 0 ? (rounds_mode & 1) != 0 : ((rounds_mode & 2) != 0 && (p.order & 1)));      // (see "Resident rounds" below)
  int tail128 = 0;
  if (p.ts == 128 && p.splitk == 1 && !tri && !getenv("CAPI_NO_TAIL")) {
    const int per_round = 2 * (h->cu_of[h->cur] ? h->cu_of[h->cur] : h->num_cu);
    const int rem = p.ntiles % per_round;
    if (p.ntiles >= 2 * per_round && rem > 0 && rem <= (3 * per_round) / 4) tail128 = rem;
  }
  const int ntiles_all = p.ntiles;
  p.tail_base = 0; p.tail_tm = p.tail_tn = 0;
  p.ntiles -= tail128;
  gemm_kernel_t k = p.ts == 128 ? pick_kernel<128>(ak, bkc) : pick_kernel<64>(ak, bkc);
  const size_t lds_bytes = sizeof(double) * 2 * (p.ts == 128 ? tile_cfg<128>::STAGE_LDS : tile_cfg<64>::STAGE_LDS);
  const int64_t nblk = (int64_t)p.ntiles * p.splitk;
  CAPI_REQUIRE(h, nblk < (int64_t)1 << 31, "too many tiles");
  // HIP events around EVERY launch of the tile kernel (capi_prof_*): one record per launch, its share of the algorithmic flops
  auto prof_open = [&](double share, capi_handle_s::prof_rec*& rec) -> int {
    rec = nullptr;
    if (!h->prof_on) return CAPI_OK;
    if (h->prof_n == h->prof_cap) {
      const int ncap = h->prof_cap ? h->prof_cap * 2 : 1024;
      auto* np_ = (capi_handle_s::prof_rec*)realloc(h->prof, sizeof(capi_handle_s::prof_rec) * ncap);
      if (!np_) return CAPI_ENOMEM;
      for (int i = h->prof_cap; i < ncap; ++i) { np_[i].e0 = nullptr; np_[i].e1 = nullptr; }
      h->prof = np_;
      h->prof_cap = ncap;
    }
    rec = &h->prof[h->prof_n++];
    if (!rec->e0) { CAPI_HIP_CHECK(h, hipEventCreate(&rec->e0)); CAPI_HIP_CHECK(h, hipEventCreate(&rec->e1)); }
    // algorithmic flops of the whole product: gemm 2MNK, triangular output N(N+1)K, trmm M^2 N / M N^2
    // (a tri_block launch is one 256-column block of a tall right-TRMM: a dense product above T's diagonal block plus that block's triangle)
    rec->flops = p.out_uplo >= 0 ? (double)p.N * ((double)p.N + 1.0) * (double)p.K
                 : (p.tri_side >= 0 ? (p.tri_block ? (double)p.M * (double)p.N * (2.0 * (double)p.tri_koff + (double)p.N) : (double)p.M * (double)p.N * (double)p.K)
                                    : 2.0 * (double)p.M * (double)p.N * (double)p.K);
    rec->flops *= share;
    rec->variant = (ak ? 2 : 0) + (bkc ? 1 : 0) + (p.ts == 128 ? 0 : 4);
    rec->m = p.M; rec->n = p.N; rec->k = p.K; rec->kind = p.out_uplo >= 0 ? 1 : (p.tri_side >= 0 ? 2 : 0);
    CAPI_HIP_CHECK(h, hipEventRecord(rec->e0, s));
    return CAPI_OK;
  };
  // the launch's device-side interval record (capi_prof_collect_intervals), or null when profiling is off / the pool is exhausted
  auto stamp_of = [&](const capi_handle_s::prof_rec* rec) -> unsigned long long* {
    const int64_t idx = rec ? rec - h->prof : -1;
    return (rec && h->d_stamps && idx < h->stamps_cap) ? h->d_stamps + 2 * idx : nullptr;
  };
  // Resident rounds (plain products).  A launch with more tiles than the chip holds (2 per CU) refills slots one by one as tiles
  // finish: within a few tile lengths the starts are smeared and tiles that share an operand panel are no longer within the ~2
  // iterations an XCD's 4 MiB L2 can bridge (its 64 resident tiles pull 2 MiB of panels through it per iteration).  One launch per
  // round restarts every XCD's 64 tiles together, as an 8 x 8 block of the tile grid (tile_of_dims' bands): 16 panels serve 64
  // tiles.  dgemm 16384^3: FETCH 139 -> 76 GB (the 8-way ideal is 69), time unchanged (118.7 vs 119.0 ms): the tiles of a plain
  // product do equal work, so the round boundary costs nothing measurable.  Triangular outputs were tried the same way (8 x 8
  // super-blocks of the triangle, 8 per round): FETCH of the n = 32768 step 310 -> 263 GB only, dsyrk 16384 62.6 -> 64.2 ms,
  // step 237 -> 242 ms (partial rounds, diagonal super-blocks with 36 live tiles): not kept.  TRMM tiles have unequal k-ranges
  // and keep the longest-first free-running order.  OFF by default (CAPI_ROUNDS=1 turns it on): inside cholinv the plain products
  // are the lookahead's rectangles only -- the step's fabric traffic falls by 5 % (3625 -> 3437 GB at n = 65536), its time does
  // not change, and the per-launch durations the roofline is computed from stretch, because the round launches of a low-priority
  // bulk stream queue behind the chain's kernels at every boundary (0.876 -> 0.825 on the same box).
  // TRMM in tile pairs (dtrmm_pair_kernel): equal work per workgroup, the launch of a plain product
  {
    const int pair_mode = deep ? h->pair_mode : h->rounds_env[1];
    // Measured (tools/pair_window.py, all three forms of the recursion): a launch of exactly one resident round +8..10 % (order 4096:
    // 63 -> 68.5 TFLOP/s; 2048 x 8192: 51..55 -> 55..59), two rounds +2..3 %, four +0.5..1 %, nine +-0.5 %; a launch that is NOT whole
    // rounds loses (1152 workgroups, order 6144: 68.3 -> 61.5 -- the equal, long workgroups of the last 128 cost a third round).
    // L2-to-fabric traffic does not change (order 32768: 2 x FETCH_SIZE 1.19 -> 1.22 TB).  Pairs therefore run up to four whole rounds
    // (CAPI_TRMM_PAIR=2: whenever the launch is whole rounds; =0: never); larger products keep the longest-first order.
    const int ntri_ = p.tri_side == CAPI_LEFT ? p.tiles_m : p.tiles_n, nfree_ = p.tri_side == CAPI_LEFT ? p.tiles_n : p.tiles_m;
    const int64_t wgs = (int64_t)(ntri_ / 2) * nfree_;
    const int pair_rounds = deep ? h->pair_rounds : h->rounds_env[2], pair_rounds_min = deep ? h->pair_rounds_min : h->rounds_env[3];
    if (pair_mode && tri && !p.tri_dense && !p.tri_block && p.tri_koff == 0 && p.ts == 128 && p.splitk == 1 && p.beta == 0.0 && p.batch <= 1 &&
        p.M % 128 == 0 && p.N % 128 == 0 && p.K % 128 == 0 && p.a_vec && p.b_vec && (ntri_ & 1) == 0 &&
        wgs % per_round == 0 && (pair_mode > 1 || wgs <= 4 * per_round || (pair_rounds && p.K >= pair_rounds_min))) {
      gemm_kernel_t kp = ak ? (bkc ? dtrmm_pair_kernel<true, true> : dtrmm_pair_kernel<true, false>)
                            : (bkc ? dtrmm_pair_kernel<false, true> : dtrmm_pair_kernel<false, false>);
      CAPI_RAISE_LDS_LIMIT(h, CAPI_ATTR_PAIR0 + (ak ? 2 : 0) + (bkc ? 1 : 0), kp, lds_bytes);
      // Pairs do equal work, so a launch can go out one resident round (512 workgroups: an 8 x 8 block of pair-tiles per XCD) at a time
      // at no cost in time, and every round's tiles start -- and, walking equal k-ranges, stay -- together: the panels an XCD's 64 tiles
      // share are fetched once instead of once per drifting tile (CAPI_TRMM_PAIR_ROUNDS; measured in round 3).
      if (pair_rounds && wgs > per_round && p.K >= pair_rounds_min) {
        for (int64_t base = 0; base < wgs; base += per_round) {
          GemmArgs q = p;
          q.pid_base = (int)base;
          capi_handle_s::prof_rec* rec;
          int rc = prof_open((double)per_round / (double)wgs, rec);
          if (rc != CAPI_OK) return rc;
          if (rec) rec->variant += 16;
          q.stamp = stamp_of(rec);
          hipLaunchKernelGGL(kp, dim3((unsigned)per_round), dim3(NTHREADS), lds_bytes, s, q);
          if (rec) CAPI_HIP_CHECK(h, hipEventRecord(rec->e1, s));
        }
        CAPI_HIP_CHECK(h, hipGetLastError());
        return CAPI_OK;
      }
      capi_handle_s::prof_rec* rec;
      int rc = prof_open(1.0, rec);
      if (rc != CAPI_OK) return rc;
      if (rec) rec->variant += 16;                                      // its own kernel symbol: not counted with dgemm_tile_kernel's launches
      p.stamp = stamp_of(rec);
      hipLaunchKernelGGL(kp, dim3((unsigned)wgs), dim3(NTHREADS), lds_bytes, s, p);
      if (rec) CAPI_HIP_CHECK(h, hipEventRecord(rec->e1, s));
      CAPI_HIP_CHECK(h, hipGetLastError());
      return CAPI_OK;
    }
  }
  if (use_rounds) {
    for (int64_t base = 0; base < nblk; base += per_round) {
      GemmArgs q = p;
      q.pid_base = (int)base;
      const int64_t cnt = nblk - base < per_round ? nblk - base : per_round;
      capi_handle_s::prof_rec* rec;
      int rc = prof_open((double)cnt / (double)ntiles_all, rec);
      if (rc != CAPI_OK) return rc;
      q.stamp = stamp_of(rec);
      hipLaunchKernelGGL(k, dim3((unsigned)cnt), dim3(NTHREADS), lds_bytes, s, q);
      if (rec) CAPI_HIP_CHECK(h, hipEventRecord(rec->e1, s));
    }
  } else {
    capi_handle_s::prof_rec* rec;
    int rc = prof_open((double)p.ntiles / (double)ntiles_all, rec);              // (the tail launch below is not part of this record)
    if (rc != CAPI_OK) return rc;
    p.stamp = stamp_of(rec);
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(NTHREADS), lds_bytes, s, p);
    if (rec) CAPI_HIP_CHECK(h, hipEventRecord(rec->e1, s));
  }
  CAPI_HIP_CHECK(h, hipGetLastError());
  if (tail128 > 0) {
    GemmArgs q = p;
    q.ts = 64;
    q.stamp = nullptr;                      // (the tail is neither in the record's flops nor in its interval)
    q.tail_base = p.ntiles;                 // first 128-tile of the tail (p.ntiles > 0 here)
    q.tail_tm = p.tiles_m; q.tail_tn = p.tiles_n;
    q.tiles_m = (int)cdiv(p.M, 64); q.tiles_n = (int)cdiv(p.N, 64);
    q.ntiles = 4 * tail128;
    hipLaunchKernelGGL(pick_kernel<64>(ak, bkc), dim3((unsigned)q.ntiles), dim3(NTHREADS),
                       sizeof(double) * 2 * tile_cfg<64>::STAGE_LDS, s, q);
    CAPI_HIP_CHECK(h, hipGetLastError());
  }
  if (p.splitk > 1) {
    dim3 grid((unsigned)cdiv(p.M, 256), (unsigned)(p.N < 65535 ? p.N : 65535));
    hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, s, p.slab, p.slab_ld, p.slab_stride, p.splitk, p.C, p.ldc,
                       p.M, p.N, p.alpha, p.beta, p.out_uplo);
    CAPI_HIP_CHECK(h, hipGetLastError());
  }
  return CAPI_OK;
}

bool ok01(int v) { return v == 0 || v == 1; }

// ---- register-resident MFMA loop: the measured fp64 matrix peak of this device -----------------------
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(double* out, int iters) {
  d4_t acc[8];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = (d4_t){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
    // inline asm pins the accumulators in VGPRs: with the builtin hipcc shuttles all 64 of them through AGPRs
    // every iteration (128 v_accvgpr moves per 8 MFMAs) and the loop measures that traffic, not the matrix pipe
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // let the last MFMAs retire before their results are read
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

extern "C" {

int capi_dgemm(capi_handle_t h, int transA, int transB, int64_t m, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ok01(transA) && ok01(transB), "transpose code");
  CAPI_REQUIRE(h, m >= 0 && n >= 0 && k >= 0 && m < (1LL << 31) && n < (1LL << 31) && k < (1LL << 31), "dims");
  if (m == 0 || n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, C && ldc >= m, "C/ldc");
  if (k > 0) {
    CAPI_REQUIRE(h, A && B, "null operand");
    CAPI_REQUIRE(h, lda >= (transA ? k : m) && ldb >= (transB ? n : k), "lda/ldb");
  }
  GemmArgs p{};
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = (int)m; p.N = (int)n; p.K = (int)k; p.alpha = alpha; p.beta = beta;
  p.out_uplo = -1; p.tri_side = -1;
  return launch_gemm(h, transA == CAPI_TRANS, transB == CAPI_NOTRANS, p, true);
}

int capi_dgemmt(capi_handle_t h, int uplo, int transA, int transB, int64_t n, int64_t k, double alpha,
                const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ok01(uplo) && ok01(transA) && ok01(transB), "enum code");
  CAPI_REQUIRE(h, n >= 0 && k >= 0 && n < (1LL << 31) && k < (1LL << 31), "dims");
  if (n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, C && ldc >= n, "C/ldc");
  if (k > 0) {
    CAPI_REQUIRE(h, A && B, "null operand");
    CAPI_REQUIRE(h, lda >= (transA ? k : n) && ldb >= (transB ? n : k), "lda/ldb");
  }
  GemmArgs p{};
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = (int)n; p.N = (int)n; p.K = (int)k; p.alpha = alpha; p.beta = beta;
  p.out_uplo = uplo; p.tri_side = -1;
  return launch_gemm(h, transA == CAPI_TRANS, transB == CAPI_NOTRANS, p, true);
}

int capi_dsyrk(capi_handle_t h, int uplo, int trans, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, double beta, double* C, int64_t ldc) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ok01(trans), "transpose code");
  // Trans: C = A^T A (A is k x n);  NoTrans: C = A A^T (A is n x k)
  return capi_dgemmt(h, uplo, trans, trans == CAPI_TRANS ? CAPI_NOTRANS : CAPI_TRANS, n, k, alpha, A, lda, A, lda, beta, C, ldc);
}

static int trmm_launch(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
                       const double* T, int64_t ldt, const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                       bool ws_free = true) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ok01(side) && ok01(uplo) && ok01(trans) && ok01(diag), "enum code");
  CAPI_REQUIRE(h, m >= 0 && n >= 0 && m < (1LL << 31) && n < (1LL << 31), "dims");
  if (m == 0 || n == 0) return CAPI_OK;
  const int64_t nt = side == CAPI_LEFT ? m : n;
  CAPI_REQUIRE(h, T && B && C && ldt >= nt && ldb >= m && ldc >= m, "operands");
  CAPI_REQUIRE(h, (const double*)C != B, "out-of-place trmm: C aliases B");
  GemmArgs p{};
  p.C = C; p.ldc = ldc; p.M = (int)m; p.N = (int)n; p.K = (int)nt; p.alpha = alpha; p.beta = beta;
  p.out_uplo = -1;
  p.tri_side = side;
  p.tri_eff_upper = ((uplo == CAPI_UPPER) != (trans == CAPI_TRANS));
  p.tri_unit = diag == CAPI_UNIT;
  // Tall right-TRMM (CholeskyQR2's Q = A R^-1 at n = 512..2048; config 5 is 2^23 x 1024 per GPU): every column tile's k-range ends in
  // eight panels that cross T's diagonal, which the kernel's generic loop masks and prunes one by one -- a fifth of all
  // iterations at n = 1024.  Here T's triangle is copied ONCE into a zeroed n x n block of the handle (8 MiB at n = 1024, read
  // by every tile anyway) and the kernel is told the other triangle holds zeros: all iterations take the lean loop; the k-range
  // of a tile is still cut at its diagonal block, the other triangle of the CALLER's T is still never read.
  static const bool no_tall = getenv("CAPI_NO_TALL") != nullptr;
  if (!no_tall && side == CAPI_RIGHT && diag == CAPI_NONUNIT && n >= 512 && n <= 4096 && m >= 64 * n) {
    void* w;
    int rc = capi_ws3_get(h, sizeof(double) * (size_t)n * (size_t)n, &w);
    if (rc != CAPI_OK) return rc;
    CAPI_HIP_CHECK(h, hipMemsetAsync(w, 0, sizeof(double) * (size_t)n * (size_t)n, h->stream));
    rc = capi_dlacpy(h, uplo == CAPI_UPPER ? 1 : 2, n, n, T, ldt, (double*)w, n);
    if (rc != CAPI_OK) return rc;
    T = (const double*)w;
    ldt = n;
    p.tri_dense = 1;
    // Upper op(T) and n a multiple of 256: one launch per 256-column block of the output, C_J = B(:, 0 : 256 (J + 1)) op(T)(0 : 256 (J + 1), J),
    // tiles dealt as a plain product's (bands of 8 row tiles x the block's two column tiles per XCD).  A launch's slice of T (<= 2 MiB at
    // n = 1024) stays in every L2 and its tiles have near-equal k-ranges: m = 2^23: 147.0 -> 142.5 ms, n = 512: 51.8 -> 56.6 TFLOP/s; block 0
    // (K = 256) is the T-stationary kernel's shape.  B is still fetched once per column tile (FETCH_SIZE unchanged): an XCD streams
    // 2 MiB of panels per iteration through its 4 MiB L2, so a partner tile one iteration behind already misses; making the two
    // column tiles of a row tile consecutive arrivals was measured slower (152.7 ms).  (CAPI_TALL_ONE_LAUNCH: A/B.)
    static const bool one_launch = getenv("CAPI_TALL_ONE_LAUNCH") != nullptr;
    const bool eff_upper = (uplo == CAPI_UPPER) != (trans == CAPI_TRANS);
    // (Round 3, measured and dropped: each block's diagonal 256 x 256 part by the T-stationary kernel -- exactly the 136 live MFMA tiles,
    //  B_J read once -- and the part above it as a dense beta = 1 product on the tile kernel: 16.25 instead of 18 executed units of
    //  m * 65536 flops at n = 1024, yet 36.1 against 35.3 ms at m = 2^21 (n = 512: 10.4 against 9.8): the accumulating products have
    //  K = 256 .. 768 only, and a 128-tile with 16-48 iterations spends too much of its life in prologue, C read and epilogue.)
    // (Round 4, measured and dropped: FAST iterations over the eight panels that cross T's diagonal with the wave's dead sub-tile COLUMNS left out
    //  (44 % of those panels' MFMAs, wave-uniform branches, no masking needed on the clean copy): 141.4 / 141.9 against 141.4 / 141.7 ms at
    //  m = 2^23, n = 1024 -- nothing.  A panel costs what its busiest wave costs: the waves of the tile's right half keep all their columns until
    //  the last three panels, and the barrier makes the others wait for them.  profiles/r4_tall_trmm_band_skip_ab.txt.)
    if (!one_launch && eff_upper && n % 256 == 0) {
      for (int64_t J = 0; J < n / 256; ++J) {
        GemmArgs q = p;
        q.N = 256;
        q.K = (int)(256 * (J + 1));
        q.tri_koff = (int)(256 * J);
        q.tri_block = 1;
        q.C = C + 256 * J * ldc;
        q.A = B; q.lda = ldb;
        // op(T)(k, j) for k < K, j in the block: NoTrans -> T[k + j ldt] (k-contiguous), Trans -> T[j + k ldt]
        q.B = trans == CAPI_NOTRANS ? T + 256 * J * ldt : T + 256 * J;
        q.ldb = ldt;
        int rc2 = launch_gemm(h, false, trans == CAPI_NOTRANS, q, ws_free);
        if (rc2 != CAPI_OK) return rc2;
      }
      return CAPI_OK;
    }
  }
  if (side == CAPI_LEFT) {  // C = alpha op(T) B : A-operand = T (transA = trans), B-operand = B (NoTrans)
    p.A = T; p.lda = ldt; p.B = B; p.ldb = ldb;
    return launch_gemm(h, trans == CAPI_TRANS, true, p, ws_free);
  } else {                  // C = alpha B op(T) : A-operand = B (NoTrans), B-operand = T (transB = trans)
    p.A = B; p.lda = ldb; p.B = T; p.ldb = ldt;
    return launch_gemm(h, false, trans == CAPI_NOTRANS, p, ws_free);
  }
}

// `batch` independent out-of-place TRMMs of one shape whose operands sit a fixed stride apart (the off-diagonal blocks
// of one level of a triangular inverse): ONE launch of the burst-load kernel when it serves the shape, a loop otherwise.
__attribute__((visibility("hidden"))) int capi_internal_trmm_oop_batched(capi_handle_t h, int side, int uplo, int trans, int diag,
                                                                           int64_t m, int64_t n, double alpha, const double* T,
                                                                           int64_t ldt, int64_t st, const double* B, int64_t ldb,
                                                                           int64_t sb_, double* C, int64_t ldc, int64_t sc_, int batch) {
  CAPI_REQUIRE(h, h && m >= 0 && n >= 0 && m < (1LL << 31) && n < (1LL << 31) && batch >= 0, "dims");      // (GemmArgs carries 32-bit extents)
  const bool small_ok = m <= 512 && n <= 512 && !getenv("CAPI_SMALL");
  const bool aligned = ((st | sb_ | sc_) & 1) == 0;      // keeps the 16-byte alignment decision valid for every batch member
  if (batch > 1 && small_ok && aligned) {
    const int64_t nt = side == CAPI_LEFT ? m : n;
    GemmArgs p{};
    p.C = C; p.ldc = ldc; p.M = (int)m; p.N = (int)n; p.K = (int)nt; p.alpha = alpha; p.beta = 0.0;
    p.out_uplo = -1; p.tri_side = side;
    p.tri_eff_upper = ((uplo == CAPI_UPPER) != (trans == CAPI_TRANS));
    p.tri_unit = diag == CAPI_UNIT;
    p.batch = batch;
    if (side == CAPI_LEFT) { p.A = T; p.lda = ldt; p.sa = st; p.B = B; p.ldb = ldb; p.sb = sb_; p.sc = sc_; return launch_gemm(h, trans == CAPI_TRANS, true, p, true); }
    p.A = B; p.lda = ldb; p.sa = sb_; p.B = T; p.ldb = ldt; p.sb = st; p.sc = sc_;
    return launch_gemm(h, false, trans == CAPI_NOTRANS, p, true);
  }
  for (int b = 0; b < batch; ++b) {
    int rc = capi_dtrmm_oop(h, side, uplo, trans, diag, m, n, alpha, T + b * st, ldt, B + b * sb_, ldb, C + b * sc_, ldc);
    if (rc != CAPI_OK) return rc;
  }
  return CAPI_OK;
}

int capi_dtrmm_oop(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
                   const double* T, int64_t ldt, const double* B, int64_t ldb, double* C, int64_t ldc) {
  return trmm_launch(h, side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb, 0.0, C, ldc);
}

int capi_dtrmm_acc(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
                   const double* T, int64_t ldt, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  return trmm_launch(h, side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb, beta, C, ldc);
}

int capi_dtrmm(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb) {
  CAPI_REQUIRE(h, h, "null handle");
  if (m <= 0 || n <= 0) return m < 0 || n < 0 ? CAPI_EINVAL : CAPI_OK;
  void* ws;
  int rc = capi_ws_get(h, sizeof(double) * (size_t)m * (size_t)n, &ws);
  if (rc != CAPI_OK) return rc;
  rc = trmm_launch(h, side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb, 0.0, (double*)ws, m, /*ws_free=*/false);
  if (rc != CAPI_OK) return rc;
  return capi_internal_copy2d(h, m, n, (const double*)ws, m, B, ldb);
}

// ---- "panel32" images of a tall panel (CholeskyQR2, n = 256) -------------------------------------------------------------------------------
// An m x 256 panel (m % 32 == 0) stored as m / 32 tiles of 32 rows, each tile column-major with ld 32, tile t at 32 * 256 * t doubles:
// element (i, j) sits at (i / 32) * 8192 + 32 j + i % 32.  A column-major tall panel is 256 column streams 8 lda bytes apart, touched
// 256 bytes at a time; a panel32 image is ONE contiguous stream.  qr::cacqr keeps the intermediate Q1 of CholeskyQR2 (written by sweep 1,
// read twice by sweep 2, never seen by the caller) in this form: three of the six passes over the panel become contiguous.
int capi_dsyrk_panel32(capi_handle_t h, int64_t n, int64_t k, double alpha, const double* A32, double beta, double* C, int64_t ldc) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, n == TSK_W && k > 0 && k % 32 == 0 && k >= 64 * n && k < (1LL << 31), "panel32 Gram matrix: n == 256, k a multiple of 32, k >= 64 n");
  CAPI_REQUIRE(h, A32 && C && ldc >= n && (((uintptr_t)A32 & 15) == 0), "operands");
  CAPI_REQUIRE(h, getenv("CAPI_NO_TS") == nullptr, "panel32 images need the full-width tall-skinny kernels (CAPI_NO_TS is set)");
  GemmArgs p{};
  p.A = A32; p.B = A32; p.C = C; p.lda = 32; p.ldb = 32; p.ldc = ldc;
  p.M = (int)n; p.N = (int)n; p.K = (int)k; p.alpha = alpha; p.beta = beta;
  p.out_uplo = CAPI_UPPER; p.tri_side = -1;
  p.a_tiled = 1;
  return launch_gemm(h, true, true, p, true);
}

// C = alpha * B * T, T 256 x 256 upper triangular (non-unit), B and C m x 256; ldb == 0: B is a panel32 image, ldc == 0: C is written as one
int capi_dtrmm_right_panel32(capi_handle_t h, int64_t m, int64_t n, double alpha, const double* T, int64_t ldt, const double* B, int64_t ldb,
                             double* C, int64_t ldc) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, n == TSK_W && m > 0 && m % 32 == 0 && m >= 64 * n && m < (1LL << 31), "panel32 right-TRMM: n == 256, m a multiple of 32, m >= 64 n");
  CAPI_REQUIRE(h, T && B && C && ldt >= n && (ldb == 0 || ldb >= m) && (ldc == 0 || ldc >= m) && (const double*)C != B, "operands");
  CAPI_REQUIRE(h, getenv("CAPI_NO_TS") == nullptr && getenv("CAPI_TS_ROWS16") == nullptr, "panel32 images need the 32-row T-stationary kernel");
  GemmArgs p{};
  p.C = C; p.ldc = ldc ? ldc : 32; p.M = (int)m; p.N = (int)n; p.K = (int)n; p.alpha = alpha; p.beta = 0.0;
  p.out_uplo = -1; p.tri_side = CAPI_RIGHT; p.tri_eff_upper = 1; p.tri_unit = 0;
  p.A = B; p.lda = ldb ? ldb : 32; p.B = T; p.ldb = ldt;
  p.a_tiled = ldb == 0; p.c_tiled = ldc == 0;
  return launch_gemm(h, false, true, p, true);
}

// which recorded launches a `variant` code of capi_prof_collect[_intervals] selects.  A record's variant is (transposed-A ? 2 : 0) +
// (k-contiguous B ? 1 : 0) [+ 4: the 64-tile kernel] [+ 16: dtrmm_pair_kernel, the 128-tile kernel's form for TRMMs in tile pairs].
//   -1: every record;  0..3: that operand orientation, any kernel;  8 + v: exactly record variant v (one kernel symbol);
//   100 + o: the 128-tile kernels of orientation o -- dgemm_tile_kernel<128, ..> and dtrmm_pair_kernel<..> (two symbols, one inner loop)
static bool prof_selected(int code, int rec) {
  if (code < 0) return true;
  if (code >= 100) return rec == code - 100 || rec == 16 + (code - 100);
  if (code >= 8) return rec == code - 8;
  return (rec & 3) == code;
}

int capi_prof_enable(capi_handle_t h, int on) {
  CAPI_REQUIRE(h, h, "null handle");
  h->prof_on = on != 0;
  if (on) {
    h->prof_n = 0;
    // device-side interval records, two 64-bit words per launch, all ones = "no workgroup has reported yet" (see stamp_begin / stamp_end)
    if (!h->d_stamps) {
      CAPI_HIP_CHECK(h, hipSetDevice(h->device));
      h->stamps_cap = 1 << 16;
      CAPI_HIP_CHECK(h, hipMalloc((void**)&h->d_stamps, sizeof(unsigned long long) * 2 * (size_t)h->stamps_cap));
      int khz = 0;
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) == hipSuccess && khz > 0) h->wall_khz = khz;
    }
    CAPI_HIP_CHECK(h, hipMemsetAsync(h->d_stamps, 0xff, sizeof(unsigned long long) * 2 * (size_t)h->stamps_cap, h->stream));
    CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));        // launches of the handle's other streams must find the records initialised
  }
  return CAPI_OK;
}

// The same records as capi_prof_collect, timed by the kernels themselves: every recorded launch carries the interval [first workgroup's
// start, last workgroup's end] in wall-clock ticks.  *union_ms = the length of the UNION of those intervals (time during which at least
// one of the selected launches had workgroups on the device), *sum_ms their sum, *max_ms the longest.  Achieved rate of a kernel in a step
// whose launches interleave across streams = total_flops / union_ms; stream-ordered event brackets (capi_prof_collect) overstate the
// durations there, because a bracket also contains the rounds of other streams that ran between its two events.
int capi_prof_collect_intervals(capi_handle_t h, int variant, int64_t* launches, double* union_ms, double* sum_ms, double* total_flops, double* max_ms) {
  CAPI_REQUIRE(h, h && launches && union_ms && sum_ms && total_flops, "args");
  *launches = 0; *union_ms = 0; *sum_ms = 0; *total_flops = 0;
  if (max_ms) *max_ms = 0;
  if (!h->d_stamps || h->prof_n == 0) return CAPI_OK;
  CAPI_HIP_CHECK(h, hipSetDevice(h->device));
  int rc = capi_sync(h);
  if (rc != CAPI_OK) return rc;
  const int n = h->prof_n < h->stamps_cap ? h->prof_n : h->stamps_cap;
  unsigned long long* st = (unsigned long long*)malloc(sizeof(unsigned long long) * 2 * (size_t)n);
  if (!st) return CAPI_ENOMEM;
  hipError_t e = hipMemcpy(st, h->d_stamps, sizeof(unsigned long long) * 2 * (size_t)n, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(st); CAPI_HIP_CHECK(h, e); }
  struct iv { unsigned long long a, b; };
  iv* v = (iv*)malloc(sizeof(iv) * (size_t)n);
  if (!v) { free(st); return CAPI_ENOMEM; }
  int m = 0;
  const double ms_per_tick = 1.0 / (double)h->wall_khz;
  static const bool dump = getenv("CAPI_PROF_DUMP") != nullptr;
  for (int i = 0; i < n; ++i) {
    if (!prof_selected(variant, h->prof[i].variant)) continue;
    const unsigned long long a = st[2 * i], b = ~st[2 * i + 1];
    if (st[2 * i] == ~0ull || st[2 * i + 1] == ~0ull || b < a) continue;       // (a launch without workgroups, or not yet run)
    v[m++] = {a, b};
    const double ms = (double)(b - a) * ms_per_tick;
    *launches += 1; *sum_ms += ms; *total_flops += h->prof[i].flops;
    if (max_ms && ms > *max_ms) *max_ms = ms;
    if (dump && variant < 0)
      fprintf(stderr, "[capi prof iv] %4d %s v%d M=%d N=%d K=%d  start %.3f ms  %9.3f ms  %6.2f TF/s\n", i, h->prof[i].kind == 0 ? "gemm" : h->prof[i].kind == 1 ? "syrk" : "trmm",
              h->prof[i].variant, h->prof[i].m, h->prof[i].n, h->prof[i].k, (double)(a - st[0]) * ms_per_tick, ms, h->prof[i].flops / ms * 1e-9);
  }
  // union of the intervals: sort by start, merge
  for (int i = 1; i < m; ++i) { iv x = v[i]; int j = i - 1; while (j >= 0 && v[j].a > x.a) { v[j + 1] = v[j]; --j; } v[j + 1] = x; }   // (nearly sorted already)
  unsigned long long covered = 0, cur_a = 0, cur_b = 0;
  for (int i = 0; i < m; ++i) {
    if (i == 0 || v[i].a > cur_b) { covered += cur_b - cur_a; cur_a = v[i].a; cur_b = v[i].b; }
    else if (v[i].b > cur_b) cur_b = v[i].b;
  }
  covered += cur_b - cur_a;
  *union_ms = (double)covered * ms_per_tick;
  free(v); free(st);
  return CAPI_OK;
}

// variant: 0 NN-like (row-contiguous A, row-contiguous B) .. 3 (k-contiguous A and B, the TN kernel of the trailing update); -1 all
int capi_prof_collect(capi_handle_t h, int variant, int64_t* launches, double* total_ms, double* total_flops, double* max_ms) {
  CAPI_REQUIRE(h, h && launches && total_ms && total_flops, "args");
  CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
  *launches = 0; *total_ms = 0; *total_flops = 0;
  if (max_ms) *max_ms = 0;
  for (int i = 0; i < h->prof_n; ++i) {
    // 0..3: operand orientations, any tile size;  8 + v: exactly variant v (bit 2 set = 64-tile kernel)
    if (!prof_selected(variant, h->prof[i].variant)) continue;
    float ms = 0;
    CAPI_HIP_CHECK(h, hipEventElapsedTime(&ms, h->prof[i].e0, h->prof[i].e1));
    *launches += 1; *total_ms += ms; *total_flops += h->prof[i].flops;
    if (max_ms && ms > *max_ms) *max_ms = ms;
    static const bool dump = getenv("CAPI_PROF_DUMP") != nullptr;       // diagnostics: one line per recorded launch
    if (dump && variant < 0)
      fprintf(stderr, "[capi prof] %4d %s v%d M=%d N=%d K=%d  %9.3f ms  %6.2f TF/s\n", i, h->prof[i].kind == 0 ? "gemm" : h->prof[i].kind == 1 ? "syrk" : "trmm",
              h->prof[i].variant, h->prof[i].m, h->prof[i].n, h->prof[i].k, ms, h->prof[i].flops / ms * 1e-9);
  }
  return CAPI_OK;
}

int capi_mfma_f64_peak(capi_handle_t h, int iters, double* tflops) {
  CAPI_REQUIRE(h, h && tflops && iters > 0, "args");
  const char* bpc = getenv("CAPI_PEAK_BLOCKS_PER_CU");
  const int blocks = h->num_cu * (bpc ? atoi(bpc) : 2);  // 4 waves per block: two blocks per CU = two waves per SIMD (as the tile kernel runs)
  double* out;
  CAPI_HIP_CHECK(h, hipMalloc((void**)&out, sizeof(double) * blocks * 256));
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, h->stream, out, 16);  // warm-up
  CAPI_HIP_CHECK(h, hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, h->stream, out, iters);
  CAPI_HIP_CHECK(h, hipEventRecord(h->ev1, h->stream));
  CAPI_HIP_CHECK(h, hipEventSynchronize(h->ev1));
  float ms = 0;
  CAPI_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  CAPI_HIP_CHECK(h, hipFree(out));
  const double flops = (double)blocks * 4.0 /*waves*/ * (double)iters * 8.0 * 2048.0;
  *tflops = flops / ((double)ms * 1e-3) / 1e12;
  return CAPI_OK;
}

}  // extern "C"
