// factor_f64.hip -- panel POTRF / TRTRI / TRSM for gfx950.
//
// Replaces LAPACKE_dpotrf / LAPACKE_dtrtri behind the reference's lapack::engine
// (src/lapack/interface.hpp:30-58) and adds the block TRSM the reference lacks.
//
// Structure: one LDS-resident MFMA leaf kernel factors AND inverts a diagonal block of order <= 128 in a single
// launch (the reference's base case is exactly this pair: potrf, memcpy, trtri -- cholinv/policy.h:199-201,
// cacqr.hpp:20-22); everything larger is the same recursion the reference runs across MPI ranks
// (cholinv.hpp:87-165), executed here on one device with the MFMA tile kernel of gemm_f64.hip:
//   R11,X11 = leaf/rec(A11);  R12 = X11^T A12;  A22 -= R12^T R12;  R22,X22 = rec(A22);  X12 = -X11 R12 X22.
#include "capi_internal.h"

int capi_ws2_get(capi_handle_t h, size_t bytes, void** p);
extern "C" int capi_internal_trmm_oop_batched(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
                                              const double* T, int64_t ldt, int64_t st, const double* B, int64_t ldb, int64_t sb,
                                              double* C, int64_t ldc, int64_t sc, int batch);

namespace {

constexpr int LEAF = 128;

// ------------------------------------------------------------------------------------------------------------------
// MFMA-blocked leaf: one workgroup factors AND inverts a diagonal block of order b <= 128 held in LDS (128 x 129 doubles: an odd
// leading dimension, 8-byte fragment accesses without bank conflicts).  Panels are 16 wide (the fp64 MFMA tile):
//   diag(k)     the 16 x 16 diagonal tile, factored and inverted by ONE wave in registers (leaf_diag_mfma: four block steps of width 4);
//   panel(k)    R_kc = V_k * A_kc        for the tiles right of the diagonal        (MFMA, one tile per wave; wave 0 takes the tile right of the
//               diagonal and goes on to update the NEXT diagonal tile from that tile's accumulators, keeping the result in registers)
//   trailing(k) A_rc -= R_kr^T * R_kc    for k < r <= c                              (MFMA; wave 0 factors the next diagonal tile instead, waves 1-3
//               and 5-7 take the tiles -- wave 4 shares wave 0's SIMD and stays idle)
// then the inverse by halving, in place over the factor (which has already gone to HBM): X12 = -(X11 R12) X22 at tile-block sizes 1, 2, 4.
// The chain of the eight diagonal tiles is the critical path: ~4.2 k cycles of one wave's dependent arithmetic per tile (16 pivots: v_rsq_f64 and one
// third-order correction, column scalings, four MFMA round trips) + a panel phase of 0.65-1.9 k between two tiles; load 7 k, stores 4.3 + 4.2 k,
// inverse 20 k (its last level bound by the one CU's MFMA rate).  profiles/r4_leaf_branch_free_pivots.txt has the history, cycle by cycle.
typedef double d4l_t __attribute__((ext_vector_type(4)));
typedef double d2l_t __attribute__((ext_vector_type(2)));
#ifndef LEAF_LLD
#define LEAF_LLD 129
#endif
constexpr int LB = 128, LLD = LEAF_LLD, LNT = 8;
// 8 waves (2 per SIMD): every phase but the diagonal tile is instruction-issue bound on one wave per SIMD (a 16 x 16 tile
// is 4 MFMAs behind ~100 address/copy instructions), so a second wave per SIMD nearly doubles those phases' rate
constexpr int LEAF_THREADS = 512, LNW = LEAF_THREADS / 64;

__device__ __forceinline__ double readlane_f64(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

// The 16 x 16 diagonal tile, factored AND inverted by ONE wave with the tile in MFMA accumulator layout (the other
// waves work on the trailing tiles meanwhile, or wait at the barrier that follows).  Four block steps of width 4 instead of sixteen scalar steps:
//   T (symmetric, both triangles kept) and Z (starts as I, ends as V = U^-T) live as D-layout registers: lane (c = l&15,
//   g = l>>4), register r  <->  element [g + 4r][c].  Rows kb..kb+3 are then register kb/4 -- which, read as an MFMA
//   operand, is exactly the 16 x 4 panel T[:,kb..kb+3] (by symmetry) resp. Z^T's panel: no data movement at all.
//   1. the 4 x 4 diagonal block travels by v_readlane; every lane factors it and inverts it (W = L4^-1), uniformly;
//   2. panels   Lp = T[:,kb:kb+4] W^T,  Ep = E[:,kb:kb+4] W^T     one MFMA each  (A = W padded to 16 x 4, B = the register)
//   3. updates  T -= Lp Lp^T,  Z -= Lp Ep^T                        one MFMA each
//   4. rows kb..kb+3 of T and Z become Lp and Ep (U's rows, V's rows).
// The dependency chain is 4 x (four pivots + two MFMA latencies) instead of 16 x (LDS round trip + barrier): 4.2 k cycles
// per tile against ~9.5 k.  GIVEN: the tile already holds a finished triangular factor U; only V is formed.  IN_REGS: the
// tile arrives in the caller's accumulators (that IS this layout; the caller keeps diagonal tiles symmetric).
template <bool GIVEN, bool IN_REGS = false>
__device__ __forceinline__ void leaf_diag_mfma(double* __restrict__ M, double* __restrict__ Vs, int k, int lane, bool unit,
                                               int* __restrict__ info, int info_base, int b, d4l_t Tin = d4l_t{0.0, 0.0, 0.0, 0.0}) {
  const int c = lane & 15, g = lane >> 4, k0 = 16 * k;
  unsigned bad_mask = 0;                                 // bit i: pivot i of this tile was not positive (one bit operation per pivot on the chain; decoded behind the tile)
  d4l_t T, Z;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = g + 4 * r;
    const int lo = row < c ? row : c, hi = row < c ? c : row;
    double t;
    if (IN_REGS) {
      t = Tin[r];                                        // the caller's accumulators ARE this layout, and hold both triangles (the diagonal tiles are kept symmetric)
    } else {
      t = M[(k0 + lo) + (k0 + hi) * LLD];                // upper triangle is authoritative: symmetrise on the way in
      if (GIVEN) {
        t = row <= c ? M[(k0 + row) + (k0 + c) * LLD] : 0.0;
        if (unit && row == c) t = 1.0;
      }
    }
    T[r] = t;
    Z[r] = row == c ? 1.0 : 0.0;
  }
#pragma unroll
  for (int kb4 = 0; kb4 < 4; ++kb4) {
    const int kb = 4 * kb4;
    // 4 x 4 diagonal block, lower triangle d[a][bb] = T[kb+a][kb+bb]: lane 16a + kb + bb, register kb4
    double d[4][4];
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
      for (int bb = 0; bb <= a_; ++bb)
        d[a_][bb] = GIVEN ? readlane_f64(T[kb4], 16 * bb + kb + a_)      // L4[a][bb] = U[kb+bb][kb+a]
                          : readlane_f64(T[kb4], 16 * a_ + kb + bb);
    double l[4][4], y[4], w[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double p = d[s][s];
      if (!GIVEN) {
#pragma unroll
        for (int m = 0; m < s; ++m) p -= l[s][m] * l[s][m];
        // (branch-free: every lane holds the same p, but the compiler cannot know -- an `if` here is an exec-mask round trip and a taken branch
        //  on the critical chain of every one of the 16 pivots; the first bad pivot is reported once, behind the tile)
        const bool bad_p = !(p > 0.0);
        bad_mask |= bad_p ? 1u << (kb + s) : 0u;
        p = bad_p ? 1.0 : p;                               // (recording without replacing was measured: the shorter chain schedules worse, leaf 86.3 k -> 87.7 k cycles)
        // 1/sqrt(p): v_rsq_f64 (good to ~2^-23) and ONE third-order step, y (15 - 10 z + 3 z^2) / 8 with z = p y^2, written in the residual e = 1 - z:
        // y + (y e) (1/2 + 3/8 e) -- error (5/2) e^3, below 2^-66 before rounding; four dependent operations instead of the six of two Newton steps
        // (the IEEE sqrt / div sequences cost ~10x more per pivot)
        double ys = __builtin_amdgcn_rsq(p);
        const double e = __builtin_fma(-(p * ys), ys, 1.0);
        ys = __builtin_fma(ys * e, __builtin_fma(0.375, e, 0.5), ys);
        y[s] = ys;
#pragma unroll
        for (int a_ = s + 1; a_ < 4; ++a_) {
          double x = d[a_][s];
#pragma unroll
          for (int m = 0; m < s; ++m) x -= l[a_][m] * l[s][m];
          l[a_][s] = x * ys;
        }
      } else {
        p = p == 0.0 ? 1.0 : p;
        y[s] = 1.0 / p;                                  // row s of U is final: y = 1/u_ss
#pragma unroll
        for (int a_ = s + 1; a_ < 4; ++a_) l[a_][s] = d[a_][s];
      }
    }
    // W = L4^-1 (lower), uniform
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_) {
      w[a_][a_] = y[a_];
#pragma unroll
      for (int bb = 0; bb < a_; ++bb) {
        double x = 0.0;
#pragma unroll
        for (int m = bb; m < a_; ++m) x += l[a_][m] * w[m][bb];
        w[a_][bb] = -x * y[a_];
      }
    }
    // A operand of the panel products: lane (i' = c, kk = g) supplies Wpad[i'][kk] (rows >= 4 and kk > i' are zero)
    double wa = 0.0;
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
      for (int bb = 0; bb <= a_; ++bb)
        if (c == a_ && g == bb) wa = w[a_][bb];
    const d4l_t zero4 = {0.0, 0.0, 0.0, 0.0};
    // panels: register 0 of W * (rows kb.. as B operand) is lane (i = c, g = column of the panel)
    double Lp;
    if (GIVEN) {
      Lp = T[kb4];                                       // U[kb+g][c] = L[c][kb+g]
    } else {
      const d4l_t pr = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, T[kb4], zero4, 0, 0, 0);
      Lp = pr[0];
    }
    if (c < kb + g) Lp = 0.0;                            // L is lower triangular; rows above the block are finished
    const d4l_t er = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, Z[kb4], zero4, 0, 0, 0);
    const double Ep = er[0];
    if (!GIVEN) T = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp, Lp, T, 0, 0, 0);
    Z = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp, Ep, Z, 0, 0, 0);
    T[kb4] = Lp;
    Z[kb4] = Ep;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = g + 4 * r;
    M[(k0 + row) + (k0 + c) * LLD] = row <= c ? T[r] : 0.0;      // U, zeros below the diagonal
    Vs[k * 256 + row + c * 16] = c <= row ? Z[r] : 0.0;          // V[row][c], stored [col][row]-major with ld 16
  }
  if (!GIVEN && bad_mask != 0 && lane == 0) {
    const int first = __ffs((int)bad_mask) - 1;
    if (k0 + first < b) atomicCAS(info, 0, info_base + k0 + first + 1);
  }
}

// upper triangle of the LDS tile to global (zeros below the diagonal on request): the LDS reads of a round are issued
// together, then its stores (16-byte stores of row pairs when the block allows)
__device__ __forceinline__ void leaf_store_upper(const double* __restrict__ M, double* __restrict__ G, int64_t ldg, int b,
                                                 int zero_lower, int tid) {
  if ((((uintptr_t)G & 15) == 0) && ((ldg & 1) == 0) && ((b & 1) == 0)) {
    const int rp = 2 * (tid & 63), cq = tid >> 6;
#pragma unroll
    for (int half = 0; half < 128 / (16 * LNW); ++half) {
      d2l_t v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int col = cq + LNW * (16 * half + q);
        d2l_t x = {M[rp + col * LLD], M[rp + 1 + col * LLD]};
        if (rp > col) x.x = 0.0;
        if (rp + 1 > col) x.y = 0.0;
        v[q] = x;
      }
      if (rp < b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int col = cq + LNW * (16 * half + q);
          if (col < b) {
            if (rp + 1 <= col || zero_lower) *(d2l_t*)&G[rp + (int64_t)col * ldg] = v[q];
            else if (rp <= col) G[rp + (int64_t)col * ldg] = v[q].x;
          }
        }
      }
    }
    return;
  }
  constexpr int NH = LEAF_THREADS / 128;
  const int row = tid & 127, half = tid >> 7;
  for (int base = 0; base < b; base += 16 * NH) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int col = base + half + NH * q;
      v[q] = (row <= col && col < LB) ? M[row + col * LLD] : 0.0;
    }
    if (row < b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int col = base + half + NH * q;
        if (col < b && (row <= col || zero_lower)) G[row + (int64_t)col * ldg] = v[q];
      }
    }
  }
}

// side job of a leaf launch: workgroups 1.. copy a rows x cols block (the row panel right of the diagonal block) into a
// compact buffer while workgroup 0 factors -- the panel product that follows then writes R12 straight into place
struct LeafCopy {
  const double* src;
  double* dst;
  int64_t lds, ldd, cols;
  int rows;
};

__global__ __launch_bounds__(LEAF_THREADS, 1) void potrf_trtri_leaf128_kernel(double* __restrict__ A, int64_t lda, double* __restrict__ X,
                                                                      int64_t ldx, int b, int want_inv, int zero_lower,
                                                                      int invert_only, int unit, int* __restrict__ info,
                                                                      int info_base, long long* __restrict__ dbg, LeafCopy cp) {
  extern __shared__ __attribute__((aligned(16))) double lds_leaf[];
  if (blockIdx.x > 0) {
    const int r = threadIdx.x & (LEAF - 1), c0 = threadIdx.x / LEAF;
    constexpr int CPW = LEAF_THREADS / LEAF;      // columns per pass
    if (r < cp.rows)
      for (int64_t c = (int64_t)(blockIdx.x - 1) * CPW + c0; c < cp.cols; c += (int64_t)(gridDim.x - 1) * CPW)
        cp.dst[r + c * cp.ldd] = cp.src[r + c * cp.lds];
    return;
  }
  int dbg_n = 0;
#define LEAF_MARK() do { if (dbg && threadIdx.x == 0) dbg[dbg_n++] = clock64(); } while (0)
  LEAF_MARK();
  double* M = lds_leaf;                       // LB x LLD
  double* Vs = M + LB * LLD;                  // LNT tiles of 16 x 16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: wave-level loops stay scalar
  const int r16 = lane & 15, g = lane >> 4;
  const int nk = (b + 15) >> 4;               // active 16-wide panels
  int tri_tab;                                // lane t: (lo, hi) of pair t in the enumeration t = hi (hi + 1) / 2 + lo
  {
    int hi_ = 0;
    while ((hi_ + 1) * (hi_ + 2) / 2 <= lane) ++hi_;
    tri_tab = (lane - hi_ * (hi_ + 1) / 2) | (hi_ << 4);
  }

  // load (upper triangle; identity padding beyond b keeps the padded problem SPD).  Every load of the tile is in flight
  // before the first LDS store (a load -> store loop exposed one HBM latency per column): 32 x 16 bytes per thread when
  // the block is 16-byte aligned, 4 rounds of 16 x 8 bytes otherwise.
  const int nrc = 16 * nk;
  if ((((uintptr_t)A & 15) == 0) && ((lda & 1) == 0) && ((b & 1) == 0)) {
    const int rp = 2 * (tid & 63), cq = tid >> 6;
    constexpr int NIT = 128 / LNW;
    d2l_t v[NIT];
    // unconditional loads from a clamped (always valid) address, selects afterwards: a load inside a branch makes the
    // compiler wait for it at the join, one exposed latency per column
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int col = cq + LNW * it;
      const bool in = rp < b && col < b && rp <= col;
      v[it] = *(const d2l_t*)(in ? &A[rp + (int64_t)col * lda] : A);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int col = cq + LNW * it;
      d2l_t x = v[it];
      if (rp < b && col < b) {
        if (rp > col) x.x = 0.0;
        if (rp + 1 > col) x.y = 0.0;
      } else {
        x = (d2l_t){rp == col ? 1.0 : 0.0, rp + 1 == col ? 1.0 : 0.0};
      }
      v[it] = x;
    }
    if (rp < nrc) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int col = cq + LNW * it;
        if (col < nrc) { M[rp + col * LLD] = v[it].x; M[rp + 1 + col * LLD] = v[it].y; }   // (odd LLD: 8-byte LDS accesses)
      }
    }
  } else {
    constexpr int NH = LEAF_THREADS / 128;
    const int row = tid & 127, half = tid >> 7;
    for (int base = 0; base < nrc; base += 16 * NH) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int col = base + half + NH * q;
        double x = (row == col) ? 1.0 : 0.0;
        if (row < b && col < b) x = (row <= col) ? A[row + (int64_t)col * lda] : 0.0;
        v[q] = x;
      }
      if (row < nrc) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int col = base + half + NH * q;
          if (col < nrc) M[row + col * LLD] = v[q];
        }
      }
    }
  }
  __syncthreads();
  LEAF_MARK();

  if (invert_only) {
    // the diagonal tiles of a given factor are independent: one per wave
    for (int k = wave; k < nk; k += LNW) leaf_diag_mfma<true>(M, Vs, k, lane, unit != 0, info, info_base, b);
    __syncthreads();
    LEAF_MARK();
  } else {
    if (wave == 0) {
      leaf_diag_mfma<false>(M, Vs, 0, lane, false, info, info_base, b);
    } else {
      // Meanwhile the idle waves make the OTHER diagonal 16 x 16 tiles symmetric in LDS (the load keeps the upper triangle only; mirrored loads from HBM were
      // measured: +3.5 k cycles of uncoalesced reads).  Every trailing update of such a tile is a symmetric product, so it stays symmetric, and wave 0 hands the
      // updated tile to its factorisation in registers -- no store, no symmetrising re-read between two links of the chain.
      for (int e = tid - 64; e < (nk - 1) * 256; e += LEAF_THREADS - 64) {
        const int kt = 1 + (e >> 8), i = e & 15, j = (e >> 4) & 15;
        if (i > j) M[(16 * kt + i) + (16 * kt + j) * LLD] = M[(16 * kt + j) + (16 * kt + i) * LLD];
      }
    }
    __syncthreads();
    LEAF_MARK();
    d4l_t next_diag = {0.0, 0.0, 0.0, 0.0};       // wave 0: tile (k+1, k+1) with step k's update, in accumulator layout
    for (int k = 0; k + 1 < nk; ++k) {
      const int k0 = 16 * k;
      // panel: R_kc = V_k * A_kc
      for (int c = k + 1 + wave; c < nk; c += LNW) {
        const int c0 = 16 * c;
        d4l_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int m = 4 * st + g;
          const double av = Vs[k * 256 + r16 + m * 16];                 // V[i=r16][m]
          const double bv = M[(k0 + m) + (c0 + r16) * LLD];             // A_kc[m][j=r16]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) M[(k0 + g + 4 * q) + (c0 + r16) * LLD] = acc[q];
        if (wave == 0 && c == k + 1) {
          // Wave 0's first tile is R_k,k+1, all that the update of the NEXT diagonal tile needs -- and as an MFMA operand R[4 st + g][r16] is lane (r16, g), register st:
          // the accumulators just computed.  The update is done here, while the other waves (two to a SIMD) still work on their panel tiles, instead of behind the barrier.
          d4l_t u, u2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; ++q) u[q] = M[(c0 + g + 4 * q) + (c0 + r16) * LLD];
          u = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[0], acc[0], u, 0, 0, 0);
          u2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[2], acc[2], u2, 0, 0, 0);
          u = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[1], acc[1], u, 0, 0, 0);
          u2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[3], acc[3], u2, 0, 0, 0);
          next_diag = u + u2;
        }
      }
      __syncthreads();
      LEAF_MARK();
      // trailing: A_rc -= R_kr^T R_kc, k < r <= c, with one step of lookahead: wave 0 updates the next diagonal tile
      // (pair 0 of the enumeration) and goes straight on to factor it -- the one-wave, latency-bound part of the next
      // step -- while waves 1.. update the other tiles.  The operands of a wave's NEXT tile are read while the MFMAs of
      // the current one run (a tile is only 4 MFMAs: unpipelined, LDS latency and accumulator round trip dominate).
      {
        const int nt = nk - 1 - k, ntiles = nt * (nt + 1) / 2;
        // (lo, hi) of tile t from a per-lane table (lane t holds pair t of the triangular enumeration): one v_readlane
        // instead of ~70 scalar instructions of index stepping per tile -- these loops are instruction-issue bound
        int lo, hi;
        auto coords = [&](int t_) { const int e = __builtin_amdgcn_readlane(tri_tab, t_ < 63 ? t_ : 63); lo = e & 15; hi = e >> 4; };
        struct frag { d4l_t acc; double av[4], bv[4]; };
        auto load = [&](int lo_, int hi_) {
          frag f;
          const int r0 = 16 * (k + 1 + lo_), c0 = 16 * (k + 1 + hi_);
#pragma unroll
          for (int q = 0; q < 4; ++q) f.acc[q] = M[(r0 + g + 4 * q) + (c0 + r16) * LLD];
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            const int m = 4 * st + g;
            f.av[st] = -M[(k0 + m) + (r0 + r16) * LLD];               // -R_kr[m][i=r16]
            f.bv[st] = M[(k0 + m) + (c0 + r16) * LLD];                //  R_kc[m][j=r16]
          }
          return f;
        };
        if (wave == 0) leaf_diag_mfma<false, true>(M, Vs, k + 1, lane, false, info, info_base, b, next_diag);   // (updated in the panel phase, still in registers)
        // waves 1, 2, 3, 5, 6, 7 take the other tiles; wave 4 sits on wave 0's SIMD and stays out of its way (the chain's VALU and MFMA issue share that SIMD)
        const int stride = LNW - 2;
        int t = (wave == 0 || wave == 4) ? ntiles : (wave < 4 ? wave : wave - 1);
        coords(t);
        frag cur;
        if (t < ntiles) cur = load(lo, hi);
        while (t < ntiles) {
          const int clo = lo, chi = hi;
          const bool more = t + stride < ntiles;
          if (more) coords(t + stride);
          const frag nxt = load(more ? lo : clo, more ? hi : chi);     // (unconditional: a re-read of this tile when it is the last)
          // two independent accumulator chains (k-steps 0,1 and 2,3)
          d4l_t acc = cur.acc, acc2 = {0.0, 0.0, 0.0, 0.0};
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.av[0], cur.bv[0], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.av[2], cur.bv[2], acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.av[1], cur.bv[1], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.av[3], cur.bv[3], acc2, 0, 0, 0);
          acc += acc2;
          const int r0 = 16 * (k + 1 + clo), c0 = 16 * (k + 1 + chi);
#pragma unroll
          for (int q = 0; q < 4; ++q) M[(r0 + g + 4 * q) + (c0 + r16) * LLD] = acc[q];
          cur = nxt;
          t += stride;
        }
      }
      __syncthreads();
      LEAF_MARK();
    }
  }

  if (!invert_only) leaf_store_upper(M, A, lda, b, zero_lower, tid);

  LEAF_MARK();
  if (want_inv) {
    // X = R^-1 in place over the factor (which has already gone to HBM), by halving: with X11 and X22 known,
    // X12 = -(X11 R12) X22.  Levels h = 16, 32, 64 (tiles per block s = 1, 2, 4); every off-diagonal block of a level is
    // independent of the others and each of its two products is a set of s*s independent 16 x 16 tiles -- 6 wide phases
    // instead of the 8 x 3 narrow ones of a column-by-column substitution.  Both products overwrite their left operand's
    // block, so a wave keeps its (at most two) tiles in registers until a barrier has seen every read.
    __syncthreads();                                               // the factor's store above has read every tile
    {
      const int i = tid & 15, q = (tid >> 4) & 15, j = tid >> 8;     // diagonal tiles X_jj = V_j^T, 2 tiles per pass
      for (int jj = j; jj < nk; jj += LEAF_THREADS / 256) M[(16 * jj + i) + (16 * jj + q) * LLD] = i <= q ? Vs[jj * 256 + q + i * 16] : 0.0;
    }
    __syncthreads();
    LEAF_MARK();
    for (int ls = 0; (1 << ls) < nk; ++ls) {
      const int s_ = 1 << ls, per = s_ * s_, npairs = (nk + 2 * s_ - 1) >> (ls + 1), ntl = npairs * per;
      // tile t of this level: pair p, row a and column c inside the pair's off-diagonal block.  A wave's second tile
      // (only the last level has 16 tiles for 8 waves) is taken from the far end of the list: rows and columns mirror,
      // so every wave gets the same number of k-steps in both phases (s + 1 of them instead of up to 2 s).
      auto coords = [&](int t, int& ti, int& tj, int& c0) {
        const int te = t < LNW ? t : ntl - 1 - (t - LNW);
        const int p_ = te >> (2 * ls), w_ = te & (per - 1);
        const int r0 = p_ << (ls + 1);
        c0 = r0 + s_;
        ti = r0 + (w_ & (s_ - 1));
        tj = c0 + (w_ >> ls);
      };
#pragma unroll
      for (int phase = 0; phase < 2; ++phase) {
        d4l_t out[2];
        int nt_ = 0;
        for (int t = wave; t < ntl; t += LNW, ++nt_) {
          int ti, tj, c0;
          coords(t, ti, tj, c0);
          d4l_t acc = {0.0, 0.0, 0.0, 0.0};
          if (tj < nk) {
            // phase 0: T_ij = sum_{k=ti}^{c0-1} X_ik R_kj        phase 1: X_ij = -sum_{k=c0}^{tj} T_ik X_kj
            const int kb = phase == 0 ? ti : c0, ke = phase == 0 ? c0 : tj + 1;
            const double* pa = M + (16 * ti + r16) + g * LLD;              // A operand: rows of tile-row ti, k along columns
            const double* pb = M + g + (16 * tj + r16) * LLD;              // B operand: columns of tile-column tj, k along rows
            double av[4], bv[4];
            auto fetch = [&](int k, double (&a_)[4], double (&b_)[4]) {
#pragma unroll
              for (int st = 0; st < 4; ++st) {
                a_[st] = pa[(16 * k + 4 * st) * LLD];
                b_[st] = pb[16 * k + 4 * st];
              }
            };
            fetch(kb, av, bv);
            for (int k = kb; k < ke; ++k) {                          // the next step's operands are read during this step's MFMAs
              double an[4], bn[4];
              fetch(k + 1 < ke ? k + 1 : k, an, bn);
#pragma unroll
              for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[st], acc, 0, 0, 0);
#pragma unroll
              for (int st = 0; st < 4; ++st) { av[st] = an[st]; bv[st] = bn[st]; }
            }
          }
          if (nt_ == 0) out[0] = acc; else out[1] = acc;
        }
        __syncthreads();
        nt_ = 0;
        for (int t = wave; t < ntl; t += LNW, ++nt_) {
          int ti, tj, c0;
          coords(t, ti, tj, c0);
          const d4l_t acc = nt_ == 0 ? out[0] : out[1];
          if (tj < nk) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) M[(16 * ti + g + 4 * qq) + (16 * tj + r16) * LLD] = phase == 0 ? acc[qq] : -acc[qq];
          }
        }
        __syncthreads();
        LEAF_MARK();
      }
    }
    leaf_store_upper(M, X, ldx, b, zero_lower, tid);
  }
  LEAF_MARK();
#undef LEAF_MARK
}

__global__ void scale2d_kernel(double* __restrict__ B, int64_t ldb, int64_t m, int64_t n, double alpha) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int64_t j = blockIdx.y; j < n; j += gridDim.y) B[i + j * ldb] = alpha == 0.0 ? 0.0 : alpha * B[i + j * ldb];
}

int64_t split_point(int64_t n) {
  int64_t h = ((n / 2 + LEAF - 1) / LEAF) * LEAF;
  if (h >= n) h = (n / 2) & ~(int64_t)1;
  if (h <= 0) h = n / 2;
  return h;
}

int leaf_launch(capi_handle_t h, double* A, int64_t lda, double* X, int64_t ldx, int b, int want_inv, int zero_lower,
                int invert_only, int unit, int info_base, const LeafCopy* side = nullptr) {
  {
    LeafCopy cp{nullptr, nullptr, 0, 0, 0, 0};
    unsigned blocks = 1;
    if (side && side->cols > 0 && side->rows > 0) {
      cp = *side;
      const int64_t want = cdiv(cp.cols, 32);       // >= 8 passes of LEAF_THREADS / LEAF columns per workgroup
      blocks += (unsigned)(want < 192 ? want : 192);
    }
    const size_t lds_bytes = sizeof(double) * (LB * LLD + LNT * 256 + 4 * 16);
    CAPI_RAISE_LDS_LIMIT(h, CAPI_ATTR_LEAF, potrf_trtri_leaf128_kernel, lds_bytes);
    static const bool trace = getenv("CAPI_LEAF_TRACE") != nullptr;
    long long* dbg = nullptr;
    if (trace) CAPI_HIP_CHECK(h, hipMalloc((void**)&dbg, sizeof(long long) * 64));
    hipLaunchKernelGGL(potrf_trtri_leaf128_kernel, dim3(blocks), dim3(LEAF_THREADS), lds_bytes, h->stream, A, lda, X, ldx, b, want_inv, zero_lower,
                       invert_only, unit, h->d_info, info_base, dbg, cp);
    if (trace) {   // diagnostics only: phase timestamps (shader clock) of this launch
      long long t[64];
      CAPI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
      CAPI_HIP_CHECK(h, hipMemcpy(t, dbg, sizeof(t), hipMemcpyDeviceToHost));
      CAPI_HIP_CHECK(h, hipFree(dbg));
      fprintf(stderr, "[leaf b=%d]", b);
      for (int i = 1; i < 26; ++i) fprintf(stderr, " %lld", t[i] - t[i - 1]);
      fprintf(stderr, "  total %lld\n", t[25] - t[0]);

    }
  }
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

#define RC(x) do { int rc__ = (x); if (rc__ != CAPI_OK) return rc__; } while (0)

// cholinv on one device: A(upper) -> R, X <- R^-1 (upper).  W: n/2 x n/2-ish scratch (ld = ldw) for R12 products.
int potrf_trtri_rec(capi_handle_t h, int64_t n, double* A, int64_t lda, double* X, int64_t ldx, int zero_lower, int info_base) {
  if (n <= LEAF) return leaf_launch(h, A, lda, X, ldx, (int)n, 1, zero_lower, 0, 0, info_base);
  const int64_t n1 = split_point(n), n2 = n - n1;
  double* A12 = A + n1 * lda;
  double* A22 = A + n1 + n1 * lda;
  double* X12 = X + n1 * ldx;
  double* X22 = X + n1 + n1 * ldx;
  RC(potrf_trtri_rec(h, n1, A, lda, X, ldx, zero_lower, info_base));
  // R12 = X11^T A12 : staged through X12's storage (free until step 5), then copied into place
  RC(capi_dtrmm_oop(h, CAPI_LEFT, CAPI_UPPER, CAPI_TRANS, CAPI_NONUNIT, n1, n2, 1.0, X, ldx, A12, lda, X12, ldx));
  RC(capi_dlacpy(h, 0, n1, n2, X12, ldx, A12, lda));
  RC(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, n2, n1, -1.0, A12, lda, A12, lda, 1.0, A22, lda));
  RC(potrf_trtri_rec(h, n2, A22, lda, X22, ldx, zero_lower, info_base + (int)n1));
  // X12 = -X11 * R12 * X22 : T = X11*R12 into the strictly-lower-free scratch below, then X12 = -T*X22
  void* w;
  RC(capi_ws2_get(h, sizeof(double) * (size_t)n1 * (size_t)n2, &w));
  RC(capi_dtrmm_oop(h, CAPI_LEFT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n1, n2, 1.0, X, ldx, A12, lda, (double*)w, n1));
  RC(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n1, n2, -1.0, X22, ldx, (double*)w, n1, X12, ldx));
  return CAPI_OK;
}

// strictly-lower triangles of both outputs in one pass (nothing in the recursion reads or writes below the diagonal)
__global__ void trizero2_kernel(int64_t n, double* __restrict__ A, int64_t lda, double* __restrict__ X, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int64_t j = blockIdx.y; j < n && j < i; j += gridDim.y) {
    A[i + j * lda] = 0.0;
    X[i + j * ldx] = 0.0;
  }
}

// The same pair (R, R^-1) for a block of moderate order by a schedule with a shorter dependency chain than the halving
// recursion above.  Every launch below the order ~2048 is latency-bound, so what counts is the NUMBER of dependent
// launches: the recursion needs, per 128 columns, a leaf + TRMM + update for the factor AND two TRMMs per tree node for
// the inverse, all in one chain (2.25 ms at order 2048).  Here
//   phase 1  right-looking blocked Cholesky, block 128: leaf (R_jj and X_jj = R_jj^-1), row panel R_j,rest = X_jj^T A_j,rest,
//            trailing update A_rest,rest -= R_j,rest^T R_j,rest -- 3 dependent launches + 1 copy per 128 columns;
//   phase 2  the inverse's off-diagonal blocks level by level (X12 = -X11 R12 X22 at block sizes 128, 256, ...): all pairs
//            of a level are independent and go out as ONE strided-batch launch per product -- 2 log2(n/128) launches.
// `W`: scratch of 128 * n + n * n / 4 doubles.
// (Round 3, measured and removed: one block of lookahead in phase 1 -- the update cut into U1 = the next diagonal block and the rest, the
//  NEXT leaf on a side stream beside that rest, the panel formed out of place and written back behind the leaf.  Per step the chain is then
//  panel + U1 + max(rest of the update, leaf) instead of leaf + panel + update, on paper 59 against 95 us.  Parity-green, and slower:
//  order 1024 509 -> 690 us, 4096 3.11 -> 3.40 ms, the n = 32768 step 234 -> 272 ms.  Two cross-stream hand-offs per 128 columns (event
//  record on one HIP stream, wait on the other) cost more than the 41 us leaf they hide: a dependency between two hardware queues is
//  resolved by barrier packets and signal polling, an order of magnitude slower than the next packet of the same queue.  profiles/r4c_*.)
int potrf_trtri_blocked(capi_handle_t h, int64_t n, double* A, int64_t lda, double* X, int64_t ldx, int info_base, double* W) {
  const int64_t nblk = cdiv(n, (int64_t)LEAF);
  for (int64_t j = 0; j < nblk; ++j) {
    const int64_t j0 = j * LEAF, jb = (n - j0) < LEAF ? (n - j0) : LEAF, rest = n - j0 - jb;
    double* Ajj = A + j0 + j0 * lda;
    double* Xjj = X + j0 + j0 * ldx;
    double* Ajr = A + j0 + (j0 + jb) * lda;
    // the row panel is copied aside by spare workgroups of the diagonal block's launch; the panel product then reads the
    // copy and writes R12 in place (no copy kernel on the critical chain)
    const LeafCopy side{Ajr, W, lda, jb, rest, (int)jb};
    RC(leaf_launch(h, Ajj, lda, Xjj, ldx, (int)jb, 1, 0, 0, 0, info_base + (int)j0, &side));
    if (rest > 0) {
      RC(capi_dtrmm_oop(h, CAPI_LEFT, CAPI_UPPER, CAPI_TRANS, CAPI_NONUNIT, jb, rest, 1.0, Xjj, ldx, W, jb, Ajr, lda));
      RC(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, rest, jb, -1.0, Ajr, lda, Ajr, lda, 1.0, A + (j0 + jb) + (j0 + jb) * lda, lda));
    }
  }
  double* T = W + (int64_t)LEAF * n;
  for (int64_t b = LEAF; b < n; b *= 2) {
    const int64_t full = n / (2 * b);                      // pairs whose second block is complete
    if (full > 0) {
      // pair p: X11 = X(p 2b, p 2b), R12 = A(p 2b, p 2b + b), X22 = X(p 2b + b, p 2b + b), X12 = X(p 2b, p 2b + b); T_p = T + p b b
      RC(capi_internal_trmm_oop_batched(h, CAPI_LEFT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, b, b, 1.0, X, ldx, 2 * b * (ldx + 1),
                                        A + b * lda, lda, 2 * b * (lda + 1), T, b, b * b, (int)full));
      RC(capi_internal_trmm_oop_batched(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, b, b, -1.0, X + b + b * ldx, ldx,
                                        2 * b * (ldx + 1), T, b, b * b, X + b * ldx, ldx, 2 * b * (ldx + 1), (int)full));
    }
    const int64_t p0 = full * 2 * b;                        // a last pair with a short second block
    if (p0 + b < n) {
      const int64_t n2 = n - p0 - b;
      RC(capi_dtrmm_oop(h, CAPI_LEFT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, b, n2, 1.0, X + p0 + p0 * ldx, ldx, A + p0 + (p0 + b) * lda, lda, T, b));
      RC(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, b, n2, -1.0, X + (p0 + b) + (p0 + b) * ldx, ldx, T, b,
                        X + p0 + (p0 + b) * ldx, ldx));
    }
  }
  return CAPI_OK;
}

// in-place inverse of an upper triangular matrix (non-unit or unit diagonal)
int trtri_upper_rec(capi_handle_t h, int diag, int64_t n, double* T, int64_t ldt) {
  if (n <= LEAF) return leaf_launch(h, T, ldt, T, ldt, (int)n, 1, 0, 1, diag == CAPI_UNIT, 0);
  const int64_t n1 = split_point(n), n2 = n - n1;
  double* T12 = T + n1 * ldt;
  double* T22 = T + n1 + n1 * ldt;
  RC(trtri_upper_rec(h, diag, n1, T, ldt));
  RC(trtri_upper_rec(h, diag, n2, T22, ldt));
  void* w;
  RC(capi_ws2_get(h, sizeof(double) * (size_t)n1 * (size_t)n2, &w));
  RC(capi_dtrmm_oop(h, CAPI_LEFT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n1, n2, 1.0, T, ldt, T12, ldt, (double*)w, n1));
  RC(capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, n1, n2, -1.0, T22, ldt, (double*)w, n1, T12, ldt));
  return CAPI_OK;
}

__global__ void transpose_tri_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd, int64_t n,
                                     int src_upper) {
  // dst (full n x n scratch) upper <- src lower^T  (src_upper == 0), or dst lower <- src upper^T
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
    const bool in_src = src_upper ? (i <= j) : (i >= j);
    if (in_src) dst[j + i * ldd] = src[i + j * lds_];
  }
}

static const int TRSM_LEAF = getenv("CAPI_TRSM_LEAF") ? atoi(getenv("CAPI_TRSM_LEAF")) : 256;   // diagonal blocks of this order are inverted (diagnostic override)

// E = op(T).  Left: E X = B;  Right: X E = B.  In place on B.
int trsm_rec(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, const double* T, int64_t ldt,
             double* B, int64_t ldb) {
  const int64_t nt = side == CAPI_LEFT ? m : n;
  const bool eff_upper = (uplo == CAPI_UPPER) != (trans == CAPI_TRANS);
  if (nt <= TRSM_LEAF) {
    // invert the diagonal block into scratch, then multiply
    void* w;
    RC(capi_ws2_get(h, sizeof(double) * (size_t)TRSM_LEAF * TRSM_LEAF * 2, &w));
    double* Ti = (double*)w + (size_t)TRSM_LEAF * TRSM_LEAF;  // first half: the inverse recursion's temporaries
    if (uplo == CAPI_UPPER) {
      RC(capi_dlacpy(h, 1, nt, nt, T, ldt, Ti, nt));
      RC(trtri_upper_rec(h, diag, nt, Ti, nt));
      return capi_dtrmm(h, side, CAPI_UPPER, trans, diag, m, n, 1.0, Ti, nt, B, ldb);
    } else {
      // lower: invert its transpose (upper), use with the opposite transpose flag
      dim3 grid((unsigned)cdiv(nt, 256), (unsigned)(nt < 65535 ? nt : 65535));
      hipLaunchKernelGGL(transpose_tri_kernel, grid, dim3(256), 0, h->stream, T, ldt, Ti, nt, nt, 0);
      CAPI_HIP_CHECK(h, hipGetLastError());
      RC(trtri_upper_rec(h, diag, nt, Ti, nt));
      return capi_dtrmm(h, side, CAPI_UPPER, trans == CAPI_TRANS ? CAPI_NOTRANS : CAPI_TRANS, diag, m, n, 1.0, Ti, nt, B, ldb);
    }
  }
  const int64_t n1 = split_point(nt), n2 = nt - n1;
  const double* T11 = T;
  const double* T22 = T + n1 + n1 * ldt;
  // off-diagonal block of E between index ranges 1 and 2
  const double* Toff = (uplo == CAPI_UPPER) ? T + n1 * ldt /* T12: n1 x n2 */ : T + n1 /* T21: n2 x n1 */;
  if (side == CAPI_LEFT) {
    double* B1 = B;
    double* B2 = B + n1;
    if (eff_upper) {
      // E12 = trans ? T21^T : T12
      RC(trsm_rec(h, side, uplo, trans, diag, n2, n, T22, ldt, B2, ldb));
      RC(capi_dgemm(h, trans, CAPI_NOTRANS, n1, n, n2, -1.0, Toff, ldt, B2, ldb, 1.0, B1, ldb));
      RC(trsm_rec(h, side, uplo, trans, diag, n1, n, T11, ldt, B1, ldb));
    } else {
      // E21 = trans ? T12^T : T21
      RC(trsm_rec(h, side, uplo, trans, diag, n1, n, T11, ldt, B1, ldb));
      RC(capi_dgemm(h, trans, CAPI_NOTRANS, n2, n, n1, -1.0, Toff, ldt, B1, ldb, 1.0, B2, ldb));
      RC(trsm_rec(h, side, uplo, trans, diag, n2, n, T22, ldt, B2, ldb));
    }
  } else {
    double* B1 = B;
    double* B2 = B + n1 * ldb;
    if (eff_upper) {
      RC(trsm_rec(h, side, uplo, trans, diag, m, n1, T11, ldt, B1, ldb));
      RC(capi_dgemm(h, CAPI_NOTRANS, trans, m, n2, n1, -1.0, B1, ldb, Toff, ldt, 1.0, B2, ldb));
      RC(trsm_rec(h, side, uplo, trans, diag, m, n2, T22, ldt, B2, ldb));
    } else {
      RC(trsm_rec(h, side, uplo, trans, diag, m, n2, T22, ldt, B2, ldb));
      RC(capi_dgemm(h, CAPI_NOTRANS, trans, m, n1, n2, -1.0, B2, ldb, Toff, ldt, 1.0, B1, ldb));
      RC(trsm_rec(h, side, uplo, trans, diag, m, n1, T11, ldt, B1, ldb));
    }
  }
  return CAPI_OK;
}

bool ok01(int v) { return v == 0 || v == 1; }

}  // namespace

extern "C" {

int capi_dpotrf_trtri(capi_handle_t h, int64_t n, double* A, int64_t lda, double* Rinv, int64_t ldi) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, n >= 0 && n < (1LL << 31), "n");
  if (n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, A && Rinv && lda >= n && ldi >= n, "operands");
  if (n <= LEAF) return leaf_launch(h, A, lda, Rinv, ldi, (int)n, 1, 1, 0, 0, 0);
  void* w;
  RC(capi_ws2_get(h, sizeof(double) * ((size_t)(n / 2 + LEAF) * (size_t)(n / 2 + LEAF) + (size_t)LEAF * (size_t)n), &w));
  // up to order 4096 the blocked schedule's shorter launch chain wins; above, the recursion's large products do
  static const char* force = getenv("CAPI_POTRF_TRTRI");   // "rec" | "blocked": diagnostics
  const bool blocked = force ? force[0] == 'b' : n <= 4096;
  auto run = [&]() -> int {
    dim3 grid((unsigned)cdiv(n, 256), (unsigned)(n < 65535 ? n : 65535));
    hipLaunchKernelGGL(trizero2_kernel, grid, dim3(256), 0, h->stream, n, A, lda, Rinv, ldi);
    CAPI_HIP_CHECK(h, hipGetLastError());
    if (blocked) return potrf_trtri_blocked(h, n, A, lda, Rinv, ldi, 0, (double*)w);
    return potrf_trtri_rec(h, n, A, lda, Rinv, ldi, 0, 0);
  };
  // The blocked routine is a fixed chain of ~6 launches per 128 columns, each a few microseconds long: the second time the same
  // block comes by (same pointers, order and scratch: the next factor() of the same matrix) the chain is captured into a hipGraph,
  // from the third on it is replayed -- one submission instead of ~50 per order-1024 block.  (CAPI_GRAPH=1; off by default.)
  static const int graph_mode = getenv("CAPI_GRAPH") ? atoi(getenv("CAPI_GRAPH")) : 0;
  if (!graph_mode || !blocked || n > 2048 || h->prof_on) return run();      // (a replay records no per-launch events: profiled runs launch plainly)
  capi_handle_s::graph_ent* e = nullptr;
  for (int i = 0; i < h->graphs_n; ++i) {
    auto& g = h->graphs[i];
    if (g.n == n && g.A == A && g.X == Rinv && g.lda == lda && g.ldx == ldi && g.w == w && g.s == h->stream) { e = &g; break; }
  }
  if (!e) {
    if (h->graphs_n == h->graphs_cap) {
      const int ncap = h->graphs_cap ? 2 * h->graphs_cap : 128;
      if (ncap > 4096) return run();
      auto* ng = (capi_handle_s::graph_ent*)realloc(h->graphs, sizeof(capi_handle_s::graph_ent) * ncap);
      if (!ng) return run();
      h->graphs = ng; h->graphs_cap = ncap;
    }
    h->graphs[h->graphs_n++] = capi_handle_s::graph_ent{n, lda, ldi, A, Rinv, w, h->stream, nullptr, 1};
    if (graph_mode > 1) fprintf(stderr, "[capi graph] order %lld first seen\n", (long long)n);
    return run();
  }
  if (e->exec) {
    if (graph_mode > 1) fprintf(stderr, "[capi graph] order %lld replayed\n", (long long)n);
    CAPI_HIP_CHECK(h, hipGraphLaunch(e->exec, h->stream));
    return CAPI_OK;
  }
  if (e->seen < 0) return run();
  // capture: nothing in the chain may allocate (the first pass sized every workspace) or record profile events
  const bool prof_was = h->prof_on;
  h->prof_on = false;
  hipGraph_t graph = nullptr;
  int rc = CAPI_OK;
  const hipError_t eb = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
  if (eb != hipSuccess) {
    if (graph_mode > 1) fprintf(stderr, "[capi graph] begin capture refused: %s (stream %p)\n", hipGetErrorString(eb), (void*)h->stream);
    h->prof_on = prof_was; e->seen = -1; (void)hipGetLastError(); return run();
  }
  rc = run();
  const hipError_t ee = hipStreamEndCapture(h->stream, &graph);
  h->prof_on = prof_was;
  if (rc != CAPI_OK || ee != hipSuccess || !graph) {
    if (graph_mode > 1) fprintf(stderr, "[capi graph] capture failed: rc %d (%s), end capture: %s\n", rc, h->err, hipGetErrorString(ee));
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    e->seen = -1;                                     // this block keeps to plain launches
    return run();
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) {
    if (graph_mode > 1) fprintf(stderr, "[capi graph] instantiate failed: %s\n", hipGetErrorString(ei));
    (void)hipGetLastError(); e->seen = -1; return run();
  }
  e->exec = exec;
  if (graph_mode > 1) {
    size_t nodes = 0;
    fprintf(stderr, "[capi graph] order %lld captured\n", (long long)n);
    (void)nodes;
  }
  CAPI_HIP_CHECK(h, hipGraphLaunch(exec, h->stream));
  return CAPI_OK;
}

// doubles of scratch-2 the upper-case capi_dpotrf uses: room for the recursion's temporaries, the diagonal block's inverse, and
// the blocked routine's row panel + pair products
static size_t potrf_upper_scratch(int64_t n) {
  const size_t nb0 = (size_t)(n < 1024 ? n : 1024);
  return nb0 * nb0 * 2 + (size_t)LEAF * nb0 + (nb0 / 2 + LEAF) * (nb0 / 2 + LEAF);
}

int capi_dpotrf(capi_handle_t h, int uplo, int64_t n, double* A, int64_t lda) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ok01(uplo), "uplo code");
  CAPI_REQUIRE(h, n >= 0 && n < (1LL << 31), "n");
  if (n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, A && lda >= n, "A/lda");
  // cholinv.hpp:9 -- the hot path only ever factors 'U'.  'L' is served by factoring the transpose.
  if (uplo == CAPI_LOWER) {
    void* w;
    const size_t inner = potrf_upper_scratch(n);      // what the upper-case call below takes from the front of this block
    RC(capi_ws2_get(h, sizeof(double) * (inner + (size_t)n * (size_t)n), &w));
    double* U = (double*)w + inner;
    dim3 grid((unsigned)cdiv(n, 256), (unsigned)(n < 65535 ? n : 65535));
    hipLaunchKernelGGL(transpose_tri_kernel, grid, dim3(256), 0, h->stream, A, lda, U, n, n, 0);
    CAPI_HIP_CHECK(h, hipGetLastError());
    RC(capi_dpotrf(h, CAPI_UPPER, n, U, n));
    hipLaunchKernelGGL(transpose_tri_kernel, grid, dim3(256), 0, h->stream, U, n, A, lda, n, 1);
    CAPI_HIP_CHECK(h, hipGetLastError());
    return CAPI_OK;
  }
  // blocked right-looking with diagonal blocks of order NB factored-and-inverted by the recursion above
  const int64_t NB = 1024;
  if (n <= LEAF) return leaf_launch(h, A, lda, A, lda, (int)n, 0, 0, 0, 0, 0);
  const int64_t nb0 = n < NB ? n : NB;
  double *Xd = nullptr, *Wb = nullptr;
  // the diagonal blocks by the blocked routine (3 dependent launches per 128 columns) instead of the halving recursion (5 + a copy):
  // order 1024 0.51 against 0.7 ms.  (CAPI_POTRF_DIAG=rec: the recursion, A/B)
  static const bool diag_rec = getenv("CAPI_POTRF_DIAG") && getenv("CAPI_POTRF_DIAG")[0] == 'r';
  {
    // diag inverse lives in its own allocation slice: first nb0*nb0 doubles of scratch-2 are used by the
    // recursion's temporaries only up to (nb0/2)^2, so place Xd after them; the blocked routine's scratch follows
    void* w2;
    RC(capi_ws2_get(h, sizeof(double) * potrf_upper_scratch(n), &w2));
    Xd = (double*)w2 + (size_t)nb0 * nb0;
    Wb = Xd + (size_t)nb0 * nb0;
  }
  for (int64_t j0 = 0; j0 < n; j0 += NB) {
    const int64_t jb = (n - j0) < NB ? (n - j0) : NB, rest = n - j0 - jb;
    double* Ajj = A + j0 + j0 * lda;
    if (diag_rec) RC(potrf_trtri_rec(h, jb, Ajj, lda, Xd, nb0, 0, (int)j0));
    else RC(potrf_trtri_blocked(h, jb, Ajj, lda, Xd, nb0, (int)j0, Wb));
    if (rest > 0) {
      double* A12 = A + j0 + (j0 + jb) * lda;
      RC(capi_dtrmm(h, CAPI_LEFT, CAPI_UPPER, CAPI_TRANS, CAPI_NONUNIT, jb, rest, 1.0, Xd, nb0, A12, lda));
      RC(capi_dgemmt(h, CAPI_UPPER, CAPI_TRANS, CAPI_NOTRANS, rest, jb, -1.0, A12, lda, A12, lda, 1.0,
                     A + (j0 + jb) + (j0 + jb) * lda, lda));
    }
  }
  return CAPI_OK;
}

int capi_dtrtri(capi_handle_t h, int uplo, int diag, int64_t n, double* A, int64_t lda) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ok01(uplo) && ok01(diag), "enum code");
  CAPI_REQUIRE(h, n >= 0 && n < (1LL << 31), "n");
  if (n == 0) return CAPI_OK;
  CAPI_REQUIRE(h, A && lda >= n, "A/lda");
  if (uplo == CAPI_UPPER) return trtri_upper_rec(h, diag, n, A, lda);
  void* w;
  RC(capi_ws2_get(h, sizeof(double) * (size_t)n * (size_t)n * 2, &w));
  double* U = (double*)w + (size_t)n * n;
  dim3 grid((unsigned)cdiv(n, 256), (unsigned)(n < 65535 ? n : 65535));
  hipLaunchKernelGGL(transpose_tri_kernel, grid, dim3(256), 0, h->stream, A, lda, U, n, n, 0);
  CAPI_HIP_CHECK(h, hipGetLastError());
  RC(trtri_upper_rec(h, diag, n, U, n));
  hipLaunchKernelGGL(transpose_tri_kernel, grid, dim3(256), 0, h->stream, U, n, A, lda, n, 1);
  CAPI_HIP_CHECK(h, hipGetLastError());
  return CAPI_OK;
}

int capi_dtrsm(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb) {
  CAPI_REQUIRE(h, h, "null handle");
  CAPI_REQUIRE(h, ok01(side) && ok01(uplo) && ok01(trans) && ok01(diag), "enum code");
  CAPI_REQUIRE(h, m >= 0 && n >= 0 && m < (1LL << 31) && n < (1LL << 31), "dims");
  if (m == 0 || n == 0) return CAPI_OK;
  const int64_t nt = side == CAPI_LEFT ? m : n;
  CAPI_REQUIRE(h, T && B && ldt >= nt && ldb >= m, "operands");
  if (alpha != 1.0) {
    dim3 grid((unsigned)cdiv(m, 256), (unsigned)(n < 65535 ? n : 65535));
    hipLaunchKernelGGL(scale2d_kernel, grid, dim3(256), 0, h->stream, B, ldb, m, n, alpha);
    CAPI_HIP_CHECK(h, hipGetLastError());
    if (alpha == 0.0) return CAPI_OK;
  }
  return trsm_rec(h, side, uplo, trans, diag, m, n, T, ldt, B, ldb);
}

}  // extern "C"
