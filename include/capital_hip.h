/*
 * capital_hip.h -- C-ABI of libcapital_hip.so: the MI355X (gfx950) replacement for
 * the BLAS/LAPACK/MPI layer underneath huttered40/capital's recursive Cholesky
 * (cholinv) and CA-CholeskyQR2 (cacqr).
 *
 * This header IS the drop-in boundary.  Each entry point names the reference
 * interface it replaces (paths relative to the reference root).  All pointers
 * called A/B/C/T/data are DEVICE pointers (HBM); matrices are column-major fp64
 * exactly as the reference passes them to MKL.  Nothing here takes or returns a
 * C++ or torch type.  Every call is asynchronous on the handle's HIP stream
 * unless stated otherwise and returns a capi_status (0 = ok).
 *
 * Enum codes are the reference's own (src/blas/engine.h:23-46, src/lapack/engine.h:23-36):
 *   Transpose NoTrans=0 Trans=1 | Side Left=0 Right=1 | UpLo Lower=0 Upper=1 | Diag NonUnit=0 Unit=1
 */
#ifndef CAPITAL_HIP_H_
#define CAPITAL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct capi_handle_s* capi_handle_t;
typedef struct capi_comm_s*   capi_comm_t;

enum capi_status {
  CAPI_OK = 0,
  CAPI_EINVAL = -1,   /* bad argument (dims, enum code, null pointer, alignment) */
  CAPI_EHIP = -2,     /* a HIP runtime call failed; see capi_last_error() */
  CAPI_ENOMEM = -3,
  CAPI_ECOMM = -4,    /* RCCL failure or communicator misuse */
  CAPI_ENOTSPD = -5   /* potrf met a non-positive pivot (see capi_get_info) */
};
enum { CAPI_NOTRANS = 0, CAPI_TRANS = 1 };
enum { CAPI_LEFT = 0, CAPI_RIGHT = 1 };
enum { CAPI_LOWER = 0, CAPI_UPPER = 1 };
enum { CAPI_NONUNIT = 0, CAPI_UNIT = 1 };
/* matrix structure policies, src/matrix/structure.h:8-72 */
enum { CAPI_RECT = 0, CAPI_UPPERTRI = 1, CAPI_LOWERTRI = 2 };

/* ---- lifecycle / memory (replaces new[]/memcpy inside matrix<>, src/matrix/structure.hpp:4-26) ---- */
int  capi_version(void);
int  capi_device_count(void);
int  capi_create(capi_handle_t* h, int device);                 /* owns a new non-blocking stream */
int  capi_create_on_stream(capi_handle_t* h, int device, void* hip_stream); /* borrows the caller's stream */
int  capi_destroy(capi_handle_t h);
void* capi_get_stream(capi_handle_t h);
const char* capi_last_error(capi_handle_t h);
int  capi_malloc(capi_handle_t h, void** dptr, size_t bytes);
int  capi_free(capi_handle_t h, void* dptr);
int  capi_memset_async(capi_handle_t h, void* dptr, int value, size_t bytes);
int  capi_memcpy_h2d(capi_handle_t h, void* dst, const void* src, size_t bytes);   /* synchronous */
int  capi_memcpy_d2h(capi_handle_t h, void* dst, const void* src, size_t bytes);   /* synchronous */
int  capi_memcpy_d2d_async(capi_handle_t h, void* dst, const void* src, size_t bytes);
int  capi_sync(capi_handle_t h);
/* grow the handle's private workspace (used by in-place trmm, split-K slabs, potrf) ahead of time */
int  capi_reserve_workspace(capi_handle_t h, size_t bytes);
/* release every private workspace block of the handle (synchronises); they are re-created on demand */
int  capi_trim_workspaces(capi_handle_t h);
/* on = 1: every large launch of the handle goes out one resident round (2 x CUs workgroups: an 8 x 8 block of tiles per XCD) at a time --
 * plain and triangular outputs, TRMMs as equal-work tile pairs: same time, about half the L2-to-fabric traffic, for callers whose products
 * all run on one stream (grids, the TRSM mode); on = 0: the handle's defaults (environment: CAPI_ROUNDS, CAPI_TRMM_PAIR, CAPI_TRMM_PAIR_ROUNDS).
 * *was (may be NULL) receives the previous setting. */
int  capi_set_launch_rounds(capi_handle_t h, int on, int* was);
/* keep the handle's compute stream off `reserve` CUs (a multiple of 32: one per shader engine and XCD; 0 = all CUs again): the communication stream's kernels (RCCL's
 * send/recv on a grid) then start at once beside a tile launch instead of at its next round boundary.  Drains the handle's streams. */
int  capi_reserve_cus(capi_handle_t h, int reserve);

/* ---- BLAS layer: replaces blas::engine::_gemm/_trmm/_syrk (src/blas/interface.h:58-66,
 *      src/blas/interface.hpp:43-97 -> cblas_dgemm/dtrmm/dsyrk) ---- */
/* C <- alpha*op(A)*op(B) + beta*C  (K1-K3; call sites summa.hpp:28,139,144; cacqr.hpp:95,144) */
int capi_dgemm(capi_handle_t h, int transA, int transB, int64_t m, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
/* C(uplo) <- alpha*A^T*A + beta*C (trans=Trans, A is k x n) or alpha*A*A^T (NoTrans, A is n x k)
 * (K7; call site cacqr.hpp:15).  Only the `uplo` triangle of C is read or written. */
int capi_dsyrk(capi_handle_t h, int uplo, int trans, int64_t n, int64_t k, double alpha,
               const double* A, int64_t lda, double beta, double* C, int64_t ldc);
/* Triangular-output GEMM: C(uplo) <- alpha*op(A)*op(B) + beta*C, C is n x n.  This is what the
 * reference's summa "syrk" (summa.hpp:111-158) needs: it runs a full cblas_dgemm there because a
 * rank's two operands are different blocks; computing only the needed triangle halves the work. */
int capi_dgemmt(capi_handle_t h, int uplo, int transA, int transB, int64_t n, int64_t k, double alpha,
                const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
/* B <- alpha*op(T)*B (Left) or alpha*B*op(T) (Right); T triangular, only its `uplo` triangle is read
 * (K4-K6; call sites summa.hpp:64,71; cacqr.hpp:25,185).  In place, as cblas_dtrmm. */
int capi_dtrmm(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb);
/* Out-of-place form used on the hot path: C <- alpha*op(T)*B or alpha*B*op(T); C must not alias B. */
int capi_dtrmm_oop(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
                   const double* T, int64_t ldt, const double* B, int64_t ldb, double* C, int64_t ldc);
/* accumulating form for multi-step SUMMA (d/c > 1 K-panels per layer): C <- alpha*op(T)*B + beta*C */
int capi_dtrmm_acc(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
                   const double* T, int64_t ldt, const double* B, int64_t ldb, double beta, double* C, int64_t ldc);
/* B <- alpha*op(T)^-1*B or alpha*B*op(T)^-1.  Not in the reference (it forms trtri+trmm instead,
 * SURVEY.md quick facts); named by BASELINE.json north_star. */
int capi_dtrsm(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
               const double* T, int64_t ldt, double* B, int64_t ldb);

/* ---- LAPACK layer: replaces lapack::engine::_potrf/_trtri (src/lapack/interface.h:49-53,
 *      src/lapack/interface.hpp:30-58 -> LAPACKE_dpotrf/dtrtri, column-major) ---- */
/* "panel32" images of a tall m x 256 panel (m % 32 == 0): m / 32 tiles of 32 rows, each tile column-major with ld 32, tile t at
 * 8192 t doubles -- element (i, j) at (i / 32) * 8192 + 32 j + i % 32.  One contiguous stream per pass instead of 256 column streams.
 * qr::cacqr (cacqr.hpp:174-193) keeps CholeskyQR2's intermediate Q1 = A R1^-1 in this form between its two sweeps (K7 / K5 of SURVEY 2.2 on
 * that image): capi_dsyrk_panel32: C(upper) = alpha A^T A + beta C, A a panel32 image of k x 256 (k >= 64 * 256);
 * capi_dtrmm_right_panel32: C = alpha B T, T 256 x 256 upper, non-unit; ldb == 0 / ldc == 0 mark B / C as panel32 images. */
int capi_dsyrk_panel32(capi_handle_t h, int64_t n, int64_t k, double alpha, const double* A32, double beta, double* C, int64_t ldc);
int capi_dtrmm_right_panel32(capi_handle_t h, int64_t m, int64_t n, double alpha, const double* T, int64_t ldt,
                             const double* B, int64_t ldb, double* C, int64_t ldc);
/* The LAPACK `info` the reference discards is kept on the device; capi_get_info() synchronises and
 * returns it (0, or 1-based index of the first non-positive pivot / zero diagonal). */
int capi_dpotrf(capi_handle_t h, int uplo, int64_t n, double* A, int64_t lda);
int capi_dtrtri(capi_handle_t h, int uplo, int diag, int64_t n, double* A, int64_t lda);
/* Fused base case of cholinv (cholinv/policy.h:190-205: potrf, memcpy, trtri): A(upper) -> R in place,
 * Rinv <- R^-1 (upper); strictly-lower parts of both outputs are zeroed (cyclic_to_local, util.hpp:131-164). */
int capi_dpotrf_trtri(capi_handle_t h, int64_t n, double* A, int64_t lda, double* Rinv, int64_t ldi);
int capi_get_info(capi_handle_t h, int* info);
/* LAPACKE_dgeqrf / LAPACKE_dorgqr behind lapack::engine::_geqrf / _orgqr (lapack/interface.hpp:60-88; the reference has
 * the slots but no caller -- CholeskyQR2 is its QR).  Householder QR, LAPACK storage: R in the upper triangle, the
 * reflectors v_j (unit first entry implied) below it, tau[min(m,n)] on the DEVICE.  capi_dorgqr overwrites A (m x n,
 * m >= n >= k) with the first n columns of Q = H_1 ... H_k.  Blocked (compact WY, width 32) on the MFMA tile kernel. */
int capi_dgeqrf(capi_handle_t h, int64_t m, int64_t n, double* A, int64_t lda, double* tau);
int capi_dorgqr(capi_handle_t h, int64_t m, int64_t n, int64_t k, double* A, int64_t lda, const double* tau);
int capi_reset_info(capi_handle_t h);

/* ---- data movement: replaces serialize<> (src/matrix/serialize.hpp:12-150), the pack/unpack and
 *      axpy loops of summa (summa.hpp:33,135,147-153,216-217) and util::remove_triangle (util.hpp:266-318) ---- */
/* sub-block copy between two (possibly packed-triangular) local layouts; ranges as in serialize<>::invoke */
int capi_serialize(capi_handle_t h, int src_struct, int dst_struct,
                   const double* src, int64_t sdimX, int64_t sdimY, double* dst, int64_t ddimX, int64_t ddimY,
                   int64_t ssx, int64_t sex, int64_t ssy, int64_t sey, int64_t dsx, int64_t dex, int64_t dsy, int64_t dey);
/* same, with the copy SHAPE (rect: whole columns, uppertri: i+1 leading entries of column i, lowertri: from the
 * diagonal down) given separately from the two storage layouts -- the reference applies serialize<uppertri,uppertri>
 * to rect-stored matrices too (cholinv.hpp:13) */
int capi_serialize_shape(capi_handle_t h, int shape, int src_struct, int dst_struct,
                         const double* src, int64_t sdimX, int64_t sdimY, double* dst, int64_t ddimX, int64_t ddimY,
                         int64_t ssx, int64_t sex, int64_t ssy, int64_t sey, int64_t dsx, int64_t dex, int64_t dsy, int64_t dey);
/* B <- A for an m x n block; part: 0 all, 1 upper triangle incl. diagonal, 2 lower incl. diagonal */
int capi_dlacpy(capi_handle_t h, int part, int64_t m, int64_t n, const double* A, int64_t lda, double* B, int64_t ldb);
/* Y(part) <- alpha*X + beta*Y on an m x n block (the beta-update after summa's Allreduce, summa.hpp:33,153) */
int capi_dgeadd(capi_handle_t h, int part, int64_t m, int64_t n, double alpha, const double* X, int64_t ldx, double beta, double* Y, int64_t ldy);
/* zero the strictly lower (uplo=Upper keeps upper) or strictly upper part of an n x n block */
int capi_dtrizero(capi_handle_t h, int keep_uplo, int64_t n, double* A, int64_t lda);
/* y <- beta*y + x over count elements (M3) */
int capi_daxpby(capi_handle_t h, int64_t count, double beta, const double* x, double* y);
/* zero by GLOBAL index parity: local (i,j) is global (px+i*P, py+j*P); dir 'U' zeroes gy>gx, 'L' zeroes gy<gx */
int capi_remove_triangle(capi_handle_t h, char dir, double* A, int64_t dimX, int64_t dimY, int64_t px, int64_t py, int64_t P);
/* element-cyclic gather/scatter of the base case (util.hpp:56-230): pieces[r] (r = x*d + y... see DESIGN.md)
 * are d*d local blocks of rows_local x cols_local; cyclic is the (rows_local*d) x (cols_local*d) aggregate */
int capi_block_to_cyclic(capi_handle_t h, const double* blocked, double* cyclic, int64_t rows_local, int64_t cols_local, int64_t d);
int capi_cyclic_to_block(capi_handle_t h, double* blocked, const double* cyclic, int64_t rows_local, int64_t cols_local, int64_t d);
/* the same assembly for an OFF-diagonal aggregate (the R12 block a grid solves in TRSM mode): no part of it is zeroed */
int capi_block_to_cyclic_full(capi_handle_t h, const double* blocked, double* cyclic, int64_t rows_local, int64_t cols_local, int64_t d);
/* the same for PACKED upper-triangular pieces of rows_local (rows_local + 1) / 2 doubles each (util::block_to_cyclic_triangle /
 * cyclic_to_block_triangle, util.hpp:57-102,167-201: the Serialize policy's base-case messages, policy.h:176,322-377) */
int capi_block_to_cyclic_tri(capi_handle_t h, const double* blocked_packed, double* cyclic, int64_t rows_local, int64_t d);
int capi_cyclic_to_block_tri(capi_handle_t h, double* blocked_packed, const double* cyclic, int64_t rows_local, int64_t d);
/* util::cyclic_to_local (util.hpp:131-164): slice rank `slice_rank`'s element-cyclic piece of the aggregated factor T and of its
 * inverse TI (bc_dim x bc_dim each) into their leading local_dim x local_dim corners (ld stays bc_dim), zero below the global diagonal */
int capi_cyclic_to_local(capi_handle_t h, double* T, double* TI, int64_t local_dim, int64_t bc_dim, int64_t d, int64_t slice_rank);

/* ---- generators: replaces rect::_distribute_* (src/matrix/structure.hpp:36-129), bit-identical output ---- */
int capi_distribute_symmetric(capi_handle_t h, double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                              int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key, int diag_dominant);
int capi_distribute_random(capi_handle_t h, double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                           int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key);
int capi_distribute_identity(capi_handle_t h, double* data, int64_t dimX, int64_t dimY, int64_t gdimX, int64_t gdimY,
                             int64_t px, int64_t py, int64_t PX, int64_t PY, double val);

/* ---- residual reductions: replaces util::residual_local's local sums (src/util/util.hpp:25-53) ---- */
/* out[0] = sum (X[i]-Y[i])^2 over the selected part, out[1] = sum Y[i]^2; part as capi_dlacpy; out is HOST memory (synchronous) */
int capi_diff_norms(capi_handle_t h, int part, int64_t m, int64_t n, const double* X, int64_t ldx, const double* Y, int64_t ldy, double* out2);

/* ---- collectives: replaces the MPI calls of summa/util/policy (C1-C10) with RCCL on the
 *      handle's stream.  A communicator is created from a 128-byte RCCL unique id that rank 0 obtains with
 *      capi_comm_unique_id() and the launcher ships to the other ranks (torch.distributed store, MPI, a file). ---- */
int capi_comm_load_rccl(const char* librccl_path);              /* optional: choose the librccl.so to bind (default: search) */
int capi_comm_unique_id(void* id128);
int capi_comm_init_rank(capi_comm_t* comm, capi_handle_t h, int nranks, const void* id128, int rank);
int capi_comm_split(capi_comm_t parent, int color, int key, capi_comm_t* child);       /* MPI_Comm_split, topology.h:28-59,84-138; color < 0 = MPI_UNDEFINED: *child = NULL */
int capi_comm_query(capi_comm_t c, int* rank, int* size);                              /* rank and size as RCCL itself reports them */
int capi_comm_rank(capi_comm_t c, int* rank);
int capi_comm_size(capi_comm_t c, int* size);
int capi_comm_destroy(capi_comm_t c);
int capi_bcast(capi_comm_t c, double* buf, int64_t count, int root);                   /* MPI_Bcast, summa.hpp:185,193 */
int capi_allreduce_sum(capi_comm_t c, double* buf, int64_t count);                     /* MPI_Allreduce(IN_PLACE,SUM), summa.hpp:236 */
int capi_reduce_sum(capi_comm_t c, const double* send, double* recv, int64_t count, int root);  /* MPI_Reduce, cacqr.hpp:98 */
int capi_allgather(capi_comm_t c, const double* send, double* recv, int64_t count_per_rank);   /* MPI_Allgather, policy.h:176 */
int capi_sendrecv_replace(capi_comm_t c, double* buf, int64_t count, int peer, double* staging); /* MPI_Sendrecv_replace, util.hpp:240 */
/* Grid-wide set of pair transfers over ALL xGMI links of the node (csrc/pair_paths.h).  The pair collectives of the 3-D SUMMA on a
 * 2 x 2 x 2 grid -- MPI_Bcast over a row / column of two (summa.hpp:185,193), the halves of MPI_Allreduce over a depth fibre of two
 * (summa.hpp:236), MPI_Sendrecv_replace with the transpose partner (util.hpp:240) -- each move 1-2 GiB between two GPUs; as calls on
 * 2-rank communicators they use one of the sender's seven links.  Here every rank of `world` calls with the SAME dst[0..size): rank r
 * sends `count` doubles from `send` to rank dst[r] (dst[r] < 0: sends nothing); a rank is the destination of at most one transfer and
 * receives into `recv`.  Long messages are cut into `size` units: two travel directly, the others are relayed by the remaining ranks
 * (two grouped rounds of ncclSend/ncclRecv) through `scratch`, capi_pairs_scratch_count(size, count) doubles on every rank. */
int64_t capi_pairs_scratch_count(int nranks, int64_t count);
int capi_pairs_transfer(capi_comm_t world, const int* dst, const double* send, double* recv, int64_t count, double* scratch);
/* root collects / deals `count_per_rank` doubles per rank, rank r's piece at recv/send + r*count (one group of RCCL send/recv) */
int capi_gather(capi_comm_t c, const double* send, double* recv, int64_t count_per_rank, int root);   /* MPI_Gather, cholinv/policy.h:322-332 */
int capi_scatter(capi_comm_t c, const double* send, double* recv, int64_t count_per_rank, int root);  /* MPI_Scatter / MPI_Iscatter, policy.h:361-377,470-488 */

/* ---- extra streams + events: lets the host layer run collectives beside the tile kernel (the reference's
 *      MPI_Ibcast/Iallreduce chunk pipeline, summa.hpp:195-215,238-249) and the bulk of a trailing update beside the
 *      next block's factorisation.  capi_stream_select(h, i) routes every following call on this handle (kernels,
 *      copies, collectives) to stream i: 0 = compute stream, 1 = communication / packing stream, 2..3 = low-priority bulk
 *      streams.  Each stream index has its own private workspace; events order work across streams (slot in [0, 1024)). ---- */
int capi_stream_select(capi_handle_t h, int which);
int capi_event_record(capi_handle_t h, int slot);   /* on the currently selected stream */
int capi_event_wait(capi_handle_t h, int slot);     /* the currently selected stream waits for that record */

/* ---- measurement helpers ---- */
/* register-resident v_mfma_f64_16x16x4_f64 loop on every CU: returns achieved TFLOP/s (synchronous) */
int capi_mfma_f64_peak(capi_handle_t h, int iters, double* tflops);
/* HIP-event bracket around EVERY launch of the MFMA tile kernel on the handle's stream (roofline measurement):
 * enable, run the workload, collect = number of launches, summed duration and summed algorithmic flops of one
 * kernel variant (0..3 = operand orientations at any tile size, 3 = both operands k-contiguous, the TN kernel of the
 * trailing update; 8 + v = exactly variant v, where bit 2 of v marks the 64-tile kernel: 11 = the 128-tile TN kernel
 * alone, i.e. one kernel symbol; 8 + 16 + v = dtrmm_pair_kernel (the 128-tile kernel's form for TRMMs in tile pairs) of orientation v; 100 + o = both 128-tile symbols of operand
 * orientation o; -1 = all variants). */
int capi_prof_enable(capi_handle_t h, int on);
int capi_prof_collect(capi_handle_t h, int variant, int64_t* launches, double* total_ms, double* total_flops, double* max_ms);
/* the same records timed by the kernels themselves (first workgroup's start .. last workgroup's end, wall-clock ticks): *union_ms = length
 * of the union of the selected launches' execution intervals, *sum_ms their sum.  Robust to launches of several streams interleaving. */
int capi_prof_collect_intervals(capi_handle_t h, int variant, int64_t* launches, double* union_ms, double* sum_ms, double* total_flops, double* max_ms);
/* phase markers: the reference's CRITTER_START/STOP(sym) regions (src/util/shared.h:26-35; cholinv.hpp:94-158, cacqr.hpp:82-116)
 * as roctx ranges for `rocprofv3 --marker-trace`; no-ops when no roctx library can be bound */
int capi_range_push(const char* name);
int capi_range_pop(void);
/* HIP-event timer on the handle's stream */
int capi_timer_start(capi_handle_t h);
int capi_timer_stop_ms(capi_handle_t h, float* ms);              /* synchronises */

#ifdef __cplusplus
}
#endif
#endif /* CAPITAL_HIP_H_ */
