// capi_cpu_shim.cpp -- TEST INFRASTRUCTURE ONLY (lives under tests/, never shipped, never loaded by capital_amd).
//
// A host-memory implementation of the part of include/capital_hip.h that the host-side C++ layer
// (capital_amd/src, capital_amd/drivers) calls, built on the CPU oracle's kernels.  Linking the driver against this
// instead of libcapital_hip.so lets the multi-rank host logic (topo::square rank maps and communicator splits,
// SUMMA schedules incl. K-class stepping and K-slicing, the base-case gather, partner exchange, 1-D CQR2 allreduce)
// run on CPUs over torch.distributed/gloo with world_size > 1 (tests/test_multirank_gloo.py).  Collectives are
// forwarded to a callback that Python registers; a communicator is just the ordered list of world ranks it contains.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "capital_hip.h"
#include "../../oracle/capital_oracle.h"
#include "../../capital_amd/csrc/pair_paths.h"

struct capi_handle_s { char err[256]; int info; };
struct capi_comm_s { std::vector<int> ranks; int me; capi_handle_t h; };

// op: 0 bcast(buf,count,root idx) 1 allreduce(buf,count) 2 reduce(send->recv,root) 3 allgather(send->recv,count each) 4 sendrecv_replace(buf,count,peer idx)
typedef int (*shim_coll_cb)(int op, const int* ranks, int nranks, int me, double* buf, double* buf2, int64_t count, int root);
static shim_coll_cb g_cb = nullptr;
extern "C" void capi_shim_set_collective(shim_coll_cb cb) { g_cb = cb; }

#define IDX(p, i, j, ld) ((p)[(i) + (int64_t)(j) * (ld)])

extern "C" {

int capi_version(void) { return -100; }  // negative: this is the CPU test shim
int capi_device_count(void) { return 0; }
int capi_create(capi_handle_t* h, int) { *h = new capi_handle_s(); (*h)->err[0] = 0; (*h)->info = 0; return 0; }
int capi_create_on_stream(capi_handle_t* h, int d, void*) { return capi_create(h, d); }
int capi_destroy(capi_handle_t h) { delete h; return 0; }
void* capi_get_stream(capi_handle_t) { return nullptr; }
const char* capi_last_error(capi_handle_t h) { return h ? h->err : ""; }
int capi_malloc(capi_handle_t, void** p, size_t bytes) { *p = aligned_alloc(64, (bytes + 63) / 64 * 64 + 64); return *p ? 0 : CAPI_ENOMEM; }
int capi_free(capi_handle_t, void* p) { free(p); return 0; }
int capi_memset_async(capi_handle_t, void* p, int v, size_t bytes) { if (bytes) memset(p, v, bytes); return 0; }
int capi_memcpy_h2d(capi_handle_t, void* d, const void* s, size_t b) { if (b) memcpy(d, s, b); return 0; }
int capi_memcpy_d2h(capi_handle_t, void* d, const void* s, size_t b) { if (b) memcpy(d, s, b); return 0; }
int capi_memcpy_d2d_async(capi_handle_t, void* d, const void* s, size_t b) { if (b) memmove(d, s, b); return 0; }
int capi_sync(capi_handle_t) { return 0; }
int capi_reserve_workspace(capi_handle_t, size_t) { return 0; }
int capi_get_info(capi_handle_t h, int* info) { *info = h->info; return 0; }
int capi_reset_info(capi_handle_t h) { h->info = 0; return 0; }

int capi_dgemm(capi_handle_t, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
               const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  orc_dgemm(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
  return 0;
}
int capi_dgemmt(capi_handle_t, int uplo, int ta, int tb, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  std::vector<double> W((size_t)n * n, 0.0);
  orc_dgemm(ta, tb, n, n, k, alpha, A, lda, B, ldb, 0.0, W.data(), n);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i)
      if (uplo == CAPI_UPPER ? i <= j : i >= j) IDX(C, i, j, ldc) = (beta == 0.0 ? 0.0 : beta * IDX(C, i, j, ldc)) + W[i + j * n];
  return 0;
}
int capi_dsyrk(capi_handle_t, int uplo, int trans, int64_t n, int64_t k, double alpha, const double* A, int64_t lda, double beta,
               double* C, int64_t ldc) {
  orc_dsyrk(uplo, trans, n, k, alpha, A, lda, beta, C, ldc);
  return 0;
}
int capi_dtrmm(capi_handle_t, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha, const double* T,
               int64_t ldt, double* B, int64_t ldb) {
  orc_dtrmm(side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb);
  return 0;
}
int capi_dtrmm_acc(capi_handle_t, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha, const double* T,
                   int64_t ldt, const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
  std::vector<double> W((size_t)m * n);
  for (int64_t j = 0; j < n; ++j) memcpy(&W[j * m], &IDX(B, 0, j, ldb), sizeof(double) * m);
  orc_dtrmm(side, uplo, trans, diag, m, n, alpha, T, ldt, W.data(), m);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) IDX(C, i, j, ldc) = (beta == 0.0 ? 0.0 : beta * IDX(C, i, j, ldc)) + W[i + j * m];
  return 0;
}
int capi_dtrmm_oop(capi_handle_t h, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha, const double* T,
                   int64_t ldt, const double* B, int64_t ldb, double* C, int64_t ldc) {
  return capi_dtrmm_acc(h, side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb, 0.0, C, ldc);
}
// panel32 images (include/capital_hip.h): un-tile / re-tile around the column-major routines
static void from_panel32(const double* src, int64_t m, int64_t n, double* dst) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) dst[i + j * m] = src[(i / 32) * 32 * n + 32 * j + i % 32];
}
static void to_panel32(const double* src, int64_t m, int64_t n, double* dst) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) dst[(i / 32) * 32 * n + 32 * j + i % 32] = src[i + j * m];
}
int capi_dsyrk_panel32(capi_handle_t h, int64_t n, int64_t k, double alpha, const double* A32, double beta, double* C, int64_t ldc) {
  if (n != 256 || k % 32 || k < 64 * n) return CAPI_EINVAL;
  std::vector<double> W((size_t)k * n);
  from_panel32(A32, k, n, W.data());
  return capi_dsyrk(h, CAPI_UPPER, CAPI_TRANS, n, k, alpha, W.data(), k, beta, C, ldc);
}
int capi_dtrmm_right_panel32(capi_handle_t h, int64_t m, int64_t n, double alpha, const double* T, int64_t ldt, const double* B, int64_t ldb,
                             double* C, int64_t ldc) {
  if (n != 256 || m % 32 || m < 64 * n) return CAPI_EINVAL;
  std::vector<double> Bc, Cc;
  if (ldb == 0) { Bc.resize((size_t)m * n); from_panel32(B, m, n, Bc.data()); B = Bc.data(); ldb = m; }
  if (ldc != 0) return capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, m, n, alpha, T, ldt, B, ldb, C, ldc);
  Cc.resize((size_t)m * n);
  int rc = capi_dtrmm_oop(h, CAPI_RIGHT, CAPI_UPPER, CAPI_NOTRANS, CAPI_NONUNIT, m, n, alpha, T, ldt, B, ldb, Cc.data(), m);
  to_panel32(Cc.data(), m, n, C);
  return rc;
}
int capi_dtrsm(capi_handle_t, int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha, const double* T,
               int64_t ldt, double* B, int64_t ldb) {
  orc_dtrsm(side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb);
  return 0;
}
int capi_dpotrf(capi_handle_t h, int uplo, int64_t n, double* A, int64_t lda) {
  int info = orc_dpotrf(uplo, n, A, lda);
  if (info && !h->info) h->info = info;
  return 0;
}
int capi_dtrtri(capi_handle_t, int uplo, int diag, int64_t n, double* A, int64_t lda) { orc_dtrtri(uplo, diag, n, A, lda); return 0; }
int capi_dpotrf_trtri(capi_handle_t h, int64_t n, double* A, int64_t lda, double* X, int64_t ldx) {
  int info = orc_dpotrf(ORC_UPPER, n, A, lda);
  if (info && !h->info) h->info = info;
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i) {
      if (i > j) { IDX(A, i, j, lda) = 0.0; IDX(X, i, j, ldx) = 0.0; } else IDX(X, i, j, ldx) = IDX(A, i, j, lda);
    }
  orc_dtrtri(ORC_UPPER, ORC_NONUNIT, n, X, ldx);
  return 0;
}

int capi_serialize_shape(capi_handle_t, int shape, int ss, int ds, const double* src, int64_t sdimX, int64_t sdimY, double* dst,
                         int64_t ddimX, int64_t ddimY, int64_t ssx, int64_t sex, int64_t ssy, int64_t sey, int64_t dsx, int64_t dex,
                         int64_t dsy, int64_t dey) {
  (void)dex; (void)dey;
  const int64_t rangeX = sex - ssx, rangeY = sey - ssy;
  for (int64_t i = 0; i < rangeX; ++i) {
    int64_t so, d_o, cnt;
    if (shape == CAPI_LOWERTRI) { so = orc_offset(ss, ssx + i, ssy + i, sdimX, sdimY); d_o = orc_offset(ds, dsx + i, dsy + i, ddimX, ddimY); cnt = rangeY - i; }
    else { so = orc_offset(ss, ssx + i, ssy, sdimX, sdimY); d_o = orc_offset(ds, dsx + i, dsy, ddimX, ddimY); cnt = shape == CAPI_UPPERTRI ? i + 1 : rangeY; }
    memmove(dst + d_o, src + so, sizeof(double) * cnt);
  }
  return 0;
}
int capi_serialize(capi_handle_t h, int ss, int ds, const double* src, int64_t sdimX, int64_t sdimY, double* dst, int64_t ddimX,
                   int64_t ddimY, int64_t ssx, int64_t sex, int64_t ssy, int64_t sey, int64_t dsx, int64_t dex, int64_t dsy, int64_t dey) {
  const int shape = (ss == 2 || ds == 2) ? 2 : ((ss == 1 || ds == 1) ? 1 : 0);
  return capi_serialize_shape(h, shape, ss, ds, src, sdimX, sdimY, dst, ddimX, ddimY, ssx, sex, ssy, sey, dsx, dex, dsy, dey);
}
int capi_dlacpy(capi_handle_t, int part, int64_t m, int64_t n, const double* A, int64_t lda, double* B, int64_t ldb) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i)
      if (part == 0 || (part == 1 && i <= j) || (part == 2 && i >= j)) IDX(B, i, j, ldb) = IDX(A, i, j, lda);
  return 0;
}
int capi_dgeadd(capi_handle_t, int part, int64_t m, int64_t n, double alpha, const double* X, int64_t ldx, double beta, double* Y, int64_t ldy) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i)
      if (part == 0 || (part == 1 && i <= j) || (part == 2 && i >= j)) {
        const double x = alpha == 0.0 ? 0.0 : alpha * IDX(X, i, j, ldx);
        IDX(Y, i, j, ldy) = beta == 0.0 ? x : x + beta * IDX(Y, i, j, ldy);
      }
  return 0;
}
int capi_dtrizero(capi_handle_t, int keep, int64_t n, double* A, int64_t lda) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i)
      if (keep == CAPI_UPPER ? i > j : i < j) IDX(A, i, j, lda) = 0.0;
  return 0;
}
int capi_daxpby(capi_handle_t, int64_t count, double beta, const double* x, double* y) {
  for (int64_t i = 0; i < count; ++i) y[i] = beta * y[i] + x[i];
  return 0;
}
int capi_remove_triangle(capi_handle_t, char dir, double* A, int64_t dimX, int64_t dimY, int64_t px, int64_t py, int64_t P) {
  for (int64_t i = 0; i < dimX; ++i)
    for (int64_t j = 0; j < dimY; ++j) {
      const int64_t gx = px + i * P, gy = py + j * P;
      if (dir == 'U' ? gy > gx : gy < gx) A[i * dimY + j] = 0.0;
    }
  return 0;
}
int capi_block_to_cyclic(capi_handle_t, const double* blocked, double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  orc_block_to_cyclic_rect(blocked, cyclic, rl, cl, d);
  return 0;
}
int capi_cyclic_to_block(capi_handle_t, double* blocked, const double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  orc_cyclic_to_block_rect(blocked, cyclic, rl, cl, d);
  return 0;
}
int capi_block_to_cyclic_full(capi_handle_t, const double* blocked, double* cyclic, int64_t rl, int64_t cl, int64_t d) {
  const int64_t rg = rl * d;                                   // piece (x, y) = slice rank x + d y holds global (column i d + x, row k d + y)
  for (int64_t y = 0; y < d; ++y)
    for (int64_t x = 0; x < d; ++x)
      for (int64_t i = 0; i < cl; ++i)
        for (int64_t k = 0; k < rl; ++k) cyclic[(i * d + x) * rg + (k * d + y)] = blocked[(y * d + x) * rl * cl + i * rl + k];
  return 0;
}
int capi_block_to_cyclic_tri(capi_handle_t, const double* blocked, double* cyclic, int64_t rl, int64_t d) {
  orc_block_to_cyclic_triangle(blocked, cyclic, d * d * (rl * (rl + 1) / 2), rl, rl, d);
  return 0;
}
int capi_cyclic_to_block_tri(capi_handle_t, double* blocked, const double* cyclic, int64_t rl, int64_t d) {
  orc_cyclic_to_block_triangle(blocked, cyclic, d * d * (rl * (rl + 1) / 2), rl, rl, d);
  return 0;
}
int capi_cyclic_to_local(capi_handle_t, double* T, double* TI, int64_t L, int64_t bc, int64_t d, int64_t sr) {
  orc_cyclic_to_local(T, TI, L, bc, d, sr);
  return 0;
}
int capi_distribute_symmetric(capi_handle_t, double* data, int64_t dimX, int64_t dimY, int64_t gX, int64_t gY, int64_t px, int64_t py,
                              int64_t PX, int64_t PY, int64_t key, int dd) {
  orc_distribute_symmetric(data, dimX, dimY, gX, gY, px, py, PX, PY, key, dd);
  return 0;
}
int capi_distribute_random(capi_handle_t, double* data, int64_t dimX, int64_t dimY, int64_t gX, int64_t gY, int64_t px, int64_t py,
                           int64_t PX, int64_t PY, int64_t key) {
  orc_distribute_random(data, dimX, dimY, gX, gY, px, py, PX, PY, key);
  return 0;
}
int capi_distribute_identity(capi_handle_t, double* data, int64_t dimX, int64_t dimY, int64_t gX, int64_t gY, int64_t px, int64_t py,
                             int64_t PX, int64_t PY, double val) {
  orc_distribute_identity(data, dimX, dimY, gX, gY, px, py, PX, PY, val);
  return 0;
}
int capi_diff_norms(capi_handle_t, int part, int64_t m, int64_t n, const double* X, int64_t ldx, const double* Y, int64_t ldy, double* out) {
  out[0] = out[1] = 0.0;
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i)
      if (part == 0 || (part == 1 && i <= j) || (part == 2 && i >= j)) {
        const double y = IDX(Y, i, j, ldy), d = IDX(X, i, j, ldx) - y;
        out[0] += d * d;
        out[1] += y * y;
      }
  return 0;
}

// ---- communicators: ordered world-rank lists; collectives forwarded to the registered callback --------------------
int capi_comm_load_rccl(const char*) { return 0; }
int capi_comm_unique_id(void* id) {   // random per call, like the real one: the rendezvous test tells launches apart by it
  FILE* f = fopen("/dev/urandom", "rb");
  if (!f || fread(id, 1, 128, f) != 128) memset(id, 0x5a, 128);
  if (f) fclose(f);
  return 0;
}
int capi_comm_init_rank(capi_comm_t* out, capi_handle_t h, int nranks, const void*, int rank) {
  capi_comm_s* c = new capi_comm_s();
  c->h = h;
  for (int i = 0; i < nranks; ++i) c->ranks.push_back(i);
  c->me = rank;
  *out = c;
  return 0;
}
int capi_comm_rank(capi_comm_t c, int* r) { if (!c) return CAPI_EINVAL; *r = c->me; return 0; }
int capi_comm_size(capi_comm_t c, int* s) { if (!c) return CAPI_EINVAL; *s = (int)c->ranks.size(); return 0; }
int capi_comm_destroy(capi_comm_t c) { delete c; return 0; }
static int coll(capi_comm_t c, int op, double* buf, double* buf2, int64_t count, int root) {
  if (c->ranks.size() == 1) return 0;
  if (!g_cb) { snprintf(c->h->err, sizeof(c->h->err), "cpu shim: no collective callback registered"); return CAPI_ECOMM; }
  return g_cb(op, c->ranks.data(), (int)c->ranks.size(), c->me, buf, buf2, count, root) ? CAPI_ECOMM : 0;
}
int capi_comm_split(capi_comm_t parent, int color, int key, capi_comm_t* child) {
  if (!parent || !child) return CAPI_EINVAL;
  const int n = (int)parent->ranks.size();
  std::vector<double> mine = {(double)color, (double)key}, all((size_t)2 * n);
  if (n == 1) all = mine;
  else if (coll(parent, 3, mine.data(), all.data(), 2, 0)) return CAPI_ECOMM;
  *child = nullptr;
  if (color < 0) return 0;                                   // MPI_UNDEFINED: took part in the exchange, gets no communicator
  capi_comm_s* c = new capi_comm_s();
  c->h = parent->h;
  std::vector<std::pair<std::pair<int, int>, int>> members;  // ((key, parent index), world rank)
  for (int i = 0; i < n; ++i)
    if ((int)all[2 * i] == color) members.push_back({{(int)all[2 * i + 1], i}, parent->ranks[i]});
  std::sort(members.begin(), members.end());
  for (size_t i = 0; i < members.size(); ++i) {
    c->ranks.push_back(members[i].second);
    if (members[i].first.second == parent->me) c->me = (int)i;
  }
  *child = c;
  return 0;
}
int capi_bcast(capi_comm_t c, double* buf, int64_t count, int root) { return count ? coll(c, 0, buf, nullptr, count, root) : 0; }
int capi_allreduce_sum(capi_comm_t c, double* buf, int64_t count) { return count ? coll(c, 1, buf, nullptr, count, 0) : 0; }
int capi_reduce_sum(capi_comm_t c, const double* send, double* recv, int64_t count, int root) {
  if (c->ranks.size() == 1) { if (send != recv) memmove(recv, send, sizeof(double) * count); return 0; }
  return coll(c, 2, (double*)send, recv, count, root);
}
int capi_allgather(capi_comm_t c, const double* send, double* recv, int64_t count) {
  if (c->ranks.size() == 1) { if (send != recv) memmove(recv, send, sizeof(double) * count); return 0; }
  return coll(c, 3, (double*)send, recv, count, 0);
}
// gather / scatter ride on the callback's allgather and bcast (the shim has no point-to-point op of its own)
int capi_gather(capi_comm_t c, const double* send, double* recv, int64_t count, int root) {
  const size_t n = c->ranks.size();
  if (n == 1) { if (send != recv) memmove(recv, send, sizeof(double) * count); return 0; }
  std::vector<double> all(n * (size_t)count);
  if (coll(c, 3, (double*)send, all.data(), count, 0)) return CAPI_ECOMM;
  if (c->me == root) memcpy(recv, all.data(), sizeof(double) * all.size());
  return 0;
}
int capi_scatter(capi_comm_t c, const double* send, double* recv, int64_t count, int root) {
  const size_t n = c->ranks.size();
  if (n == 1) { if (send != recv) memmove(recv, send, sizeof(double) * count); return 0; }
  std::vector<double> all(n * (size_t)count);
  if (c->me == root) memcpy(all.data(), send, sizeof(double) * all.size());
  if (coll(c, 0, all.data(), nullptr, (int64_t)all.size(), root)) return CAPI_ECOMM;
  memcpy(recv, all.data() + (size_t)c->me * count, sizeof(double) * count);
  return 0;
}
// multi-path pair transfers: the product's own algorithm (capital_amd/csrc/pair_paths.h) over the callback's grouped point-to-point op
//   op 5: buf = array of p2p_entry, count = entries; every entry is posted (isend / irecv) before any is awaited -- one RCCL group
struct p2p_entry { double* ptr; int64_t count; int32_t peer_world; int32_t is_send; };
int64_t capi_pairs_scratch_count(int nranks, int64_t count) { return pair_paths::scratch_count(nranks, count); }
int capi_pairs_transfer(capi_comm_t c, const int* dst, const double* send, double* recv, int64_t count, double* scratch) {
  if (!c || !dst || count < 0) return CAPI_EINVAL;
  struct ShimPaths {
    capi_comm_t c;
    std::vector<p2p_entry> posted;
    int group_begin() { posted.clear(); return 0; }
    void send(const double* p, int64_t n, int peer) { posted.push_back({(double*)p, n, c->ranks[(size_t)peer], 1}); }
    void recv(double* p, int64_t n, int peer) { posted.push_back({p, n, c->ranks[(size_t)peer], 0}); }
    int group_end() {
      if (posted.empty()) return 0;
      if (!g_cb) return CAPI_ECOMM;
      return g_cb(5, c->ranks.data(), (int)c->ranks.size(), c->me, (double*)posted.data(), nullptr, (int64_t)posted.size(), 0) ? CAPI_ECOMM : 0;
    }
  };
  if (c->ranks.size() == 1) return dst[0] < 0 ? 0 : CAPI_EINVAL;
  const int64_t min_count = getenv("CAPITAL_MULTIPATH_MIN") ? atoll(getenv("CAPITAL_MULTIPATH_MIN")) : ((int64_t)1 << 20);
  ShimPaths x{c, {}};
  const int rc = pair_paths::transfer(x, c->me, (int)c->ranks.size(), dst, send, recv, count, scratch, min_count);
  return rc < 0 ? CAPI_EINVAL : rc;
}
int capi_comm_query(capi_comm_t c, int* r, int* s) { if (!c) return CAPI_EINVAL; *r = c->me; *s = (int)c->ranks.size(); return 0; }
int capi_trim_workspaces(capi_handle_t) { return 0; }
int capi_set_launch_rounds(capi_handle_t, int, int* was) { if (was) *was = 0; return 0; }
int capi_range_push(const char*) { return 0; }
int capi_range_pop(void) { return 0; }
int capi_sendrecv_replace(capi_comm_t c, double* buf, int64_t count, int peer, double* staging) {
  (void)staging;
  if (peer == c->me || count == 0) return 0;
  return coll(c, 4, buf, nullptr, count, peer);
}

// one host thread executes everything in program order: streams and events are no-ops
int capi_stream_select(capi_handle_t, int) { return 0; }
int capi_event_record(capi_handle_t, int) { return 0; }
int capi_event_wait(capi_handle_t, int) { return 0; }
int capi_prof_enable(capi_handle_t, int) { return 0; }
int capi_prof_collect(capi_handle_t, int, int64_t* l, double* ms, double* fl, double* mx) { *l = 0; *ms = 0; *fl = 0; if (mx) *mx = 0; return 0; }
int capi_mfma_f64_peak(capi_handle_t, int, double* t) { *t = 0; return 0; }
int capi_timer_start(capi_handle_t) { return 0; }
int capi_timer_stop_ms(capi_handle_t, float* ms) { *ms = 0; return 0; }

}  // extern "C"
