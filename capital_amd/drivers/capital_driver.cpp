// capital_driver.cpp -- C-ABI over the host-side C++ layer (capital_amd/src): what bench.py, the parity tests and a
// foreign-language caller bind.  It plays the part of the reference's bench mains (bench/cholesky/cholinv.cpp:8-71,
// bench/qr/cacqr.cpp:8-77) split into create / generate / factor / validate steps so that a launcher can time factor()
// alone.  Policy templates are instantiated for every combination the reference exposes.
#include <memory>

#include "../src/alg/cholesky/cholinv/cholinv.h"
#include "../src/alg/qr/cacqr/cacqr.h"
#include "../test/cholesky/validate.h"
#include "../test/qr/validate.h"

namespace {

CAPITAL_RANK_LOCAL std::string g_err;      // (per rank: thread-local in the ranks-as-threads rehearsal build)
template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

using T = double;
using U = int64_t;
using MatrixType = matrix<T, U, rect>;

struct cholinv_problem {
  virtual ~cholinv_problem() {}
  virtual void generate() = 0;
  virtual void set_A(const double* host) = 0;
  virtual void factor() = 0;
  virtual double residual() = 0;
  virtual void get(int which, double* host) = 0;
  virtual void dims(int64_t* nloc, int* x, int* y, int* z, int* d, int* c) = 0;
  virtual void stats(int64_t* bc, int64_t* levels, int64_t* bcdim) = 0;
  virtual void set_trsm_mode(bool on) = 0;
};

template <class Alg>
struct cholinv_impl : cholinv_problem {
  topo::square grid;
  MatrixType A;
  typename Alg::template info<T, U> pack;
  cholinv_impl(int64_t n, int c, int layout, int chunks, int complete_inv, int split, int bc_mult)
      : grid(capital::world(), (size_t)c, (size_t)layout, (size_t)chunks), A(n, n, grid.d, grid.d), pack(complete_inv, split, bc_mult, 'U') {}
  void generate() override { A.distribute_symmetric(grid.x, grid.y, grid.d, grid.d, grid.rank / grid.c, true); }   // bench/cholesky/cholinv.cpp:40
  void set_A(const double* host) override { A.from_host(host); }
  void factor() override { Alg::factor(A, pack, grid); }
  double residual() override { return cholesky::validate<Alg>::residual(A, pack, grid); }
  void get(int which, double* host) override {
    if (which == 0) { auto v = A.to_host(); std::memcpy(host, v.data(), sizeof(double) * v.size()); return; }
    auto M = which == 1 ? Alg::construct_R(pack, grid) : Alg::construct_Rinv(pack, grid);
    auto v = M.to_host();
    std::memcpy(host, v.data(), sizeof(double) * v.size());
  }
  void dims(int64_t* nloc, int* x, int* y, int* z, int* d, int* c) override {
    *nloc = A.num_rows_local(); *x = (int)grid.x; *y = (int)grid.y; *z = (int)grid.z; *d = (int)grid.d; *c = (int)grid.c;
  }
  void stats(int64_t* bc, int64_t* levels, int64_t* bcdim) override { *bc = pack.num_base_cases; *levels = pack.num_levels; *bcdim = pack.bcDimension; }
  void set_trsm_mode(bool on) override { pack.solve_with_trsm = on; }
};

template <class SP, class IP>
cholinv_problem* make_cholinv(int bc_policy, int64_t n, int c, int layout, int chunks, int ci, int split, int bcm) {
  namespace P = cholesky::policy::cholinv;
  switch (bc_policy) {
    case 0: return new cholinv_impl<cholesky::cholinv<SP, IP, P::ReplicateCommComp>>(n, c, layout, chunks, ci, split, bcm);
    case 1: return new cholinv_impl<cholesky::cholinv<SP, IP, P::ReplicateComp>>(n, c, layout, chunks, ci, split, bcm);
    case 2: return new cholinv_impl<cholesky::cholinv<SP, IP, P::NoReplication>>(n, c, layout, chunks, ci, split, bcm);
    case 3: return new cholinv_impl<cholesky::cholinv<SP, IP, P::NoReplicationOverlap>>(n, c, layout, chunks, ci, split, bcm);
  }
  throw std::invalid_argument("base-case policy id must be 0..3 (policy.h get_id)");
}

struct cacqr_problem {
  virtual ~cacqr_problem() {}
  virtual void generate() = 0;
  virtual void set_A(const double* host) = 0;
  virtual void factor() = 0;
  virtual double residual() = 0;
  virtual double orthogonality() = 0;
  virtual void get(int which, double* host) = 0;
  virtual void dims(int64_t* mloc, int64_t* n) = 0;
  virtual void get_rows(int which, int64_t row0, int64_t nrows, double* host) = 0;
};

template <class Alg>
struct cacqr_impl : cacqr_problem {
  using CI = cholesky::cholinv<cholesky::policy::cholinv::Serialize, cholesky::policy::cholinv::SaveIntermediates, cholesky::policy::cholinv::NoReplication>;
  topo::rect grid;
  MatrixType A;
  typename Alg::template info<T, U, CI> pack;
  cacqr_impl(int64_t m, int64_t n, int c, int variant, int layout, int chunks, int ci, int split, int bcm)
      : grid(capital::world(), (size_t)c, (size_t)layout, (size_t)chunks), A(n, m, grid.c, grid.d),
        pack((size_t)variant, typename CI::template info<T, U>(ci, split, bcm, 'U')) {}
  void generate() override { A.distribute_random(grid.x, grid.y, grid.c, grid.d, grid.rank / grid.c); }             // bench/qr/cacqr.cpp:34
  void set_A(const double* host) override { A.from_host(host); }
  void factor() override { Alg::factor(A, pack, grid); }
  double residual() override { return qr::validate<Alg>::residual(A, pack, grid); }
  double orthogonality() override { return qr::validate<Alg>::orthogonality(A, pack, grid); }
  void get(int which, double* host) override {
    if (which == 0) { auto v = A.to_host(); std::memcpy(host, v.data(), sizeof(double) * v.size()); return; }
    if (which == 1) { auto v = Alg::construct_Q(pack, grid).to_host(); std::memcpy(host, v.data(), sizeof(double) * v.size()); return; }
    if (which == 3) {                                     // Q^T Q (summed over the ranks), n x n
      matrix<double, int64_t, rect> I(A.num_columns_global(), A.num_columns_global(), 1, 1);
      qr::validate<Alg>::gram_of_Q(pack, grid, I);
      auto v = I.to_host();
      std::memcpy(host, v.data(), sizeof(double) * v.size());
      return;
    }
    auto v = Alg::construct_R(pack, grid).to_host();
    std::memcpy(host, v.data(), sizeof(double) * v.size());
  }
  void dims(int64_t* mloc, int64_t* n) override { *mloc = A.num_rows_local(); *n = A.num_columns_local(); }
  // local rows [row0, row0 + nrows) of A (which = 0) or Q (1), column-major nrows x n: a bounded window of a panel that is
  // too large to fetch whole (config 5: 64 GiB per GPU)
  void get_rows(int which, int64_t row0, int64_t nrows, double* host) override {
    const int64_t m = A.num_rows_local(), n = A.num_columns_local();
    if (row0 < 0 || nrows < 0 || row0 + nrows > m || (which != 0 && which != 1)) throw std::invalid_argument("cacqr get_rows: window outside the local panel");
    const double* src = (which == 0 ? A.data() : pack.Q.data()) + row0;
    double* tmp = capital::dev_alloc(nrows * n);
    CAPITAL_CHECK(capi_dlacpy(capital::handle(), 0, nrows, n, src, m, tmp, nrows));
    CAPITAL_CHECK(capi_memcpy_d2h(capital::handle(), host, tmp, sizeof(double) * (size_t)(nrows * n)));
    capital::dev_free(tmp);
  }
};

}  // namespace

extern "C" {

const char* capital_drv_last_error(void) { return g_err.c_str(); }

// stream != NULL: run on the caller's HIP stream (e.g. torch's current stream); uid: 128-byte RCCL id when size > 1
int capital_drv_init(int device, int rank, int size, const void* uid, void* stream) {
  return guarded([&] {
    if (stream) {
      capi_handle_t h;
      if (capi_create_on_stream(&h, device, stream) != CAPI_OK) throw std::runtime_error("capi_create_on_stream failed");
      capital::ctx().owns_handle = true;
      capital::ctx().device = device;
      capital::init_with_handle(h, rank, size, uid);
    } else {
      capital::init(device, rank, size, uid);
    }
  });
}
int capital_drv_finalize(void) { return guarded([] { capital::finalize(); }); }
int capital_drv_sync(void) { return guarded([] { capital::sync(); }); }
void* capital_drv_handle(void) { return capital::ctx().handle; }
// rank and size of the world communicator as RCCL itself reports them (a launcher prints them next to its own)
int capital_drv_world_query(int* rank, int* size) {
  return guarded([&] {
    if (!capital::world()) throw std::runtime_error("no world communicator: call capital_drv_init first");
    CAPITAL_CHECK(capi_comm_query(capital::world(), rank, size));
  });
}

// serialize_: 0 NoSerialize, 1 Serialize; + 2: FlushIntermediates instead of SaveIntermediates (policy.h:21-156)
void* capital_cholinv_create(int64_t n, int c, int layout, int num_chunks, int complete_inv, int split, int bc_mult, int serialize_, int bc_policy) {
  cholinv_problem* p = nullptr;
  int rc = guarded([&] {
    namespace P = cholesky::policy::cholinv;
    const bool ser = (serialize_ & 1) != 0, flush = (serialize_ & 2) != 0;
    if (ser && flush) p = make_cholinv<P::Serialize, P::FlushIntermediates>(bc_policy, n, c, layout, num_chunks, complete_inv, split, bc_mult);
    else if (ser) p = make_cholinv<P::Serialize, P::SaveIntermediates>(bc_policy, n, c, layout, num_chunks, complete_inv, split, bc_mult);
    else if (flush) p = make_cholinv<P::NoSerialize, P::FlushIntermediates>(bc_policy, n, c, layout, num_chunks, complete_inv, split, bc_mult);
    else p = make_cholinv<P::NoSerialize, P::SaveIntermediates>(bc_policy, n, c, layout, num_chunks, complete_inv, split, bc_mult);
  });
  return rc ? nullptr : p;
}
int capital_cholinv_generate(void* p) { return guarded([&] { ((cholinv_problem*)p)->generate(); }); }
int capital_cholinv_set_A(void* p, const double* host) { return guarded([&] { ((cholinv_problem*)p)->set_A(host); }); }
int capital_cholinv_factor(void* p) { return guarded([&] { ((cholinv_problem*)p)->factor(); }); }
int capital_cholinv_residual(void* p, double* out) { return guarded([&] { *out = ((cholinv_problem*)p)->residual(); }); }
int capital_cholinv_get(void* p, int which, double* host) { return guarded([&] { ((cholinv_problem*)p)->get(which, host); }); }
int capital_cholinv_dims(void* p, int64_t* nloc, int* x, int* y, int* z, int* d, int* c) { return guarded([&] { ((cholinv_problem*)p)->dims(nloc, x, y, z, d, c); }); }
int capital_cholinv_stats(void* p, int64_t* bc, int64_t* levels, int64_t* bcdim) { return guarded([&] { ((cholinv_problem*)p)->stats(bc, levels, bcdim); }); }
// TRSM mode (info::solve_with_trsm): potrf + block TRSM + SYRK recursion, no inverse formed (one GPU, or a d x d x c grid: potrf_rec_grid)
int capital_cholinv_set_trsm_mode(void* p, int on) { return guarded([&] { ((cholinv_problem*)p)->set_trsm_mode(on != 0); }); }
int capital_cholinv_destroy(void* p) { return guarded([&] { capital::sync(); delete (cholinv_problem*)p; }); }

void* capital_cacqr_create(int64_t m, int64_t n, int c, int variant, int layout, int num_chunks, int complete_inv, int split, int bc_mult, int serialize_) {
  cacqr_problem* p = nullptr;
  int rc = guarded([&] {
    namespace P = qr::policy::cacqr;
    if (serialize_) p = new cacqr_impl<qr::cacqr<P::Serialize, P::SaveIntermediates>>(m, n, c, variant, layout, num_chunks, complete_inv, split, bc_mult);
    else p = new cacqr_impl<qr::cacqr<P::NoSerialize, P::SaveIntermediates>>(m, n, c, variant, layout, num_chunks, complete_inv, split, bc_mult);
  });
  return rc ? nullptr : p;
}
int capital_cacqr_generate(void* p) { return guarded([&] { ((cacqr_problem*)p)->generate(); }); }
int capital_cacqr_set_A(void* p, const double* host) { return guarded([&] { ((cacqr_problem*)p)->set_A(host); }); }
int capital_cacqr_factor(void* p) { return guarded([&] { ((cacqr_problem*)p)->factor(); }); }
int capital_cacqr_residual(void* p, double* out) { return guarded([&] { *out = ((cacqr_problem*)p)->residual(); }); }
int capital_cacqr_orthogonality(void* p, double* out) { return guarded([&] { *out = ((cacqr_problem*)p)->orthogonality(); }); }
int capital_cacqr_get(void* p, int which, double* host) { return guarded([&] { ((cacqr_problem*)p)->get(which, host); }); }
int capital_cacqr_get_rows(void* p, int which, int64_t row0, int64_t nrows, double* host) { return guarded([&] { ((cacqr_problem*)p)->get_rows(which, row0, nrows, host); }); }
int capital_cacqr_dims(void* p, int64_t* mloc, int64_t* n) { return guarded([&] { ((cacqr_problem*)p)->dims(mloc, n); }); }
int capital_cacqr_destroy(void* p) { return guarded([&] { capital::sync(); delete (cacqr_problem*)p; }); }

}  // extern "C"
