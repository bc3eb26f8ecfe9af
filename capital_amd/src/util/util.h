// util/util.h -- grid helpers of the reference's util class (src/util/util.h:6-40, util.hpp) on device-resident blocks:
// partner exchange ("transpose"), triangle removal by global index, block<->cyclic re-indexing of the base case,
// and the residual reduction used by the validators.
#ifndef CAPITAL_UTIL_H_
#define CAPITAL_UTIL_H_

#include "./../matrix/matrix.h"
#include "topology.h"

class util {
public:
  // util.hpp:232-247: swap the local block with the (y,x) partner; elements are NOT transposed locally -- consumers
  // pass a Trans flag instead.  One xGMI link, both directions.  `staging` must hold mat.num_elems() doubles.
  // On a grid that relays (topo::square::multipath) the exchange is a grid-wide set of pair transfers over all links of the node
  // (capi_pairs_transfer): EVERY rank calls, with the same count -- the diagonal ranks (x == y) exchange nothing themselves but carry
  // other pairs' units; `relay` = capi_pairs_scratch_count(size, count) doubles.
  template <typename MatrixType, typename CommType>
  static void transpose(MatrixType& mat, CommType&& CommInfo) {
    double* relay = nullptr;
    if (CommInfo.multipath) relay = capital::dev_alloc(capi_pairs_scratch_count(CommInfo.size, mat.num_elems()));
    transpose_raw(mat.data(), mat.num_elems(), mat.scratch(), CommInfo, relay);
    if (relay) { capital::sync(); capital::dev_free(relay); }
  }
  template <typename CommType>
  static void transpose_raw(double* buf, int64_t count, double* staging, CommType&& CommInfo, double* relay = nullptr) {
    if (CommInfo.multipath) {
      const std::vector<int> dst = CommInfo.transpose_dst();
      const bool moves = CommInfo.x != CommInfo.y;
      CAPITAL_CHECK(capi_pairs_transfer(CommInfo.world, dst.data(), moves ? buf : nullptr, moves ? staging : nullptr, count, relay));
      if (moves) capital::dev_copy(buf, staging, count);
      return;
    }
    if (CommInfo.x == CommInfo.y) return;
    const int partner = CommInfo.rank_of(CommInfo.y, CommInfo.x, CommInfo.z);
    CAPITAL_CHECK(capi_sendrecv_replace(CommInfo.world, buf, count, partner, staging));
  }

  // util.hpp:266-291: zero by GLOBAL index; for packed structures the unpacked `pad` image is the target
  template <typename MatrixType>
  static void remove_triangle(MatrixType& m, int64_t sliceX, int64_t sliceY, int64_t sliceDim, char dir) {
    double* data = std::is_same<typename MatrixType::StructureType, rect>::value ? m.data() : m.pad();
    CAPITAL_CHECK(capi_remove_triangle(capital::handle(), dir, data, m.num_columns_local(), m.num_rows_local(), sliceX, sliceY, sliceDim));
  }
  // util.hpp:293-318 compares LOCAL indices (its own comparison `j > localHoriz` is against the extent, i.e. a no-op for
  // square blocks); what its callers need is "local strictly-lower is zero", which holds for every upper-triangular
  // factor stored by this layer.  Kept for source compatibility.
  template <typename MatrixType>
  static void remove_triangle_local(MatrixType& m, int64_t, int64_t, int64_t, char dir) {
    double* data = std::is_same<typename MatrixType::StructureType, rect>::value ? m.data() : m.pad();
    if (m.num_rows_local() == m.num_columns_local())
      CAPITAL_CHECK(capi_dtrizero(capital::handle(), dir == 'U' ? CAPI_UPPER : CAPI_LOWER, m.num_rows_local(), data, m.num_rows_local()));
  }

  // util.hpp:105-128 / 203-217 (rect pieces); piece index = slice rank = x + d*y
  static void block_to_cyclic_rect(const double* blocked, double* cyclic, int64_t rows_local, int64_t cols_local, int64_t sliceDim) {
    CAPITAL_CHECK(capi_block_to_cyclic(capital::handle(), blocked, cyclic, rows_local, cols_local, sliceDim));
  }
  static void cyclic_to_block_rect(double* blocked, const double* cyclic, int64_t rows_local, int64_t cols_local, int64_t sliceDim) {
    CAPITAL_CHECK(capi_cyclic_to_block(capital::handle(), blocked, cyclic, rows_local, cols_local, sliceDim));
  }
  // packed upper-triangular pieces (util.hpp:57-102,167-201); num_elems = sliceDim^2 * rows_local (rows_local + 1) / 2 as in the reference
  static void block_to_cyclic_triangle(const double* blocked, double* cyclic, int64_t num_elems, int64_t rows_local, int64_t cols_local, int64_t sliceDim) {
    if (rows_local != cols_local || num_elems != sliceDim * sliceDim * (rows_local * (rows_local + 1) / 2)) throw std::invalid_argument("block_to_cyclic_triangle: square packed pieces expected");
    CAPITAL_CHECK(capi_block_to_cyclic_tri(capital::handle(), blocked, cyclic, rows_local, sliceDim));
  }
  static void cyclic_to_block_triangle(double* blocked, const double* cyclic, int64_t num_elems, int64_t rows_local, int64_t cols_local, int64_t sliceDim) {
    if (rows_local != cols_local || num_elems != sliceDim * sliceDim * (rows_local * (rows_local + 1) / 2)) throw std::invalid_argument("cyclic_to_block_triangle: square packed pieces expected");
    CAPITAL_CHECK(capi_cyclic_to_block_tri(capital::handle(), blocked, cyclic, rows_local, sliceDim));
  }
  // util.hpp:131-164
  static void cyclic_to_local(double* storeT, double* storeTI, int64_t localDimension, int64_t bcDimension, int64_t sliceDim, int64_t sliceRank) {
    CAPITAL_CHECK(capi_cyclic_to_local(capital::handle(), storeT, storeTI, localDimension, bcDimension, sliceDim, sliceRank));
  }

  // util.hpp:249-264
  static int64_t get_next_power2(int64_t v) {
    if ((v & (v - 1)) != 0) { --v; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32; ++v; }
    return v;
  }

  // util.hpp:25-53 specialised to the validators' three lambdas: returns sqrt(sum err^2) / sqrt(sum control^2) where the
  // local sums come from one device reduction and the two scalar Allreduces (C12) run over `slice`.
  //   mode 0: err = X - Y, control = Y  over the entries with GLOBAL row <= col  (cholesky 'U', validate.hpp:37-46)
  //   mode 1: err = X - Y, control = Y  over all entries                          (qr residual, validate.hpp:46-51)
  // X and Y are local blocks with the same shape; padded entries are zero in both and drop out.
  template <typename MatrixType>
  static double residual_local(const MatrixType& X, const MatrixType& Y, int mode, capi_comm_t slice, int64_t sliceX, int64_t sliceY,
                               int64_t sliceDimX, int64_t sliceDimY) {
    double sums[2] = {0, 0};
    const int64_t m = X.num_rows_local(), n = X.num_columns_local();
    if (mode == 0 && (sliceDimX != 1 || sliceDimY != 1 || sliceX || sliceY)) {
      // global-upper selection on a cyclic piece: zero the global-lower part of (X - Y) and Y through scratch copies
      double* e = capital::dev_alloc(m * n);
      double* c = capital::dev_alloc(m * n);
      capital::dev_copy(e, X.data(), m * n);
      capital::dev_copy(c, Y.data(), m * n);
      CAPITAL_CHECK(capi_remove_triangle(capital::handle(), 'U', e, n, m, sliceX, sliceY, sliceDimX));
      CAPITAL_CHECK(capi_remove_triangle(capital::handle(), 'U', c, n, m, sliceX, sliceY, sliceDimX));
      CAPITAL_CHECK(capi_diff_norms(capital::handle(), 0, m, n, e, m, c, m, sums));
      capital::dev_free(e);
      capital::dev_free(c);
    } else {
      CAPITAL_CHECK(capi_diff_norms(capital::handle(), mode == 0 ? 1 : 0, m, n, X.data(), m, Y.data(), m, sums));
    }
    allreduce_host(sums, 2, slice);
    return std::sqrt(sums[0]) / std::sqrt(sums[1]);
  }

  // tiny host-value Allreduce(SUM) (util.hpp:49-50): staged through a device word so that it rides the same RCCL communicator
  static void allreduce_host(double* v, int n, capi_comm_t comm) {
    int size = 1;
    CAPITAL_CHECK(capi_comm_size(comm, &size));
    if (size == 1) return;
    double* d = capital::dev_alloc(n);
    CAPITAL_CHECK(capi_memcpy_h2d(capital::handle(), d, v, sizeof(double) * n));
    CAPITAL_CHECK(capi_allreduce_sum(comm, d, n));
    CAPITAL_CHECK(capi_memcpy_d2h(capital::handle(), v, d, sizeof(double) * n));
    capital::dev_free(d);
  }
};

#endif  // CAPITAL_UTIL_H_
