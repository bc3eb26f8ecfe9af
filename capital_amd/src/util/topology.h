// util/topology.h -- processor grids on the GPUs of one node (reference src/util/topology.h:16-143).
//
// topo::square is the d x d x c grid of the recursive Cholesky / 3-D SUMMA, topo::rect the tunable c x d x c grid of
// CA-CholeskyQR2.  Rank -> (x,y,z) maps and the sub-communicator memberships are the reference's; the communicators
// are RCCL ones (capi_comm_split) living on the context's HIP stream, one process per GPU.  On a fully connected xGMI
// node the row / column / depth partners of a 2x2x2 grid are three different peers, i.e. three different links.
//
// Beyond the reference: it only supports cubic grids for Cholesky (c == d; its SUMMA roots are x==z / y==z).  Here any
// c that divides d is accepted (layer z owns the K-classes q = z, z+c, ...), which gives 4-GPU (d=2,c=1) and, with
// d == 1, pure K-replication (2-GPU: d=1,c=2) decompositions -- see matmult::summa.
#ifndef CAPITAL_TOPOLOGY_H_
#define CAPITAL_TOPOLOGY_H_

#include <algorithm>
#include <cmath>
#include <vector>

#include "shared.h"

namespace topo {

namespace detail {
inline capi_comm_t split(capi_comm_t parent, int color, int key) {
  capi_comm_t c = nullptr;
  CAPITAL_CHECK(capi_comm_split(parent, color, key, &c));
  return c;
}
inline void release(capi_comm_t& c) { if (c) { capi_comm_destroy(c); c = nullptr; } }
inline size_t isqrt_ceil(size_t v) { size_t r = (size_t)std::llround(std::ceil(std::sqrt((double)v))); while (r * r < v) ++r; while (r > 0 && (r - 1) * (r - 1) >= v) --r; return r; }
}  // namespace detail

class square {
public:
  // comm: parent communicator (capital::world()); c: replication depth; layout as in the reference (0, 1, 2)
  square(capi_comm_t comm, size_t c_, size_t layout_ = 0, size_t num_chunks_ = 0) : c(c_), layout(layout_), num_chunks(num_chunks_) {
    CAPITAL_CHECK(capi_comm_rank(comm, &rank));
    CAPITAL_CHECK(capi_comm_size(comm, &size));
    if (c == 0 || size % (int)c) throw std::invalid_argument("topo::square: c must divide the communicator size");
    d = detail::isqrt_ceil(size / c);                                   // topology.h:77
    if (d * d * c != (size_t)size) throw std::invalid_argument("topo::square: size must be d*d*c");
    const size_t TopFaceSize = d * c, FrontFaceSize = d * d;
    if (layout > 2) throw std::invalid_argument("topo::square: layout must be 0, 1 or 2");
    coords(rank, x, y, z);
    if (layout == 2) {
      // 64-rank sub-cubes (topology.h:104-123): only meaningful when the rule yields a bijection onto the d x d x c grid
      // (it does for cubic grids, c == d, up to 64 ranks -- on one node it then coincides with layout 1)
      std::vector<char> seen((size_t)size, 0);
      for (int r = 0; r < size; ++r) {
        size_t px, py, pz;
        coords(r, px, py, pz);
        const size_t id = px + d * py + d * d * pz;
        if (px >= d || py >= d || pz >= c || seen[id]) throw std::invalid_argument("topo::square: layout 2 does not tile this grid (needs a cubic grid, c == d)");
        seen[id] = 1;
      }
    }
    world = comm;
    depth = detail::split(comm, (int)(x + d * y), (int)z);              // same (x,y), ordered by z
    slice = detail::split(comm, (int)z, (int)(x + d * y));              // same z; slice rank = x + d*y
    row = detail::split(slice, (int)y, (int)x);                         // same y (and z), ordered by x
    column = detail::split(slice, (int)x, (int)y);                      // same x (and z), ordered by y
    // Multi-path pair transfers (capi_pairs_transfer): on a grid whose rows / columns (d == 2) or depth fibres (c == 2) are PAIRS,
    // the pair collectives of SUMMA are issued by ALL ranks of `comm` together and cut over every link of the node's mesh instead
    // of one link per pair.  Needs ranks that can relay (size >= 4).  CAPITAL_MULTIPATH=0 keeps the per-pair RCCL calls.
    // (CAPITAL_MULTIPATH=2 also takes the pair algorithms on a 2-rank grid, where nothing can relay: rehearsal of the depth halves)
    const char* mp = getenv("CAPITAL_MULTIPATH");
    multipath = size >= ((mp && atoi(mp) >= 2) ? 2 : 4) && (d == 2 || c == 2) && !(mp && atoi(mp) == 0);
  }
  square(const square&) = delete;
  square& operator=(const square&) = delete;
  ~square() { detail::release(row); detail::release(column); detail::release(slice); detail::release(depth); }

  // grid position of a world rank under this layout (topology.h:80-123)
  void coords(int r, size_t& px, size_t& py, size_t& pz) const {
    const size_t TopFaceSize = d * c, FrontFaceSize = d * d, ur = (size_t)r;
    if (layout == 0) {                                                  // topology.h:80-95
      pz = ur % c; py = ur / TopFaceSize; px = (ur % TopFaceSize) / c;
    } else if (layout == 1) {                                           // topology.h:96-103
      py = ur % d; px = (ur % FrontFaceSize) / d; pz = ur / FrontFaceSize;
    } else {                                                            // topology.h:104-123
      const size_t sub = (size_t)std::min(size, 64);
      const size_t sub_slice = (size_t)std::llround(std::ceil(std::pow((double)sub, 2. / 3.) - 1e-9));
      const size_t sub_dim = (size_t)std::llround(std::ceil(std::pow((double)sub, 1. / 3.) - 1e-9));
      const size_t rmod = ur % sub, rdiv = ur / sub;
      const size_t lx = (rmod % sub_slice) / sub_dim, ly = rmod % sub_dim, lz = rmod / sub_slice;
      const size_t per = c / sub_dim ? c / sub_dim : 1;
      const size_t gx = TopFaceSize >= sub_slice ? (rdiv % (TopFaceSize / sub_slice)) / per : 0;
      const size_t gy = rdiv % per;
      const size_t gz = TopFaceSize >= sub_slice ? rdiv / (TopFaceSize / sub_slice) : 0;
      px = gx * sub_dim + lx; py = gy * sub_dim + ly; pz = gz * sub_dim + lz;
    }
  }
  // world rank of grid position (x,y,z) under this layout (util::transpose's partner rule, util.hpp:237-239)
  int rank_of(size_t px, size_t py, size_t pz) const {
    if (layout == 0) return (int)(pz + c * px + c * d * py);
    if (layout == 1) return (int)(py + d * px + d * d * pz);
    for (int r = 0; r < size; ++r) {
      size_t qx, qy, qz;
      coords(r, qx, qy, qz);
      if (qx == px && qy == py && qz == pz) return r;
    }
    return -1;
  }

  // ---- transfer sets for capi_pairs_transfer: dst[r] for every rank r of `world` (identical on all ranks) ------------------------
  enum axis_t { AX_ROW = 0, AX_COLUMN = 1 };
  bool pairs_along(int axis) const { (void)axis; return multipath && d == 2; }        // rows and columns of two
  bool pairs_in_depth() const { return multipath && c == 2; }
  // broadcast inside every row (column) from the member whose x (y) equals the K-class of its layer at step s, q = z + s c
  // (summa.hpp:185,193: the roots of the reference are x == z and y == z)
  std::vector<int> bcast_dst(int axis, size_t s) const {
    std::vector<int> dst((size_t)size, -1);
    for (int r = 0; r < size; ++r) {
      size_t px, py, pz;
      coords(r, px, py, pz);
      const size_t q = pz + s * c;
      if ((axis == AX_ROW ? px : py) != q) continue;
      dst[(size_t)r] = axis == AX_ROW ? rank_of(1 - px, py, pz) : rank_of(px, 1 - py, pz);
    }
    return dst;
  }
  // every rank sends to the other member of its depth fibre (the two halves of MPI_Allreduce, summa.hpp:236)
  std::vector<int> depth_dst() const {
    std::vector<int> dst((size_t)size, -1);
    for (int r = 0; r < size; ++r) {
      size_t px, py, pz;
      coords(r, px, py, pz);
      dst[(size_t)r] = rank_of(px, py, 1 - pz);
    }
    return dst;
  }
  // every off-diagonal rank sends to its transpose partner (y, x, z) (util.hpp:237-240)
  std::vector<int> transpose_dst() const {
    std::vector<int> dst((size_t)size, -1);
    for (int r = 0; r < size; ++r) {
      size_t px, py, pz;
      coords(r, px, py, pz);
      if (px != py) dst[(size_t)r] = rank_of(py, px, pz);
    }
    return dst;
  }

  capi_comm_t world = nullptr, row = nullptr, column = nullptr, slice = nullptr, depth = nullptr;
  int rank = 0, size = 1;
  size_t c = 1, d = 1, x = 0, y = 0, z = 0, layout = 0, num_chunks = 0;
  bool multipath = false;
};

class rect {
public:
  // c x d x c grid over `comm` (topology.h:16-65).  With c == 1 (the 1-D CholeskyQR2 of BASELINE configs 3 and 5) only
  // `world` carries traffic; the cube / row / column_* / depth communicators matter for the 3-D variants.
  rect(capi_comm_t comm, size_t c_, size_t layout_ = 0, size_t num_chunks_ = 0) : c(c_), layout(layout_), num_chunks(num_chunks_) {
    CAPITAL_CHECK(capi_comm_rank(comm, &rank));
    CAPITAL_CHECK(capi_comm_size(comm, &size));
    if (c == 0 || size % (int)(c * c)) throw std::invalid_argument("topo::rect: c*c must divide the communicator size");
    const size_t SubCubeSize = c * c * c, SubCubeSliceSize = c * c;
    world = comm;
    d = size / (c * c);
    z = rank % c; y = rank / SubCubeSliceSize; x = (rank % SubCubeSliceSize) / c;   // topology.h:46-50
    cube = detail::split(comm, (int)(rank / SubCubeSize), rank);
    int cubeRank = 0;
    CAPITAL_CHECK(capi_comm_rank(cube, &cubeRank));
    depth = detail::split(cube, (int)(cubeRank / c), cubeRank);
    row = detail::split(cube, (int)((cubeRank % c) + c * (cubeRank / SubCubeSliceSize)), cubeRank);
    capi_comm_t column = detail::split(comm, (int)(rank % SubCubeSliceSize), rank);
    slice = detail::split(comm, (int)(rank % c), rank);
    int columnRank = 0;
    CAPITAL_CHECK(capi_comm_rank(column, &columnRank));
    column_contig = detail::split(column, (int)(columnRank / c), columnRank);
    column_alt = detail::split(column, (int)(columnRank % c), columnRank);
    detail::release(column);
  }
  rect(const rect&) = delete;
  rect& operator=(const rect&) = delete;
  ~rect() {
    detail::release(row); detail::release(column_contig); detail::release(column_alt);
    detail::release(depth); detail::release(slice); detail::release(cube);
  }

  capi_comm_t world = nullptr, row = nullptr, column_contig = nullptr, column_alt = nullptr, depth = nullptr, slice = nullptr, cube = nullptr;
  int rank = 0, size = 1;
  size_t c = 1, d = 1, x = 0, y = 0, z = 0, layout = 0, num_chunks = 0;
};

}  // namespace topo

#endif  // CAPITAL_TOPOLOGY_H_
