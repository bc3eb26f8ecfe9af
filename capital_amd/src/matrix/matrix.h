// matrix/matrix.h -- local block of an element-cyclically distributed matrix, resident in HBM.
//
// Same public surface as the reference's matrix<ScalarT,DimensionT,StructurePolicy> (src/matrix/matrix.h:9-97):
// local (column i, row j) is global (x + i*gridX, y + j*gridY); column-major with ld = num_rows_local(); at most one
// padded row/column when the grid does not divide the dimension (matrix.hpp:8-11).  Three buffers data/scratch/pad as
// in the reference, but scratch and pad are allocated on first use: at n = 65536 a block is 32 GiB and the
// reference's eager 2-3x allocation is the difference between fitting in 288 GB and not.
#ifndef CAPITAL_MATRIX_H_
#define CAPITAL_MATRIX_H_

#include "structure.h"

class OffloadEachGemm;   // tag types the reference declares and never defines (src/alg/alg.h:9-11); kept for source
class OffloadImmediate;  // compatibility.  Here every matrix IS device-resident.

template <typename ScalarT = double, typename DimensionT = int64_t, typename StructurePolicy = rect, typename OffloadPolicy = OffloadImmediate>
class matrix : public StructurePolicy {
  static_assert(std::is_same<ScalarT, double>::value, "only double is specialised, as in the reference's engines");

public:
  using ScalarType = ScalarT;
  using DimensionType = DimensionT;
  using StructureType = StructurePolicy;
  using OffloadType = OffloadPolicy;

  matrix() {}
  // regular constructor: global extents + grid extents (matrix.hpp:5-20)
  matrix(DimensionType gX, DimensionType gY, int64_t pX, int64_t pY) { shape_from_global(gX, gY, pX, pY); allocate(); }
  // injection constructors (matrix.hpp:57-92): local extents given; data == nullptr allocates, otherwise the DEVICE
  // pointer is borrowed (never freed here)
  matrix(ScalarType* data, DimensionType dX, DimensionType dY, DimensionType pX, DimensionType pY) {
    _dimensionX = dX; _dimensionY = dY; _globalDimensionX = dX * pX; _globalDimensionY = dY * pY;
    _numElems = StructurePolicy::_num_elems(dX, dY);
    if (data) { _data = data; _owns = false; _filled = true; } else allocate();
  }
  matrix(const matrix& rhs) { copy_from(rhs); }
  matrix(matrix&& rhs) noexcept { move_from(std::move(rhs)); }
  matrix& operator=(const matrix& rhs) { if (this != &rhs) { _destroy_(); copy_from(rhs); } return *this; }
  matrix& operator=(matrix&& rhs) noexcept { if (this != &rhs) { _destroy_(); move_from(std::move(rhs)); } return *this; }
  ~matrix() { _destroy_(); }

  // two-phase construction used by the algorithm packs (matrix.hpp:147-160): no-op once filled
  void _register_(DimensionType gX, DimensionType gY, int64_t pX, int64_t pY) {
    if (!_filled) { shape_from_global(gX, gY, pX, pY); allocate(); }
  }
  void _destroy_() {
    if (_owns) capital::dev_free(_data);
    capital::dev_free(_scratch);
    capital::dev_free(_pad);
    _data = _scratch = _pad = nullptr;
    _owns = false;
    _filled = false;
  }

  inline ScalarType*& data() { return _data; }
  inline ScalarType* data() const { return _data; }
  inline ScalarType*& scratch() { ensure_scratch(); return _scratch; }
  inline ScalarType* scratch() const { const_cast<matrix*>(this)->ensure_scratch(); return _scratch; }
  inline ScalarType*& pad() { ensure_pad(); return _pad; }
  inline ScalarType* pad() const { const_cast<matrix*>(this)->ensure_pad(); return _pad; }
  inline DimensionType num_elems() const { return _numElems; }
  inline DimensionType num_elems(DimensionType rX, DimensionType rY) const { return StructurePolicy::_num_elems(rX, rY); }
  inline DimensionType num_rows_local() const { return _dimensionY; }
  inline DimensionType num_columns_local() const { return _dimensionX; }
  inline DimensionType num_rows_global() const { return _globalDimensionY; }
  inline DimensionType num_columns_global() const { return _globalDimensionX; }
  inline DimensionType offset_local(DimensionType x, DimensionType y, size_t buffer = 0) const {
    return buffer != 2 ? StructurePolicy::_offset(x, y, _dimensionX, _dimensionY) : rect::_offset(x, y, _dimensionX, _dimensionY);
  }
  inline void swap() { ensure_scratch(); std::swap(_data, _scratch); }
  inline void swap_pad() { ensure_scratch(); ensure_pad(); std::swap(_scratch, _pad); }
  inline bool filled() const { return _filled; }

  // generators (matrix.hpp:227-243 -> structure.hpp:36-129), run on the device, bit-identical values
  void distribute_random(int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key) {
    static_assert(std::is_same<StructurePolicy, rect>::value, "generators fill rect blocks (as the reference benches do)");
    CAPITAL_CHECK(capi_distribute_random(capital::handle(), _data, _dimensionX, _dimensionY, _globalDimensionX, _globalDimensionY, px, py, PX, PY, key));
  }
  void distribute_symmetric(int64_t px, int64_t py, int64_t PX, int64_t PY, int64_t key, bool diagonallyDominant) {
    static_assert(std::is_same<StructurePolicy, rect>::value, "generators fill rect blocks (as the reference benches do)");
    CAPITAL_CHECK(capi_distribute_symmetric(capital::handle(), _data, _dimensionX, _dimensionY, _globalDimensionX, _globalDimensionY, px, py, PX, PY, key, diagonallyDominant));
  }
  void distribute_identity(int64_t px, int64_t py, int64_t PX, int64_t PY, ScalarType val = 1.) {
    static_assert(std::is_same<StructurePolicy, rect>::value, "generators fill rect blocks (as the reference benches do)");
    CAPITAL_CHECK(capi_distribute_identity(capital::handle(), _data, _dimensionX, _dimensionY, _globalDimensionX, _globalDimensionY, px, py, PX, PY, val));
  }
  // host copies for validation / fixtures (synchronous)
  std::vector<ScalarType> to_host() const {
    std::vector<ScalarType> h((size_t)_numElems);
    CAPITAL_CHECK(capi_memcpy_d2h(capital::handle(), h.data(), _data, sizeof(ScalarType) * (size_t)_numElems));
    return h;
  }
  void from_host(const ScalarType* h) { CAPITAL_CHECK(capi_memcpy_h2d(capital::handle(), _data, h, sizeof(ScalarType) * (size_t)_numElems)); }

private:
  void shape_from_global(DimensionType gX, DimensionType gY, int64_t pX, int64_t pY) {
    _dimensionX = gX / pX + ((gX % pX) ? 1 : 0);
    _dimensionY = gY / pY + ((gY % pY) ? 1 : 0);
    _globalDimensionX = gX;
    _globalDimensionY = gY;
    _numElems = StructurePolicy::_num_elems(_dimensionX, _dimensionY);
  }
  void allocate() {
    _data = capital::dev_alloc(_numElems);
    capital::dev_zero(_data, _numElems);   // structure.hpp:7
    _owns = true;
    _filled = true;
  }
  void ensure_scratch() {
    if (!_scratch) { _scratch = capital::dev_alloc(_numElems); capital::dev_zero(_scratch, _numElems); }
  }
  void ensure_pad() {
    const DimensionType n = StructurePolicy::_pad_elems(_dimensionX, _dimensionY);
    if (!_pad && n > 0) { _pad = capital::dev_alloc(n); capital::dev_zero(_pad, n); }
  }
  void copy_from(const matrix& rhs) {
    _dimensionX = rhs._dimensionX; _dimensionY = rhs._dimensionY; _numElems = rhs._numElems;
    _globalDimensionX = rhs._globalDimensionX; _globalDimensionY = rhs._globalDimensionY;
    if (rhs._filled) { _data = capital::dev_alloc(_numElems); capital::dev_copy(_data, rhs._data, _numElems); _owns = true; _filled = true; }
  }
  void move_from(matrix&& rhs) {
    _dimensionX = rhs._dimensionX; _dimensionY = rhs._dimensionY; _numElems = rhs._numElems;
    _globalDimensionX = rhs._globalDimensionX; _globalDimensionY = rhs._globalDimensionY;
    _data = rhs._data; _scratch = rhs._scratch; _pad = rhs._pad; _owns = rhs._owns; _filled = rhs._filled;
    rhs._data = rhs._scratch = rhs._pad = nullptr; rhs._owns = false; rhs._filled = false;
  }

  ScalarType* _data = nullptr;
  ScalarType* _scratch = nullptr;
  ScalarType* _pad = nullptr;
  bool _owns = false, _filled = false;
  DimensionType _numElems = 0, _dimensionX = 0, _dimensionY = 0, _globalDimensionX = 0, _globalDimensionY = 0;
};

#endif  // CAPITAL_MATRIX_H_
