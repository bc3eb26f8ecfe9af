// matrix/structure.h -- structure policies of the distributed-matrix type (reference src/matrix/structure.h:8-72,
// structure.hpp): element counts and packed offsets for rect / uppertri / lowertri local blocks.  The arithmetic is
// the reference's; storage lives in HBM and is filled by device generators (structure.hpp:36-129 -> capi_distribute_*).
#ifndef CAPITAL_MATRIX_STRUCTURE_H_
#define CAPITAL_MATRIX_STRUCTURE_H_

#include "./../util/shared.h"

class rect {
public:
  static constexpr int code = CAPI_RECT;
  template <typename U> static inline U _num_elems(U rangeX, U rangeY) { return rangeX * rangeY; }
  template <typename U> static inline U _offset(U x, U y, U /*dimX*/, U dimY) { return x * dimY + y; }
  template <typename U> static inline U _pad_elems(U, U) { return 0; }
};

class uppertri {
public:
  static constexpr int code = CAPI_UPPERTRI;
  template <typename U> static inline U _num_elems(U rangeX, U /*rangeY*/) { return (rangeX * (rangeX + 1)) >> 1; }
  template <typename U> static inline U _offset(U x, U y, U, U) { return ((x * (x + 1)) >> 1) + y; }
  template <typename U> static inline U _pad_elems(U dimX, U dimY) { return dimX * dimY; }  // unpacked image (structure.hpp:149)
};

class lowertri {
public:
  static constexpr int code = CAPI_LOWERTRI;
  template <typename U> static inline U _num_elems(U rangeX, U /*rangeY*/) { return (rangeX * (rangeX + 1)) >> 1; }
  template <typename U> static inline U _offset(U x, U y, U, U dimY) { return x * dimY + y - (x * (x + 1) / 2); }
  template <typename U> static inline U _pad_elems(U dimX, U dimY) { return dimX * dimY; }
};

#endif  // CAPITAL_MATRIX_STRUCTURE_H_
