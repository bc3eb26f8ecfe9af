// matrix/serialize.h -- serialize<Structure1,Structure2>::invoke (reference src/matrix/serialize.h:16-70,
// serialize.hpp:12-150): copy a sub-range between two local blocks.  As in the reference, the class's structure
// pair fixes the SHAPE that is copied per column (whole column / i+1 leading entries / from the diagonal down)
// while each matrix's own structure fixes its offsets (offset_local; buffer 2 = the unpacked `pad` image).
// The reference loops memcpy per column on the host; here one HBM-bound kernel does the whole range.
#ifndef CAPITAL_MATRIX_SERIALIZE_H_
#define CAPITAL_MATRIX_SERIALIZE_H_

#include "matrix.h"

template <typename S1, typename S2>
class serialize {
  static_assert(std::is_same<S1, rect>::value || std::is_same<S2, rect>::value || std::is_same<S1, S2>::value,
                "serialize<uppertri,lowertri> / <lowertri,uppertri> do not exist in the reference either");
  static constexpr int shape = (S1::code == CAPI_LOWERTRI || S2::code == CAPI_LOWERTRI)
                                   ? CAPI_LOWERTRI
                                   : ((S1::code == CAPI_UPPERTRI || S2::code == CAPI_UPPERTRI) ? CAPI_UPPERTRI : CAPI_RECT);

public:
  template <typename SrcType, typename DestType>
  static void invoke(const SrcType& src, DestType& dest, typename SrcType::DimensionType ssx, typename SrcType::DimensionType sex,
                     typename SrcType::DimensionType ssy, typename SrcType::DimensionType sey, typename SrcType::DimensionType dsx,
                     typename SrcType::DimensionType dex, typename SrcType::DimensionType dsy, typename SrcType::DimensionType dey,
                     size_t src_buffer = 0, size_t dest_buffer = 0) {
    using T = typename SrcType::ScalarType;
    const T* s = src_buffer == 0 ? src.data() : (src_buffer == 1 ? src.scratch() : src.pad());
    T* d = dest_buffer == 0 ? dest.data() : (dest_buffer == 1 ? dest.scratch() : dest.pad());
    const int ss = src_buffer == 2 ? CAPI_RECT : SrcType::StructureType::code;
    const int ds = dest_buffer == 2 ? CAPI_RECT : DestType::StructureType::code;
    CAPITAL_CHECK(capi_serialize_shape(capital::handle(), shape, ss, ds, s, src.num_columns_local(), src.num_rows_local(), d,
                                       dest.num_columns_local(), dest.num_rows_local(), ssx, sex, ssy, sey, dsx, dex, dsy, dey));
  }
};

#endif  // CAPITAL_MATRIX_SERIALIZE_H_
