// alg/qr/cacqr/policy.h -- policy classes of cacqr (reference src/alg/qr/cacqr/policy.h:9-183).
//   Serialize / NoSerialize : R handed back packed (upper triangle) or as a full n x n block; with Serialize the Gram
//                             matrix also crosses the links packed (policy.h:77-85: n(n+1)/2 instead of n^2 doubles).
//   Save / FlushIntermediates : keep or release the n x n work blocks between factor() calls.
#ifndef CAPITAL_QR_POLICY_CACQR_H_
#define CAPITAL_QR_POLICY_CACQR_H_

namespace qr {
namespace policy {
namespace cacqr {

class NoSerialize {
protected:
  using structure = rect;
  static constexpr bool packed_gram = false;
};
class Serialize {
protected:
  using structure = uppertri;
  static constexpr bool packed_gram = true;
};
class SaveIntermediates {
protected:
  static constexpr bool keep_work = true;
};
class FlushIntermediates {
protected:
  static constexpr bool keep_work = false;
};

}  // namespace cacqr
}  // namespace policy
}  // namespace qr

#endif  // CAPITAL_QR_POLICY_CACQR_H_
