// alg/cholesky/cholinv/policy.h -- policy classes of cholinv (reference src/alg/cholesky/cholinv/policy.h:9-514).
//
//   Serialize / NoSerialize            : storage of the factors handed back in info::R / info::Rinv (packed upper
//                                        triangle vs full local block).  The device recursion always works on full
//                                        local blocks (MFMA tiles want a leading dimension); Serialize packs once at
//                                        the end instead of packing/unpacking around every BLAS call (summa.hpp:216).
//   SaveIntermediates / FlushIntermediates : keep or release the work arena between factor() calls.
//   ReplicateCommComp / ReplicateComp / NoReplication / NoReplicationOverlap : how a base-case diagonal block that is
//                                        spread over the d x d slice is brought together, factored and handed back.
// Base case on device (all four strategies end in the same R and R^-1):
//   gather the d*d local pieces (C5/C6) -> element-cyclic aggregate (M5, capi_block_to_cyclic)
//   -> capi_dpotrf_trtri on the aggregate (K8+K9 fused) -> this rank's piece back out (M5) -> into R, Rinv.
#ifndef CAPITAL_CHOLESKY_POLICY_CHOLINV_H_
#define CAPITAL_CHOLESKY_POLICY_CHOLINV_H_

namespace cholesky {
namespace policy {
namespace cholinv {

class Serialize {
protected:
  using structure = uppertri;
};
class NoSerialize {
protected:
  using structure = rect;
};

class SaveIntermediates {
protected:
  static constexpr bool keep_arena = true;
};
class FlushIntermediates {
protected:
  static constexpr bool keep_arena = false;
};

// How the aggregated block is produced and returned.  `gather_all`: every rank of the slice assembles and factors the
// block (Allgather, policy.h:176); otherwise one root does and the others receive their piece.  `every_layer`: each
// depth layer repeats the work; otherwise layer z == 0 does it and broadcasts over depth (C7, policy.h:288-289).
class ReplicateCommComp {
protected:
  static size_t get_id() { return 0; }
  static constexpr bool gather_all = true, every_layer = true;
};
class ReplicateComp {
protected:
  static size_t get_id() { return 1; }
  static constexpr bool gather_all = true, every_layer = false;
};
class NoReplication {
protected:
  static size_t get_id() { return 2; }
  static constexpr bool gather_all = false, every_layer = false;
};
class NoReplicationOverlap {
protected:
  static size_t get_id() { return 3; }
  static constexpr bool gather_all = false, every_layer = false;
};

}  // namespace cholinv
}  // namespace policy
}  // namespace cholesky

#endif  // CAPITAL_CHOLESKY_POLICY_CHOLINV_H_
