// alg/cholesky/cholinv/policy.h -- policy classes of cholinv (reference src/alg/cholesky/cholinv/policy.h:9-514).
//
//   Serialize / NoSerialize            : storage of the factors handed back in info::R / info::Rinv (packed upper
//                                        triangle vs full local block).  The device recursion always works on full
//                                        local blocks (MFMA tiles want a leading dimension); Serialize packs once at
//                                        the end instead of packing/unpacking around every BLAS call (summa.hpp:216).
//   SaveIntermediates / FlushIntermediates : keep, or release after every factor() call, everything that is not a result: the
//                                        work arena and (with Serialize) the full-storage working images of R and R^-1.
//   ReplicateCommComp / ReplicateComp / NoReplication / NoReplicationOverlap : how a base-case diagonal block that is
//                                        spread over the d x d slice is brought together, factored and handed back.
// Base case on device (all four strategies end in the same R and R^-1; they differ in who computes and what travels):
//   the d*d local pieces come together (Allgather, or Gather to the slice root) -> element-cyclic aggregate
//   (capi_block_to_cyclic[_tri]) -> potrf + trtri on the aggregate -> pieces back out (capi_cyclic_to_block[_tri]; Scatter
//   from the root) -> Bcast over depth where only layer 0 worked -> into R, Rinv.  With Serialize the pieces travel packed.
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

// How the aggregated block is produced and returned (cholinv.h: base_case).
//   gather_all  : Allgather over the slice, every rank assembles, factors and inverts the block (policy.h:176,240);
//                 otherwise Gather to the slice's rank 0, which alone factors and Scatters the pieces back (:322-377).
//   every_layer : each depth layer repeats the work; otherwise layer z == 0 does it and a Bcast over depth follows --
//                 of the two aggregates (ReplicateComp, :288-289) or of each rank's own two pieces (NoReplication*, :398-399).
//   overlap     : the Scatter of R's pieces runs on the communication stream while the root inverts (MPI_Iscatter around
//                 trtri, :470-488); needs potrf and trtri as separate launches instead of the fused base-case kernel.
class ReplicateCommComp {
protected:
  static size_t get_id() { return 0; }
  static constexpr bool gather_all = true, every_layer = true, overlap = false;
};
class ReplicateComp {
protected:
  static size_t get_id() { return 1; }
  static constexpr bool gather_all = true, every_layer = false, overlap = false;
};
class NoReplication {
protected:
  static size_t get_id() { return 2; }
  static constexpr bool gather_all = false, every_layer = false, overlap = false;
};
class NoReplicationOverlap {
protected:
  static size_t get_id() { return 3; }
  static constexpr bool gather_all = false, every_layer = false, overlap = true;
};

}  // namespace cholinv
}  // namespace policy
}  // namespace cholesky

#endif  // CAPITAL_CHOLESKY_POLICY_CHOLINV_H_
