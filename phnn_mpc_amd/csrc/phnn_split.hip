// phnn_split.hip -- split-tile rollout kernels: the whole-tile kernel templates instantiated with the split models
// (four waves per 16-rollout tile; phnn_kernels.hip.h, "Split-tile models").  Used for small batches, where the
// whole-tile kernels would leave one wave per CU marching alone.  A translation unit of its own (parallel build).
#define PHNN_ADJOINT_UNIT
#include "phnn_variants.h"

template <class M>
static SplitSet split_set() {
  SplitSet g;
  g.fwd[0] = k_rollout_fwd<M, PHNN_INTEG_EULER, false>;
  g.fwd[1] = k_rollout_fwd<M, PHNN_INTEG_RK4, false>;
  g.fwd_stash[0] = k_rollout_fwd<M, PHNN_INTEG_EULER, true>;
  g.fwd_stash[1] = k_rollout_fwd<M, PHNN_INTEG_RK4, true>;
  g.grad[0] = k_rollout_grad<M, PHNN_INTEG_EULER, false>;
  g.grad[1] = k_rollout_grad<M, PHNN_INTEG_RK4, false>;
  g.grad_stash[0] = k_rollout_grad<M, PHNN_INTEG_EULER, true>;
  g.grad_stash[1] = k_rollout_grad<M, PHNN_INTEG_RK4, true>;
  g.lds_floats = M::IMG + 4 * M::SCR + kXchFloats;
  return g;
}

bool phnn_split_kernels(int variant, SplitSet* g) {
  switch (variant) {
    case V_PHNN_4_128_FIX_H: *g = split_set<PhnnSplit<4>>(); return true;
    case V_PHNN_2_128_FIX_H: *g = split_set<PhnnSplit<2>>(); return true;
    case V_CANON_128_H: *g = split_set<CanonSplit<>>(); return true;
    default: return false;
  }
}
