// phnn_grad.hip -- instantiates the adjoint kernels (K2 and the single-call VJP) of every variant.  A translation
// unit of its own because these kernels want another instruction scheduler than the forward ones: with
// -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers=1 K2 is 3 % faster and K1 up to 5 % slower
// (same arithmetic, same results), so the Makefile passes those flags to this file only.  It also lets the two halves compile in parallel.
#define PHNN_ADJOINT_UNIT
#include "phnn_variants.h"

template <class M>
static GradSet grad_set() {
  GradSet g;
  g.grad[0] = k_rollout_grad<M, PHNN_INTEG_EULER, false>;
  g.grad[1] = k_rollout_grad<M, PHNN_INTEG_RK4, false>;
  g.grad_stash[0] = k_rollout_grad<M, PHNN_INTEG_EULER, true>;
  g.grad_stash[1] = k_rollout_grad<M, PHNN_INTEG_RK4, true>;
  g.mvjp = k_model_vjp<M>;
  return g;
}

bool phnn_grad_kernels(int variant, GradSet* g) {
  switch (variant) {
#define PHNN_CASE(V, M, NAME) \
  case V: *g = grad_set<M>(); return true;
    PHNN_FOR_EACH_VARIANT(PHNN_CASE)
#undef PHNN_CASE
    default: return false;
  }
}
