// phnn_grad.hip -- instantiates the adjoint kernels (K2 and the single-call VJP) of every variant.  A translation
// unit of its own so that the halves compile in parallel, and so that scheduler flags can be tried on the adjoint
// kernels alone (Makefile: GRAD_SCHED; round 1 shipped max-ILP here, round 2 builds every unit with the default
// scheduler -- DESIGN.md section 9).
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
