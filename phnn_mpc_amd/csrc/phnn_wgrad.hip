// phnn_wgrad.hip -- weight-gradient kernels of the training-side path: the adjoint march
// and the single-evaluation VJP built with the record flag (they stream the tapes the parameter gradient needs to
// HBM), and k_wgrad_reduce / k_wgrad_finish, which turn the records into d loss / d theta.  A translation unit of its
// own so that it compiles in parallel with the other two; the adjoint kernels here use the same scheduler flags as
// phnn_grad.hip.
#define PHNN_ADJOINT_UNIT
#define PHNN_WGRAD_UNIT
#include "phnn_variants.h"

template <class M>
static WgradSet wgrad_set() {
  WgradSet g;
  g.grad[0] = k_rollout_grad<M, PHNN_INTEG_EULER, false, true>;
  g.grad[1] = k_rollout_grad<M, PHNN_INTEG_RK4, false, true>;
  g.grad_t[0] = k_rollout_grad<M, PHNN_INTEG_EULER, true, true>;
  g.grad_t[1] = k_rollout_grad<M, PHNN_INTEG_RK4, true, true>;
  g.mvjp = k_model_vjp<M, true>;
  g.reduce = k_wgrad_reduce<M, false>;
  g.reduce_t = k_wgrad_reduce<M, true>;
  g.tape_floats[0] = StashStep<M, PHNN_INTEG_EULER>::FLOATS;
  g.tape_floats[1] = StashStep<M, PHNN_INTEG_RK4>::SLOT;  // four slots per step = one per record
  g.rec_floats = M::Rec::SIZE;
  g.blob_floats = BlobOf<M>::SIZE;
  g.reduce_waves = M::T;
  g.reduce_lds_bytes = (int)sizeof(float) * WgGeom<M>::LDS_FLOATS;
  return g;
}

bool phnn_wgrad_kernels(int variant, WgradSet* g) {
  switch (variant) {
#define PHNN_WCASE(V, M) \
  case V: *g = wgrad_set<M>(); return true;
    PHNN_WCASE(V_PHNN_4_128_FIX, M_PHNN_4_128_FIX)
    PHNN_WCASE(V_PHNN_4_128_FIX_H, M_PHNN_4_128_FIX_H)
    PHNN_WCASE(V_PHNN_4_64_FIX, M_PHNN_4_64_FIX)
    PHNN_WCASE(V_PHNN_2_64_GNET, M_PHNN_2_64_GNET)
    PHNN_WCASE(V_PHNN_2_64_FIX, M_PHNN_2_64_FIX)
    PHNN_WCASE(V_PHNN_4_128_GNET_H, M_PHNN_4_128_GNET_H)
    PHNN_WCASE(V_PHNN_2_128_GNET_H, M_PHNN_2_128_GNET_H)
    PHNN_WCASE(V_PHNN_2_128_FIX_H, M_PHNN_2_128_FIX_H)
    PHNN_WCASE(V_CANON_128, M_CANON_128)
    PHNN_WCASE(V_CANON_128_H, M_CANON_128_H)
    PHNN_WCASE(V_CANON_64, M_CANON_64)
    PHNN_WCASE(V_PHNN_4_128_FIX_H_M2, M_PHNN_4_128_FIX_H_M2)
    PHNN_WCASE(V_PHNN_4_128_GNET_H_M2, M_PHNN_4_128_GNET_H_M2)
    PHNN_WCASE(V_CANON_128_H_M2, M_CANON_128_H_M2)
    PHNN_WCASE(V_PHNN_4_128_FIX_H_M3, M_PHNN_4_128_FIX_H_M3)
    PHNN_WCASE(V_PHNN_4_128_GNET_H_M3, M_PHNN_4_128_GNET_H_M3)
    PHNN_WCASE(V_CANON_128_H_M3, M_CANON_128_H_M3)
    PHNN_WCASE(V_PHNN_4_128_FIX_H_M4, M_PHNN_4_128_FIX_H_M4)
    PHNN_WCASE(V_PHNN_4_128_GNET_H_M4, M_PHNN_4_128_GNET_H_M4)
    PHNN_WCASE(V_CANON_128_H_M4, M_CANON_128_H_M4)
    PHNN_WCASE(V_CANON_128_H_MCONST, M_CANON_128_H_MCONST)
    PHNN_WCASE(V_CANON_128_H_MDIAG, M_CANON_128_H_MDIAG)
    PHNN_WCASE(V_CANON_128_H_MFULL, M_CANON_128_H_MFULL)
#undef PHNN_WCASE
    default: return false;
  }
}

hipError_t phnn_wgrad_finish(const float* slab, int rows, int PP, const int* map, int P, float* out, int accumulate,
                             hipStream_t stream) {
  const int threads = 256;
  hipLaunchKernelGGL(k_wgrad_finish, dim3((unsigned)((P + threads - 1) / threads)), dim3(threads), 0, stream, slab, rows,
                     PP, map, P, out, accumulate);
  return hipGetLastError();
}
