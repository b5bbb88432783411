// phnn_variants.h -- the kernel variants the library instantiates, shared by the two translation units:
// phnn_mpc.hip (host side, forward kernels) and phnn_grad.hip (adjoint kernels, compiled with another
// instruction-scheduling strategy).
#pragma once
#include "phnn_kernels.hip.h"

enum Variant {
  V_NONE = 0,
  V_PHNN_4_128_FIX,  // cart-pole pHNN (cartpole_mpc_config.yaml)
  V_PHNN_4_64_FIX,
  V_PHNN_2_64_GNET,  // pendulum pHNN with learned G (pendulum_config.yaml)
  V_PHNN_2_64_FIX,
  V_CANON_128,       // canonical cart-pole pHNN
  V_CANON_64,
  V_ODE_2_128,       // ODEFunc(2,1), hidden [128,128,128]
  V_ODE_2_64,
  V_ODE_3_128,
  V_PHNN_4_128_FIX_BF,  // same models, 128x128 products as bf16x3 on the matrix pipe
  V_CANON_128_BF,
  V_PHNN_4_128_FIX_H,   // same models, 128x128 products as f16x2 on the matrix pipe
  V_CANON_128_H,
  V_ODE_2_128_H,
  V_ODE_3_128_H,
  V_ODE_4_128,  // the reference's default ODEFunc(4,1): 5 inputs; f32 only (the f16x2 image would not fit LDS)
  V_PHNN_4_128_GNET_H,  // remaining (n, G) combinations at width 128, f16x2 only
  V_PHNN_2_128_GNET_H,
  V_PHNN_2_128_FIX_H,
  V_PHNN_4_64_FIX_H,  // f16x2 forms of the 64-wide models
  V_PHNN_2_64_GNET_H,
  V_PHNN_2_64_FIX_H,
  V_CANON_64_H,
  V_ODE_2_64_H,
  V_PHNN_4_128_FIX_H_M2,   // two controls per step (m = 2): G is (n, 2), u is (B, H, 2)
  V_PHNN_4_128_GNET_H_M2,
  V_CANON_128_H_M2,
  V_PHNN_4_128_FIX_H_M3,   // three and four controls per step
  V_PHNN_4_128_GNET_H_M3,
  V_CANON_128_H_M3,
  V_PHNN_4_128_FIX_H_M4,
  V_PHNN_4_128_GNET_H_M4,
  V_CANON_128_H_M4,
  V_CANON_128_H_MCONST,  // canonical pHNN with MassMatrixNetwork 'constant' / 'diagonal' / 'full' (src/mass_matrix.py:15-216)
  V_CANON_128_H_MDIAG,
  V_CANON_128_H_MFULL,
  V_PHNN_4_128_FIX_SILU,  // other activations than Tanh (src/NN.py:13 default nn.SiLU; src/baseline_node.py:49-50 relu): all-f32 kernels
  V_PHNN_4_128_FIX_RELU,
  V_CANON_128_SILU,
  V_CANON_128_RELU,
  V_ODE_2_128_RELU,
  V_ODE_4_128_RELU,
  V_PHNN_4_128_FIX_ELU,  // nn.ELU / nn.GELU (src/baseline_node.py:53-56; src/pHNN.py:41 by name)
  V_PHNN_4_128_FIX_GELU,
  V_CANON_128_ELU,
  V_CANON_128_GELU,
  V_ODE_2_128_ELU,
  V_ODE_2_128_GELU,
  V_ODE_4_128_ELU,
  V_ODE_4_128_GELU,
};

using M_PHNN_4_128_FIX = PhnnModel<4, 128, true>;
using M_PHNN_4_64_FIX = PhnnModel<4, 64, true>;
using M_PHNN_2_64_GNET = PhnnModel<2, 64, false>;
using M_PHNN_2_64_FIX = PhnnModel<2, 64, true>;
using M_CANON_128 = CanonModel<128>;
using M_CANON_64 = CanonModel<64>;
using M_ODE_2_128 = OdeModel<2, 128>;
using M_ODE_2_64 = OdeModel<2, 64>;
using M_ODE_3_128 = OdeModel<3, 128>;
using M_PHNN_4_128_FIX_BF = PhnnModel<4, 128, true, MM_BF16X3>;
using M_CANON_128_BF = CanonModel<128, MM_BF16X3>;
using M_PHNN_4_128_FIX_H = PhnnModel<4, 128, true, MM_F16X2>;
using M_CANON_128_H = CanonModel<128, MM_F16X2>;
using M_ODE_2_128_H = OdeModel<2, 128, MM_F16X2>;
using M_ODE_3_128_H = OdeModel<3, 128, MM_F16X2>;
using M_ODE_4_128 = OdeModel<4, 128>;
using M_PHNN_4_128_GNET_H = PhnnModel<4, 128, false, MM_F16X2>;
using M_PHNN_2_128_GNET_H = PhnnModel<2, 128, false, MM_F16X2>;
using M_PHNN_2_128_FIX_H = PhnnModel<2, 128, true, MM_F16X2>;
using M_PHNN_4_64_FIX_H = PhnnModel<4, 64, true, MM_F16X2>;
using M_PHNN_2_64_GNET_H = PhnnModel<2, 64, false, MM_F16X2>;
using M_PHNN_2_64_FIX_H = PhnnModel<2, 64, true, MM_F16X2>;
using M_CANON_64_H = CanonModel<64, MM_F16X2>;
using M_ODE_2_64_H = OdeModel<2, 64, MM_F16X2>;
using M_PHNN_4_128_FIX_H_M2 = PhnnModel<4, 128, true, MM_F16X2, 2>;
using M_PHNN_4_128_GNET_H_M2 = PhnnModel<4, 128, false, MM_F16X2, 2>;
using M_CANON_128_H_M2 = CanonModel<128, MM_F16X2, 2>;
using M_PHNN_4_128_FIX_H_M3 = PhnnModel<4, 128, true, MM_F16X2, 3>;
using M_PHNN_4_128_GNET_H_M3 = PhnnModel<4, 128, false, MM_F16X2, 3>;
using M_CANON_128_H_M3 = CanonModel<128, MM_F16X2, 3>;
using M_PHNN_4_128_FIX_H_M4 = PhnnModel<4, 128, true, MM_F16X2, 4>;
using M_PHNN_4_128_GNET_H_M4 = PhnnModel<4, 128, false, MM_F16X2, 4>;
using M_CANON_128_H_M4 = CanonModel<128, MM_F16X2, 4>;
using M_CANON_128_H_MCONST = CanonModel<128, MM_F16X2, 1, MASS_CONSTANT>;
using M_CANON_128_H_MDIAG = CanonModel<128, MM_F16X2, 1, MASS_DIAGONAL>;
using M_CANON_128_H_MFULL = CanonModel<128, MM_F16X2, 1, MASS_FULL>;
using M_PHNN_4_128_FIX_SILU = PhnnModel<4, 128, true, MM_F32, 1, ACT_SILU>;
using M_PHNN_4_128_FIX_RELU = PhnnModel<4, 128, true, MM_F32, 1, ACT_RELU>;
using M_CANON_128_SILU = CanonModel<128, MM_F32, 1, MASS_CARTPOLE, ACT_SILU>;
using M_CANON_128_RELU = CanonModel<128, MM_F32, 1, MASS_CARTPOLE, ACT_RELU>;
using M_ODE_2_128_RELU = OdeModel<2, 128, MM_F32, ACT_RELU>;
using M_ODE_4_128_RELU = OdeModel<4, 128, MM_F32, ACT_RELU>;
using M_PHNN_4_128_FIX_ELU = PhnnModel<4, 128, true, MM_F32, 1, ACT_ELU>;
using M_PHNN_4_128_FIX_GELU = PhnnModel<4, 128, true, MM_F32, 1, ACT_GELU>;
using M_CANON_128_ELU = CanonModel<128, MM_F32, 1, MASS_CARTPOLE, ACT_ELU>;
using M_CANON_128_GELU = CanonModel<128, MM_F32, 1, MASS_CARTPOLE, ACT_GELU>;
using M_ODE_2_128_ELU = OdeModel<2, 128, MM_F32, ACT_ELU>;
using M_ODE_2_128_GELU = OdeModel<2, 128, MM_F32, ACT_GELU>;
using M_ODE_4_128_ELU = OdeModel<4, 128, MM_F32, ACT_ELU>;
using M_ODE_4_128_GELU = OdeModel<4, 128, MM_F32, ACT_GELU>;

struct GradSet {
  void (*grad[2])(RollParams);     // Euler, RK4: recompute the tape
  void (*grad_stash[2])(RollParams);  // Euler, RK4: K2 reads the tape(s) K1 stashed
  void (*mvjp)(PointParams);
};

// defined in phnn_grad.hip
bool phnn_grad_kernels(int variant, GradSet* g);

// Split-tile kernels (small batches: four waves share one 16-rollout tile; phnn_kernels.hip.h "Split-tile models"),
// defined in phnn_split.hip for the 128-wide f16x2 pHNN (fixed G) and canonical variants.  Bitwise the same results
// and the same stash format as the whole-tile kernels.
struct SplitSet {
  void (*fwd[2])(RollParams);
  void (*grad[2])(RollParams);
  void (*fwd_stash[2])(RollParams);   // Euler, RK4
  void (*grad_stash[2])(RollParams);
  int lds_floats;  // image + 4 x per-wave scratch + exchange area
};
bool phnn_split_kernels(int variant, SplitSet* g);  // false: no split-tile kernels for this variant

// Weight-gradient kernels (training side), defined in phnn_wgrad.hip for the pHNN and canonical
// variants: the adjoint kernels built with the record flag (recompute mode) and the record reduction.
struct WgradSet {
  void (*grad[2])(RollParams);  // K2 + records: Euler, RK4
  void (*grad_t[2])(RollParams);  // the same fed by K1's tapes (a2, q1 stay out of the record)
  void (*mvjp)(PointParams);    // single-evaluation VJP + record
  void (*reduce)(WgradParams);
  void (*reduce_t)(WgradParams);  // reads a2, q1 from the tapes
  int tape_floats[2];  // floats per record slot of the tapes: Euler, RK4
  int rec_floats;     // floats per record
  int blob_floats;    // floats per slab row = size of the (width-padded) weight blob
  int reduce_waves;   // waves per workgroup of the reduce kernel (= hidden width / 16)
  int reduce_lds_bytes;
};
bool phnn_wgrad_kernels(int variant, WgradSet* g);  // false: no weight-gradient kernels for this variant
// launches k_wgrad_finish (phnn_wgrad.hip); returns the launch status
hipError_t phnn_wgrad_finish(const float* slab, int rows, int PP, const int* map, int P, float* out, int accumulate,
                             hipStream_t stream);

#define PHNN_FOR_EACH_VARIANT(X)                                                                                       \
  X(V_PHNN_4_128_FIX, M_PHNN_4_128_FIX, "phnn<n=4,hid=128,fixedG>") \
  X(V_PHNN_4_64_FIX, M_PHNN_4_64_FIX, "phnn<n=4,hid=64,fixedG>") \
  X(V_PHNN_2_64_GNET, M_PHNN_2_64_GNET, "phnn<n=2,hid=64,Gnet>") \
  X(V_PHNN_2_64_FIX, M_PHNN_2_64_FIX, "phnn<n=2,hid=64,fixedG>") \
  X(V_CANON_128, M_CANON_128, "canonical<hid=128>") \
  X(V_CANON_64, M_CANON_64, "canonical<hid=64>") \
  X(V_ODE_2_128, M_ODE_2_128, "odefunc<n=2,hid=128>") \
  X(V_ODE_2_64, M_ODE_2_64, "odefunc<n=2,hid=64>") \
  X(V_ODE_3_128, M_ODE_3_128, "odefunc<n=3,hid=128>") \
  X(V_PHNN_4_128_FIX_BF, M_PHNN_4_128_FIX_BF, "phnn<n=4,hid=128,fixedG,bf16x3>") \
  X(V_CANON_128_BF, M_CANON_128_BF, "canonical<hid=128,bf16x3>") \
  X(V_PHNN_4_128_FIX_H, M_PHNN_4_128_FIX_H, "phnn<n=4,hid=128,fixedG,f16x2>") \
  X(V_CANON_128_H, M_CANON_128_H, "canonical<hid=128,f16x2>") \
  X(V_ODE_2_128_H, M_ODE_2_128_H, "odefunc<n=2,hid=128,f16x2>") \
  X(V_ODE_3_128_H, M_ODE_3_128_H, "odefunc<n=3,hid=128,f16x2>") \
  X(V_ODE_4_128, M_ODE_4_128, "odefunc<n=4,hid=128>") \
  X(V_PHNN_4_128_GNET_H, M_PHNN_4_128_GNET_H, "phnn<n=4,hid=128,Gnet,f16x2>") \
  X(V_PHNN_2_128_GNET_H, M_PHNN_2_128_GNET_H, "phnn<n=2,hid=128,Gnet,f16x2>") \
  X(V_PHNN_2_128_FIX_H, M_PHNN_2_128_FIX_H, "phnn<n=2,hid=128,fixedG,f16x2>") \
  X(V_PHNN_4_64_FIX_H, M_PHNN_4_64_FIX_H, "phnn<n=4,hid=64,fixedG,f16x2>") \
  X(V_PHNN_2_64_GNET_H, M_PHNN_2_64_GNET_H, "phnn<n=2,hid=64,Gnet,f16x2>") \
  X(V_PHNN_2_64_FIX_H, M_PHNN_2_64_FIX_H, "phnn<n=2,hid=64,fixedG,f16x2>") \
  X(V_CANON_64_H, M_CANON_64_H, "canonical<hid=64,f16x2>") \
  X(V_ODE_2_64_H, M_ODE_2_64_H, "odefunc<n=2,hid=64,f16x2>") \
  X(V_PHNN_4_128_FIX_H_M2, M_PHNN_4_128_FIX_H_M2, "phnn<n=4,m=2,hid=128,fixedG,f16x2>") \
  X(V_PHNN_4_128_GNET_H_M2, M_PHNN_4_128_GNET_H_M2, "phnn<n=4,m=2,hid=128,Gnet,f16x2>") \
  X(V_CANON_128_H_M2, M_CANON_128_H_M2, "canonical<m=2,hid=128,f16x2>") \
  X(V_PHNN_4_128_FIX_H_M3, M_PHNN_4_128_FIX_H_M3, "phnn<n=4,m=3,hid=128,fixedG,f16x2>") \
  X(V_PHNN_4_128_GNET_H_M3, M_PHNN_4_128_GNET_H_M3, "phnn<n=4,m=3,hid=128,Gnet,f16x2>") \
  X(V_CANON_128_H_M3, M_CANON_128_H_M3, "canonical<m=3,hid=128,f16x2>") \
  X(V_PHNN_4_128_FIX_H_M4, M_PHNN_4_128_FIX_H_M4, "phnn<n=4,m=4,hid=128,fixedG,f16x2>") \
  X(V_PHNN_4_128_GNET_H_M4, M_PHNN_4_128_GNET_H_M4, "phnn<n=4,m=4,hid=128,Gnet,f16x2>") \
  X(V_CANON_128_H_M4, M_CANON_128_H_M4, "canonical<m=4,hid=128,f16x2>") \
  X(V_CANON_128_H_MCONST, M_CANON_128_H_MCONST, "canonical<hid=128,f16x2,mass=constant>") \
  X(V_CANON_128_H_MDIAG, M_CANON_128_H_MDIAG, "canonical<hid=128,f16x2,mass=diagonal>") \
  X(V_CANON_128_H_MFULL, M_CANON_128_H_MFULL, "canonical<hid=128,f16x2,mass=full>") \
  X(V_PHNN_4_128_FIX_SILU, M_PHNN_4_128_FIX_SILU, "phnn<n=4,hid=128,fixedG,silu>") \
  X(V_PHNN_4_128_FIX_RELU, M_PHNN_4_128_FIX_RELU, "phnn<n=4,hid=128,fixedG,relu>") \
  X(V_CANON_128_SILU, M_CANON_128_SILU, "canonical<hid=128,silu>") \
  X(V_CANON_128_RELU, M_CANON_128_RELU, "canonical<hid=128,relu>") \
  X(V_ODE_2_128_RELU, M_ODE_2_128_RELU, "odefunc<n=2,hid=128,relu>") \
  X(V_ODE_4_128_RELU, M_ODE_4_128_RELU, "odefunc<n=4,hid=128,relu>") \
  X(V_PHNN_4_128_FIX_ELU, M_PHNN_4_128_FIX_ELU, "phnn<n=4,hid=128,fixedG,elu>") \
  X(V_PHNN_4_128_FIX_GELU, M_PHNN_4_128_FIX_GELU, "phnn<n=4,hid=128,fixedG,gelu>") \
  X(V_CANON_128_ELU, M_CANON_128_ELU, "canonical<hid=128,elu>") \
  X(V_CANON_128_GELU, M_CANON_128_GELU, "canonical<hid=128,gelu>") \
  X(V_ODE_2_128_ELU, M_ODE_2_128_ELU, "odefunc<n=2,hid=128,elu>") \
  X(V_ODE_2_128_GELU, M_ODE_2_128_GELU, "odefunc<n=2,hid=128,gelu>") \
  X(V_ODE_4_128_ELU, M_ODE_4_128_ELU, "odefunc<n=4,hid=128,elu>") \
  X(V_ODE_4_128_GELU, M_ODE_4_128_GELU, "odefunc<n=4,hid=128,gelu>")
