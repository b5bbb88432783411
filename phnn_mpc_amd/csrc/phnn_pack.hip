// phnn_pack.hip -- device execution of the weight packing (phnn_pack.h): phnn_update_weights_dev turns a GPU-resident
// parameter blob into the kernels' LDS image with one small launch, in stream order (no host round trip; may be
// captured into a HIP graph).
#include <hip/hip_runtime.h>

#define PHNN_ADJOINT_UNIT  // the non-template kernels of phnn_kernels.hip.h belong to phnn_mpc.hip
#include "phnn_pack.h"

template <class M>
__global__ __launch_bounds__(kPackThreads) void k_pack_image(PackParams p) {
  __shared__ float red[kPackThreads];
  for (int k = (int)threadIdx.x; k < p.n_pad; k += (int)blockDim.x) {
    const int s = p.pad_src[k];
    p.pblob[k] = s >= 0 ? p.orig[s] : 0.f;
  }
  __syncthreads();
  PackOf<M>::run(p.img, &p.pdesc, p.pblob, red);
}

int phnn_pack_launch(int variant, const PackParams& p, hipStream_t st) {
  switch (variant) {
#define PHNN_CASE(V, M, NAME) \
  case V: hipLaunchKernelGGL(k_pack_image<M>, dim3(1), dim3(kPackThreads), 0, st, p); break;
    PHNN_FOR_EACH_VARIANT(PHNN_CASE)
#undef PHNN_CASE
    default: return -1;
  }
  return (int)hipGetLastError();
}
