// Does v_mfma_f32_16x16x32_f16 keep f16 denormal inputs?  A = 2^-20 (f16 denormal) in every element, B = 1:
// D = 32 * 2^-20 = 3.0518e-05 if denormals are honoured, 0 if they are flushed.  Also A = 2^-14 (smallest normal).
// hipcc --offload-arch=gfx950 -O2 -o build/probe/f16_denorm tools/probe_f16_denorm.hip && build/probe/f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float aval, float bval) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float cases[][2] = {{9.5367431640625e-07f, 1.0f}, {6.103515625e-05f, 1.0f}, {1.0f, 9.5367431640625e-07f},
                            {9.5367431640625e-07f, 1024.0f}, {5.9604644775390625e-08f, 1.0f}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, cs[0], cs[1]);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("A=%g B=%g -> D=%g (expected %g)\n", cs[0], cs[1], h, 32.0 * cs[0] * cs[1]);
  }
  return 0;
}
