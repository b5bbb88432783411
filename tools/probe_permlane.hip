// probe_permlane.hip -- the sum over the four 16-lane rows of a wave (lanes i, i+16, i+32, i+48) with
// v_permlane16_swap_b32 / v_permlane32_swap_b32, two ways:
//   (a) through __builtin_amdgcn_permlane16_swap / permlane32_swap with the same value in both operands: this compiler
//       (AMD clang 22.0.0git, ROCm 7.2) adds the FIRST result to itself (v_add_f32 v1, v1, v1) and the sum comes out as
//       4 v -- the second element of the returned pair is lost;
//   (b) through inline assembly on two registers (what reduce_q / to4_kslots in phnn_kernels.hip.h do): exact.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_permlane.hip -o tools/bin/probe_permlane
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ float allsum4_builtin(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("" : "+v"(b));
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  float s = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
  unsigned c = __builtin_bit_cast(unsigned, s), d = c;
  asm volatile("" : "+v"(d));
  auto r2 = __builtin_amdgcn_permlane32_swap(c, d, false, false);
  return __builtin_bit_cast(float, r2[0]) + __builtin_bit_cast(float, r2[1]);
}
__device__ float allsum4_asm(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  float s = a + b, t = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s), "+v"(t));
  return s + t;
}
__global__ void k(float* o) {
  const float v = (float)(threadIdx.x * threadIdx.x);
  o[threadIdx.x] = allsum4_builtin(v);
  o[64 + threadIdx.x] = allsum4_asm(v);
}
int main() {
  float* d;
  if (hipMalloc(&d, 512) != hipSuccess) return 1;
  k<<<1, 64>>>(d);
  float h[128];
  if (hipMemcpy(h, d, 512, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int w = 0; w < 2; ++w) {
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15;
      float e = 0;
      for (int q = 0; q < 4; ++q) e += (float)((i + 16 * q) * (i + 16 * q));
      bad += h[64 * w + l] != e;
    }
    printf("%s: %d of 64 lanes differ from the row sum\n", w ? "inline assembly" : "builtin        ", bad);
  }
  return 0;
}
