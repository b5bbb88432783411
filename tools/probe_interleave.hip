// probe_interleave.hip -- within ONE wave, how much vector work hides in the issue gaps of a stream of
// v_mfma_f32_16x16x32_f16 (16 cycles in the matrix pipe)?  Each case runs `n` iterations of 4 x {MFMA, fillers}
// with the order pinned by sched_barrier, one wave on one CU; prints cycles per MFMA slot.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_interleave.hip -o tools/bin/probe_interleave && tools/bin/probe_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
#define SB __builtin_amdgcn_sched_barrier(0)
// fillers as volatile asm so that neither the SLP vectoriser nor the scheduler changes what is measured
#define FMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(k0), "v"(k1))
#define EXP(v) asm volatile("v_exp_f32 %0, %0" : "+v"(v))
#define RCP(v) asm volatile("v_rcp_f32 %0, %0" : "+v"(v))
#define PKFMA(v) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(kk0), "v"(kk1))

template <int MODE>
__global__ void k(int n, long long* out, float* sink, const float* lds_src) {
  __shared__ float sh[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sh[i] = lds_src[i];
  __syncthreads();
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (threadIdx.x + j)); b[j] = (_Float16)(0.5f + j); }
  f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.001f * (threadIdx.x + j);
  f32x2 p[4];
  for (int j = 0; j < 4; ++j) p[j] = f32x2{0.5f + j, 0.25f + threadIdx.x};
  float k0 = 0.999f, k1 = 0.001f;
  f32x2 kk0 = {0.999f, 0.999f}, kk1 = {0.001f, 0.001f};
  f32x4 ld = {0, 0, 0, 0};
  const float* lp = sh + 4 * threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE != 9 && MODE != 10) { c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u], 0, 0, 0); SB; }
      if (MODE == 1 || MODE == 9) { EXP(x[u]); SB; }
      if (MODE == 2 || MODE == 10) { FMA(x[u]); FMA(x[u + 4]); SB; }
      if (MODE == 3) { EXP(x[u]); FMA(x[u + 4]); SB; }
      if (MODE == 4) { PKFMA(p[u]); SB; }
      if (MODE == 5) { ld += *reinterpret_cast<const volatile f32x4*>(lp + 256 * u); SB; }
      if (MODE == 6) { EXP(x[u]); RCP(x[u + 4]); SB; }
      if (MODE == 7) { FMA(x[u]); SB; }
      if (MODE == 8) { FMA(x[u]); FMA(x[u + 4]); FMA(p[u][0]); SB; }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = c[0][0] + c[1][1] + c[2][2] + c[3][3] + ld[0] + ld[3];
  for (int j = 0; j < 8; ++j) r += x[j];
  for (int j = 0; j < 4; ++j) r += p[j][0] + p[j][1];
  sink[threadIdx.x] = r;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  long long* dT; float *dS, *dL;
  hipMalloc(&dT, 8); hipMalloc(&dS, 256); hipMalloc(&dL, 4096 * 4); hipMemset(dL, 0, 4096 * 4);
  const int N = 2000;
  const char* names[11] = {"MFMA only", "MFMA + v_exp", "MFMA + 2 v_fma", "MFMA + v_exp + v_fma", "MFMA + v_pk_fma_f32",
                           "MFMA + ds_read_b128 (+ v_pk_add x2)", "MFMA + v_exp + v_rcp", "MFMA + 1 v_fma", "MFMA + 3 v_fma",
                           "v_exp only", "2 v_fma only"};
  auto launch = [&](int m) {
    switch (m) {
      case 0: k<0><<<1, 64>>>(N, dT, dS, dL); break;  case 1: k<1><<<1, 64>>>(N, dT, dS, dL); break;
      case 2: k<2><<<1, 64>>>(N, dT, dS, dL); break;  case 3: k<3><<<1, 64>>>(N, dT, dS, dL); break;
      case 4: k<4><<<1, 64>>>(N, dT, dS, dL); break;  case 5: k<5><<<1, 64>>>(N, dT, dS, dL); break;
      case 6: k<6><<<1, 64>>>(N, dT, dS, dL); break;  case 7: k<7><<<1, 64>>>(N, dT, dS, dL); break;
      case 8: k<8><<<1, 64>>>(N, dT, dS, dL); break;  case 9: k<9><<<1, 64>>>(N, dT, dS, dL); break;
      case 10: k<10><<<1, 64>>>(N, dT, dS, dL); break;
    }
  };
  for (int m = 0; m < 11; ++m) {
    long long T = 0;
    for (int rep = 0; rep < 3; ++rep) { launch(m); hipMemcpy(&T, dT, 8, hipMemcpyDeviceToHost); }
    printf("%-40s %6.2f s_memtime ticks per slot\n", names[m], (double)T / (4.0 * N));
  }
  return 0;
}
