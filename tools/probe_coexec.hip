// probe_coexec.hip -- do matrix-pipe MFMAs of one wave overlap with VALU work of the OTHER wave on the same SIMD?
// One workgroup of 512 threads on one CU (2 waves per SIMD: wave w and w+4 share SIMD).  Waves 0-3 run `na` rounds
// of role A, waves 4-7 `nb` rounds of role B; the kernel's wall time (s_memtime) is compared for A alone, B alone, both.
//   roles: 0 = nothing, 1 = bf16 MFMA 16x16x32 (4 independent accumulators), 2 = f32 MFMA 16x16x4, 3 = VALU v_fma_f32,
//          4 = bf16 MFMA 32x32x16
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// bf16 MFMA 16x16x32 stream that leaves the vector issue port alone for a while after each MFMA:
// Y = 1: s_nop 0, 2: s_nop 1, 3: two scalar moves
template <int Y>
__device__ float mfma_yield(int n, float seed) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(0.5f + j); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
#define YIELD() do { if (Y == 1) asm volatile("s_nop 0"); else if (Y == 2) asm volatile("s_nop 1"); else asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0" ::: "s20", "s21"); __builtin_amdgcn_sched_barrier(0); } while (0)
  for (int k = 0; k < n; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); YIELD();
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0); YIELD();
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); YIELD();
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0); YIELD();
  }
  return c0[0] + c1[1] + c2[2] + c3[3];
}

__device__ float run_role(int role, int n, float seed) {
  float acc = seed;
  if (role == 1) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(0.5f + j); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int k = 0; k < n; ++k) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (role == 2) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int k = 0; k < n; ++k) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.5f, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.5f, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.5f, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 0.5f, c3, 0, 0, 0);
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3];
  } else if (role == 3) {
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    for (int k = 0; k < n; ++k) {
      x0 = __builtin_fmaf(x0, 0.999f, 0.001f); x1 = __builtin_fmaf(x1, 0.999f, 0.001f);
      x2 = __builtin_fmaf(x2, 0.999f, 0.001f); x3 = __builtin_fmaf(x3, 0.999f, 0.001f);
      x4 = __builtin_fmaf(x4, 0.999f, 0.001f); x5 = __builtin_fmaf(x5, 0.999f, 0.001f);
      x6 = __builtin_fmaf(x6, 0.999f, 0.001f); x7 = __builtin_fmaf(x7, 0.999f, 0.001f);
    }
    acc = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  } else if (role == 5) { acc = mfma_yield<1>(n, seed);
  } else if (role == 6) { acc = mfma_yield<2>(n, seed);
  } else if (role == 7) { acc = mfma_yield<3>(n, seed);
  } else if (role == 4) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(0.5f + j); }
    f32x16 c0 = {}, c1 = {};
    for (int k = 0; k < n; ++k) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
    acc = c0[0] + c1[5];
  }
  return acc;
}

__global__ __launch_bounds__(512) void k(int roleA, int na, int roleB, int nb, long long* out, float* sink) {
  int wave = threadIdx.x >> 6;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  float r = wave < 4 ? run_role(roleA, na, 1.0f + threadIdx.x) : run_role(roleB, nb, 1.0f + threadIdx.x);
  sink[threadIdx.x] = r;
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  long long* dT; float* dS;
  hipMalloc(&dT, 8); hipMalloc(&dS, 512 * 4);
  auto run = [&](int ra, int na, int rb, int nb) {
    long long T = 0;
    for (int rep = 0; rep < 2; ++rep) { k<<<1, 512>>>(ra, na, rb, nb, dT, dS); hipMemcpy(&T, dT, 8, hipMemcpyDeviceToHost); }
    return (double)T;
  };
  const int N = 4000;
  const char* names[8] = {"-", "bf16 16x16x32", "f32 16x16x4", "VALU fma", "bf16 32x32x16", "bf16+s_nop0", "bf16+s_nop1", "bf16+2 s_mov"};
  struct { int a, na, b, nb; } cases[] = {
      {1, N, 0, 0}, {3, 0, 3, 4 * N}, {1, N, 3, 4 * N}, {1, N, 1, N},
      {2, N, 0, 0}, {2, N, 3, 4 * N}, {4, N, 0, 0}, {4, N, 3, 4 * N}, {4, N, 4, N}, {3, 4 * N, 3, 4 * N},
      {5, N, 0, 0}, {5, N, 3, 4 * N}, {6, N, 0, 0}, {6, N, 3, 4 * N}, {7, N, 0, 0}, {7, N, 3, 4 * N}, {1, N, 3, 2 * N}, {6, N, 3, 2 * N}};
  for (auto c : cases) {
    double t = run(c.a, c.na, c.b, c.nb);
    printf("waves0-3: %-14s x%-6d | waves4-7: %-14s x%-6d -> %10.0f ticks\n", names[c.a], c.a ? (c.a == 3 ? c.na * 8 : c.na * (c.a == 4 ? 2 : 4)) : 0,
           names[c.b], c.b ? (c.b == 3 ? c.nb * 8 : c.nb * (c.b == 4 ? 2 : 4)) : 0, t);
  }
  return 0;
}
