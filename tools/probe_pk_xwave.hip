// probe_pk_xwave.hip -- does a packed-f32 VALU stream of ONE wave disturb v_mfma_f32_16x16x32_f16 results of the OTHER
// wave on the same SIMD (or its own)?  Evidence for DESIGN.md section 9 (kernels built with v_pk_*_f32 were not bitwise
// repeatable at two waves per SIMD).
//
// One workgroup = 512 threads = 8 waves; wave w and w + 4 share SIMD w.  Waves 0-3 run role A, waves 4-7 role B, both
// with exactly known answers:
//   role M : chains of v_mfma_f32_16x16x32_f16 on 4 accumulators, A = alpha, B = beta everywhere -> acc = n 32 alpha beta
//   role P : v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 stream on integers          -> exact counts
//   role V : the same arithmetic with plain v_fma_f32 / v_mul / v_add (control)
//   role X : M and P interleaved in the SAME wave (mfma, pk, mfma, pk ...), independent registers
// Every (A, B) pairing is run; mismatches are counted per role.  Usage: probe_pk_xwave [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

enum { ROLE_NONE = 0, ROLE_M = 1, ROLE_P = 2, ROLE_V = 3, ROLE_X = 4, ROLE_SG = 5, ROLE_MG = 6 };

__device__ unsigned role_m(int n, int al, int be) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(float)al; b[j] = (_Float16)(float)be; }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int k = 0; k < n; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
  }
  const float e = (float)n * 32.0f * (float)(al * be);
  unsigned bad = 0;
  for (int r = 0; r < 4; ++r) bad += (c0[r] != e) + (c1[r] != e) + (c2[r] != e) + (c3[r] != e);
  return bad;
}

// MFMA chains with idle gaps of varying length between the instructions (the matrix pipe goes idle and is re-entered
// while the partner wave's stream is in flight)
__device__ unsigned role_mg(int n, int al, int be) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(float)al; b[j] = (_Float16)(float)be; }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
  for (int k = 0; k < n; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    asm volatile("s_nop 15");
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    if (k & 1) asm volatile("s_nop 9");
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    if (k & 2) asm volatile("s_nop 4");
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
  }
  const float e = (float)n * 3.0f * 32.0f * (float)(al * be);
  unsigned bad = 0;
  for (int r = 0; r < 4; ++r) bad += (c0[r] != e) + (c1[r] != e);
  return bad;
}

// packed stream: x <- x * 1 + 1 (pk_fma), y <- y * 1 (pk_mul), z <- z + 1 (pk_add); values stay small integers
__device__ unsigned role_p(int n) {
  f32x2 x = {0.f, 1.f}, y = {3.f, 5.f}, z = {0.f, 2.f}, x2 = {7.f, 9.f};
  const f32x2 one = {1.f, 1.f};
  for (int k = 0; k < n; ++k) {
    asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n"
                 "v_pk_mul_f32 %1, %1, %4\n"
                 "v_pk_add_f32 %2, %2, %4\n"
                 "v_pk_fma_f32 %3, %3, %4, %4\n"
                 : "+v"(x), "+v"(y), "+v"(z), "+v"(x2) : "v"(one));
  }
  return (x[0] != (float)n) + (x[1] != (float)(n + 1)) + (y[0] != 3.f) + (y[1] != 5.f) + (z[0] != (float)n) +
         (z[1] != (float)(n + 2)) + (x2[0] != (float)(n + 7)) + (x2[1] != (float)(n + 9));
}

__device__ unsigned role_v(int n) {
  float x0 = 0.f, x1 = 1.f, y0 = 3.f, y1 = 5.f, z0 = 0.f, z1 = 2.f, w0 = 7.f, w1 = 9.f;
  const float one = 1.f;
  for (int k = 0; k < n; ++k) {
    asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\n"
                 "v_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\n"
                 "v_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\n"
                 "v_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n"
                 : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1), "+v"(z0), "+v"(z1), "+v"(w0), "+v"(w1) : "v"(one));
  }
  return (x0 != (float)n) + (x1 != (float)(n + 1)) + (y0 != 3.f) + (y1 != 5.f) + (z0 != (float)n) + (z1 != (float)(n + 2)) +
         (w0 != (float)(n + 7)) + (w1 != (float)(n + 9));
}

// f32 "SGEMM" MFMAs (they run on the vector ALUs): acc += 1 * 1 per k-step of the 4x4x1 form (16 blocks)
__device__ unsigned role_sg(int n) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
  for (int k = 0; k < n; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, 1.0f, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, c1, 0, 0, 0);
  }
  unsigned bad = 0;
  for (int r = 0; r < 4; ++r) bad += (c0[r] != (float)n) + (c1[r] != 4.0f * (float)n);
  return bad;
}

// same wave: MFMA chain and packed stream interleaved, independent registers (the compiler pads what it knows of)
__device__ unsigned role_x(int n, int al, int be) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(float)al; b[j] = (_Float16)(float)be; }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
  f32x2 x = {0.f, 1.f}, z = {0.f, 2.f};
  const f32x2 one = {1.f, 1.f};
  for (int k = 0; k < n; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(one));
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(z) : "v"(one));
  }
  const float e = (float)n * 32.0f * (float)(al * be);
  unsigned bad = 0;
  for (int r = 0; r < 4; ++r) bad += (c0[r] != e) + (c1[r] != e);
  return bad + (x[0] != (float)n) + (x[1] != (float)(n + 1)) + (z[0] != (float)n) + (z[1] != (float)(n + 2));
}

__device__ unsigned run_role(int role, int n, int al, int be) {
  switch (role) {
    case ROLE_M: return role_m(n, al, be);
    case ROLE_P: return role_p(n);
    case ROLE_V: return role_v(n);
    case ROLE_X: return role_x(n, al, be);
    case ROLE_SG: return role_sg(n);
    case ROLE_MG: return role_mg(n, al, be);
    default: return 0;
  }
}

__global__ __launch_bounds__(512) void k(int roleA, int roleB, int n, unsigned* bad) {  // bad[0]: role A lanes, bad[1]: role B lanes
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? roleA : roleB;
  // stagger the partners a little differently in every workgroup
  for (int j = 0; j < (blockIdx.x * 7 + wave * 5) % 23; ++j) asm volatile("s_nop 3");
  unsigned b = run_role(role, n, 1 + (blockIdx.x + wave) % 3, 1 + (blockIdx.x / 3) % 3);
  if (b) atomicAdd(&bad[wave < 4 ? 0 : 1], b);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  const char* names[] = {"none", "mfma16", "pk", "valu", "mfma16+pk same wave", "f32 mfma", "mfma16 with gaps"};
  const int pairs[][2] = {{ROLE_M, ROLE_NONE}, {ROLE_M, ROLE_M}, {ROLE_M, ROLE_V}, {ROLE_M, ROLE_P}, {ROLE_P, ROLE_M},
                          {ROLE_X, ROLE_NONE}, {ROLE_X, ROLE_X}, {ROLE_X, ROLE_P}, {ROLE_M, ROLE_X}, {ROLE_SG, ROLE_P},
                          {ROLE_SG, ROLE_V}, {ROLE_P, ROLE_P}, {ROLE_M, ROLE_SG}, {ROLE_X, ROLE_SG},
                          {ROLE_MG, ROLE_NONE}, {ROLE_MG, ROLE_P}, {ROLE_P, ROLE_MG}, {ROLE_MG, ROLE_X}, {ROLE_MG, ROLE_MG}, {ROLE_MG, ROLE_V}, {ROLE_MG, ROLE_M}};
  unsigned* d;
  if (hipMalloc(&d, 2 * sizeof(unsigned)) != hipSuccess) return 1;
  printf("%-24s %-24s %14s %14s   (mismatching values over %d launches of 1024 workgroups, n = 2000 rounds per wave)\n",
         "waves 0-3", "waves 4-7", "bad in 0-3", "bad in 4-7", rounds);
  for (auto& p : pairs) {
    unsigned tot[2] = {0, 0};
    for (int r = 0; r < rounds; ++r) {
      (void)hipMemset(d, 0, 2 * sizeof(unsigned));
      hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, p[0], p[1], 2000, d);
      unsigned h[2];
      if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 1; }
      tot[0] += h[0];
      tot[1] += h[1];
    }
    printf("%-24s %-24s %14u %14u\n", names[p[0]], names[p[1]], tot[0], tot[1]);
  }
  return 0;
}
