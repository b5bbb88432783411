// probe_bf16_split.hip -- de-risking probe for the NEXT kernel generation (DESIGN.md 3.3 "next step"): can a
// 3-way bf16 split of both operands on v_mfma_f32_16x16x32_bf16 (matrix pipe, 16x the f32 MFMA rate, co-executes
// with VALU) reproduce an f32 mat-vec to f32 accuracy on real hardware?  Standalone: hipcc -O2
// --offload-arch=gfx950 tools/probe_bf16_split.hip -o /tmp/probe && /tmp/probe
//
//   test 1  one MFMA, random bf16 operands: error of the hardware's internal accumulation vs exact (double)
//   test 2  y = W a (128x128, f32 data) three ways: f32 MFMA chain, bf16x3 with 6 products, bf16x3 with 9;
//           errors relative to sum|w||a| against a double reference
//   test 3  issue rate: cycles per v_mfma_f32_16x16x32_bf16 and per v_mfma_f32_16x16x4_f32 (s_memtime)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ inline uint16_t f2bf(float x) {  // round-to-nearest-even
  uint32_t u = __float_as_uint(x);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ inline float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

union Frag {
  bf16x8 v;
  uint16_t h[8];
};

// A (16 x 32) and B (32 x 16) given as bf16 bit patterns, row-major; D (16x16) f32
__global__ void k_one_mfma(const uint16_t* A, const uint16_t* B, float* D) {
  int l = threadIdx.x, i = l & 15, q = l >> 4;
  Frag a, b;
  for (int j = 0; j < 8; ++j) {
    a.h[j] = A[i * 32 + 8 * q + j];
    b.h[j] = B[(8 * q + j) * 16 + i];
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r];
}

// y[u][col] = sum_k W[u][k] x[k][col], 16 columns (rollouts), K = 128, out 128 -> modes: 0 f32 MFMA, 1 bf16x3 (6), 2 (9)
__global__ void k_matvec(const float* W, const float* X, float* Y, int mode) {
  int l = threadIdx.x, i = l & 15, q = l >> 4;
  for (int nt = 0; nt < 8; ++nt) {
    f32x4 acc = {0, 0, 0, 0};
    if (mode == 0) {
      for (int ks = 0; ks < 32; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(16 * nt + i) * 128 + 4 * ks + q], X[(4 * ks + q) * 16 + i], acc, 0, 0, 0);
    } else {
      for (int s = 0; s < 4; ++s) {
        Frag a[3], b[3];
        for (int j = 0; j < 8; ++j) {
          float w = W[(16 * nt + i) * 128 + 32 * s + 8 * q + j], x = X[(32 * s + 8 * q + j) * 16 + i];
          uint16_t wh = f2bf(w);
          float w1 = w - bf2f(wh);
          uint16_t wm = f2bf(w1);
          uint16_t wl = f2bf(w1 - bf2f(wm));
          uint16_t xh = f2bf(x);
          float x1 = x - bf2f(xh);
          uint16_t xm = f2bf(x1);
          uint16_t xl = f2bf(x1 - bf2f(xm));
          a[0].h[j] = wh; a[1].h[j] = wm; a[2].h[j] = wl;
          b[0].h[j] = xh; b[1].h[j] = xm; b[2].h[j] = xl;
        }
        // smallest terms first
        if (mode == 2) {
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[2].v, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[1].v, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[2].v, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
      }
    }
    for (int r = 0; r < 4; ++r) Y[(16 * nt + 4 * q + r) * 16 + i] = acc[r];
  }
}

__global__ void k_rate(long long* out, int n) {
  Frag a, b;
  for (int j = 0; j < 8; ++j) { a.h[j] = 0x3f80 + threadIdx.x; b.h[j] = 0x3f00 + j; }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < n; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c3, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float fa = 1.0f + threadIdx.x, fb = 0.5f;
  f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
  for (int k = 0; k < n; ++k) {
    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, d3, 0, 0, 0);
  }
  long long t2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = t2 - t1;
  }
  if (c0[0] + c1[0] + c2[0] + c3[0] + d0[0] + d1[0] + d2[0] + d3[0] == 12345.f) out[2] = 1;
}

static uint16_t h_f2bf(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static float h_bf2f(uint16_t b) {
  uint32_t u = ((uint32_t)b) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  // ---- test 1
  std::vector<uint16_t> A(16 * 32), B(32 * 16);
  for (auto& v : A) v = h_f2bf(rnd() * powf(2.f, (float)(rand() % 16) - 8));
  for (auto& v : B) v = h_f2bf(rnd());
  uint16_t *dA, *dB;
  float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k_one_mfma<<<1, 64>>>(dA, dB, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) {
      double ex = 0, sc = 0;
      for (int k = 0; k < 32; ++k) {
        double p = (double)h_bf2f(A[r * 32 + k]) * h_bf2f(B[k * 16 + c]);
        ex += p; sc += fabs(p);
      }
      worst = fmax(worst, fabs(D[r * 16 + c] - ex) / sc);
    }
  printf("test1 one bf16 MFMA (K=32, wide dynamic range): max |err| / sum|a||b| = %.3e  (f32 eps = 5.96e-08)\n", worst);
  // ---- test 2
  std::vector<float> W(128 * 128), X(128 * 16), Y(128 * 16);
  for (auto& v : W) v = rnd() * 0.2f;
  for (int k = 0; k < 128; ++k)
    for (int c = 0; c < 16; ++c) X[k * 16 + c] = tanhf(rnd() * 2.f) * powf(10.f, (float)(c % 4) - 2);
  float *dW, *dX, *dY;
  hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  const char* names[3] = {"f32 MFMA 16x16x4 chain      ", "bf16x3, 6 products          ", "bf16x3, 9 products          "};
  for (int mode = 0; mode < 3; ++mode) {
    k_matvec<<<1, 64>>>(dW, dX, dY, mode);
    hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
    double w2 = 0, sum = 0;
    for (int u = 0; u < 128; ++u)
      for (int c = 0; c < 16; ++c) {
        double ex = 0, sc = 0;
        for (int k = 0; k < 128; ++k) {
          double p = (double)W[u * 128 + k] * X[k * 16 + c];
          ex += p; sc += fabs(p);
        }
        double e = fabs(Y[u * 16 + c] - ex) / sc;
        w2 = fmax(w2, e); sum += e;
      }
    printf("test2 %s max err/scale %.3e  mean %.3e\n", names[mode], w2, sum / 2048);
  }
  // ---- test 3
  long long* dT;
  hipMalloc(&dT, 3 * 8);
  hipMemset(dT, 0, 24);
  k_rate<<<1, 64>>>(dT, 1000);
  long long T[3];
  hipMemcpy(T, dT, 24, hipMemcpyDeviceToHost);
  printf("test3 one wave, 4 independent accumulators: %.1f s_memtime ticks per v_mfma_f32_16x16x32_bf16, %.1f per v_mfma_f32_16x16x4_f32\n",
         T[0] / 4000.0, T[1] / 4000.0);
  return 0;
}
