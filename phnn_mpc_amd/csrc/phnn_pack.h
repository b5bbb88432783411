// phnn_pack.h -- weight blob (include/phnn_mpc.h order, zero-padded to a kernel width) -> the LDS image the kernels
// stage.  ONE source for two executions:
//   * on the host, serially (phnn_create / phnn_update_weights: phnn_mpc.hip);
//   * on the device, by one workgroup (phnn_update_weights_dev: k_pack_image in phnn_pack.hip), so that an optimizer
//     step on GPU-resident parameters needs no device-to-host copy, host packing or upload.
// Every loop is a PK_FOR over independent destination elements (a strided thread loop on the device), reductions
// (max |w|, row-sum norms) go through pk_max (exact in any order), and every value is formed by the same float
// expression on both sides: the two executions give the same image bit for bit, except the handful of constants of the
// canonical models that go through expf / log1pf (R_diag, the mass constants), which may differ in the last bit between
// the host's and the device's math library.
#pragma once
#include "phnn_variants.h"

#define PK_HD __host__ __device__ inline
#if defined(__HIP_DEVICE_COMPILE__)
#define PK_FOR(k, n) for (int k = (int)threadIdx.x; k < (int)(n); k += (int)blockDim.x)
#define PK_SYNC() __syncthreads()
#else
#define PK_FOR(k, n) for (int k = 0; k < (int)(n); ++k)
#define PK_SYNC() ((void)0)
#endif

constexpr int kPackThreads = 1024;

// max over k < n of f(k) >= 0.  red: kPackThreads floats of LDS on the device (unused on the host); every thread
// returns the result.
template <class F>
PK_HD float pk_max(int n, float* red, F f) {
  float m = 0.f;
  PK_FOR(k, n) m = fmaxf(m, f(k));
#if defined(__HIP_DEVICE_COMPILE__)
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = (int)blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
#else
  (void)red;
#endif
  return m;
}

// position pos of a permuted 128-/64-wide row (k-slot order of the split-product fragments) -> hidden unit
PK_HD int pk_unit_of_pos(int pos) {
  const int s = pos / 32, w = pos % 32, q = w / 8, j = w % 8;
  return 32 * s + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
}

template <int HID>
PK_HD void pack_in_frag(float* dst, const float* W, int nin, float sc) {  // sc * W (HID, nin) -> [T][64]
  PK_FOR(e, (HID / 16) * 64) {
    const int nt = e >> 6, lane = e & 63, i = lane & 15, q = lane >> 4;
    dst[e] = q < nin ? W[(size_t)(16 * nt + i) * nin + q] * sc : 0.f;
  }
}
// f16x2 input layer (in_layer_h): [T][64] lanes x 8 halves; lane (i,0): hi(W[u][0..3]) twice, lane (i,1): lo(W[u][0..3])
// twice, lanes q >= 2: zeros
template <int HID>
PK_HD void pack_in_frag_h(float* dstf, const float* W, int nin, float sc) {
  _Float16* dst = reinterpret_cast<_Float16*>(dstf);
  PK_FOR(e, (HID / 16) * 64 * 8) {
    const int j = e & 7, lane = (e >> 3) & 63, nt = e >> 9, i = lane & 15, q = lane >> 4, c = j & 3;
    const float x = (q < 2 && c < nin) ? W[(size_t)(16 * nt + i) * nin + c] * sc : 0.f;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    dst[e] = q == 0 ? h : (q == 1 ? l : (_Float16)0.f);
  }
}
template <int HID>
PK_HD void pack_in_frag_T(float* dst, const float* W, int nout) {  // W (nout, HID): frag of W^T (HID, nout)
  PK_FOR(e, (HID / 16) * 64) {
    const int nt = e >> 6, lane = e & 63, i = lane & 15, q = lane >> 4;
    dst[e] = q < nout ? W[(size_t)q * HID + 16 * nt + i] : 0.f;
  }
}
PK_HD void pack_rows(float* dst, const float* W, int rows, int cols, int ld, float sc = 1.0f) {  // row-major -> ld-padded
  PK_FOR(e, rows * cols) {
    const int r = e / cols, c = e % cols;
    dst[(size_t)r * ld + c] = W[e] * sc;
  }
}
PK_HD void pack_cols_as_rows(float* dst, const float* W, int rows, int cols, int ld, float div = 1.0f) {  // dst[c][r] = W[r][c] / div
  PK_FOR(e, rows * cols) {
    const int r = e / cols, c = e % cols;
    dst[(size_t)c * ld + r] = W[e] / div;
  }
}
PK_HD void pack_copy(float* dst, const float* src, int n, float sc = 1.0f) {
  PK_FOR(k, n) dst[k] = src[k] * sc;
}

PK_HD uint16_t pk_bf16_rne(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
PK_HD float pk_bf16_to_f32(uint16_t b) { return __builtin_bit_cast(float, ((uint32_t)b) << 16); }

// sc * W (HID x HID, row-major) -> three bf16 parts [HID][RS] with the columns in k-slot order: position
// 32s + 8q + j holds unit 32s + (j < 4 ? 4q + j : 16 + 4q + j - 4)  (phnn_kernels.hip.h, "bf16x3 products")
template <int HID>
PK_HD void pack_bf16x3(float* dstf, const float* W, float sc) {
  using I = BfImg<HID>;
  uint16_t* dst = reinterpret_cast<uint16_t*>(dstf);
  PK_FOR(e, HID * HID) {
    const int r = e / HID, pos = e % HID, u = pk_unit_of_pos(pos);
    const float x = W[(size_t)r * HID + u] * sc;
    const uint16_t h = pk_bf16_rne(x);
    const float r1 = x - pk_bf16_to_f32(h);
    const uint16_t m = pk_bf16_rne(r1);
    const uint16_t l = pk_bf16_rne(r1 - pk_bf16_to_f32(m));
    const size_t at = (size_t)r * I::RS + pos;
    dst[at] = h;
    dst[(size_t)I::PART / 2 + at] = m;
    dst[(size_t)I::PART + at] = l;
  }
}

// power of two that brings mx into [0.5, 1) when mx lies outside [0.5, 1024) (0 stays)
PK_HD float pk_pow2_scale(float mx, bool always) {
  int e = 0;
  if (mx > 0.f && (always || mx < 0.5f || mx >= 1024.0f)) (void)__builtin_frexpf(mx, &e);
  return __builtin_ldexpf(1.0f, -e);
}

// sc * W -> two f16 parts of S * (sc * W) in the same permuted layout; returns the power of two S.  S = 1 while
// max|sc W| lies in [0.5, 1024): the hi/lo pair then resolves 2^-25 absolute, i.e. <= 2^-24 of the largest weight, and
// f16's range is far away.  Smaller matrices are scaled up into [0.5, 1) (keeps that relative resolution), larger ones down.
template <int HID>
PK_HD float pack_f16x2(float* dstf, const float* W, float sc, float* red) {
  using I = HfImg<HID>;
  const float mx = pk_max(HID * HID, red, [&](int k) { return __builtin_fabsf(W[k] * sc); });
  const float S = pk_pow2_scale(mx, false);
  _Float16* dst = reinterpret_cast<_Float16*>(dstf);
  PK_FOR(e, HID * HID) {
    const int r = e / HID, pos = e % HID, u = pk_unit_of_pos(pos);
    const float x = (W[(size_t)r * HID + u] * sc) * S;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    const size_t at = (size_t)r * I::RS + pos;
    dst[at] = h;
    dst[(size_t)I::PART / 2 + at] = l;
  }
  return S;
}

// FOLD: Tanh models only -- SiLU / ReLU images carry the plain weights (S = Sb = k1 = 1)
template <int HID, int MM, bool FOLD = true>
PK_HD const float* pack_h2(float* dst, const float* p, int nin, float* red) {  // H_net: consumes W1,b1,W2,b2,W3,b3 from p
  using Y = LayH2<HID, MM>;
  const float* W1 = p; p += (size_t)HID * nin;
  const float* b1 = p; p += HID;
  const float* W2 = p; p += (size_t)HID * HID;
  const float* b2 = p; p += HID;
  const float* W3 = p; p += HID;
  const float* b3 = p; p += 1;
  // 128-wide: tanh's 2 log2(e) is folded into the weights and biases in front of each tanh (kPreScaled in the
  // kernels); every later use of those pre-activations' scale goes through S and k1 below.
  const float k1 = (FOLD && kPreScaled<HID / 16>) ? 2.8853900817779268f : 1.0f;
  float S = 1.0f;  // scale carried by the second pre-activation: k1 x (f16x2: the image's power of two)
  if (MM == MM_BF16X3) pack_bf16x3<HID>(dst + Y::oW2, W2, k1);
  else if (MM == MM_F16X2) S = pack_f16x2<HID>(dst + Y::oW2, W2, k1, red);
  else pack_rows(dst + Y::oW2, W2, HID, HID, Y::LD, k1);
  S *= k1;
  pack_in_frag<HID>(dst + Y::oW1f, W1, nin, k1);
  if (MM == MM_F16X2) pack_in_frag_h<HID>(dst + Y::oW1h, W1, nin, k1);
  pack_copy(dst + Y::oB1, b1, HID, k1);
  pack_copy(dst + Y::oW3, W3, HID);
  // Sb: power of two (<= 1) that keeps the backward-type MFMA inputs g2 = w3 (1-a2^2) and
  // gdot2 = w3 (-2 a2 (1-a2^2)) zdot2 inside f16 range for any weights: |g2| <= max|w3| and, with the
  // Hessian-vector input normalised below 1, |gdot2| <= 0.77 max|w3| ||W2||_inf ||W1||_inf.  1 for ordinary weights.
  float Sb = 1.0f;
  if (MM == MM_F16X2) {
    const float w3max = pk_max(HID, red, [&](int k) { return __builtin_fabsf(W3[k]); });
    const float n1 = pk_max(HID, red, [&](int r) {
      float a = 0.f;
      for (int c = 0; c < nin; ++c) a += __builtin_fabsf(W1[(size_t)r * nin + c]);
      return a;
    });
    const float n2 = pk_max(HID, red, [&](int r) {
      float b = 0.f;
      for (int c = 0; c < HID; ++c) b += __builtin_fabsf(W2[(size_t)r * HID + c]);
      return b;
    });
    const float bound = fmaxf(w3max, 0.77f * w3max * n1 * n2);
    while (bound * Sb > 1024.0f) Sb *= 0.5f;
  }
  PK_FOR(k, HID) {
    dst[Y::oB2 + k] = b2[k] * S;
    dst[Y::oW3B + k] = W3[k] * Sb;
    dst[Y::oW3S + k] = W3[k] * Sb / S;
  }
  pack_cols_as_rows(dst + Y::oW1T, W1, HID, nin, Y::LR, S * Sb);
  PK_FOR(k, 1) {
    dst[Y::oB3] = b3[0];
    dst[Y::oB3 + 1] = 2.8853900817779268f / S;
    dst[Y::oB3 + 2] = 1.0f / k1;
    dst[Y::oB3 + 3] = Sb;
  }
  return p;
}

template <int HID, int MM, bool FOLD = true>
PK_HD const float* pack_h1(float* dst, const float* p, int nin, int nout, float* red) {  // R_net / G_net
  using Y = LayH1<HID, MM>;
  const float* V1 = p; p += (size_t)HID * nin;
  const float* c1 = p; p += HID;
  const float* V2 = p; p += (size_t)nout * HID;
  const float* c2 = p; p += nout;
  if (Y::HF) {
    // Sr * V2 as f16 hi/lo, twice: rows = outputs with the hidden units in k-slot order (forward), and rows =
    // hidden units with the 16 outputs in natural order (transposed product); Sr = 2^k, max|V2| Sr in [0.5, 1)
    const float mx = pk_max(nout * HID, red, [&](int k) { return __builtin_fabsf(V2[k]); });
    const float Sr = pk_pow2_scale(mx, true);
    _Float16* fw = reinterpret_cast<_Float16*>(dst + Y::oV2);
    _Float16* bw = reinterpret_cast<_Float16*>(dst + Y::oV2T);
    PK_FOR(e, nout * HID) {
      const int o = e / HID, pos = e % HID, u = pk_unit_of_pos(pos);
      const float x = V2[(size_t)o * HID + u] * Sr;
      const _Float16 h = (_Float16)x, l = (_Float16)(x - (float)h);
      fw[(size_t)o * Y::RS + pos] = h;
      fw[(size_t)Y::FPART / 2 + (size_t)o * Y::RS + pos] = l;
      bw[(size_t)u * 16 + o] = h;
      bw[(size_t)Y::BPART / 2 + (size_t)u * 16 + o] = l;
    }
    PK_FOR(k, 1) dst[Y::oSc] = 1.0f / Sr;
  } else {
    pack_rows(dst + Y::oV2, V2, nout, HID, Y::LD);
    PK_FOR(k, 1) dst[Y::oSc] = 1.0f;
  }
  const float k1 = (FOLD && kPreScaled<HID / 16>) ? 2.8853900817779268f : 1.0f;  // folded tanh constant (see pack_h2)
  pack_in_frag<HID>(dst + Y::oV1f, V1, nin, k1);
  if (Y::HF) pack_in_frag_h<HID>(dst + Y::oV1h, V1, nin, k1);
  pack_copy(dst + Y::oC1, c1, HID, k1);
  pack_copy(dst + Y::oC2, c2, nout);
  pack_cols_as_rows(dst + Y::oV1T, V1, HID, nin, Y::LR);
  return p;
}

PK_HD float pk_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch softplus, threshold 20

template <class M>
struct PackOf;

template <int N, int HID, bool FIXG, int MM, int MI, int ACT>
struct PackOf<PhnnModel<N, HID, FIXG, MM, MI, ACT>> {
  using M = PhnnModel<N, HID, FIXG, MM, MI, ACT>;
  PK_HD static void run(float* img, const phnn_desc* d, const float* p, float* red) {
    PK_FOR(k, M::IMG) img[k] = 0.f;
    PK_SYNC();
    const float* J = p; p += N * N;
    const float* G = nullptr;
    if (d->fixed_G) { G = p; p += N * MI; }
    constexpr bool FOLD = ACT == ACT_TANH;
    p = pack_h1<HID, MM, FOLD>(img + M::oR, p, N, N * N, red);
    p = pack_h2<HID, MM, FOLD>(img + M::oH, p, N, red);
    if (!d->fixed_G) p = pack_h1<HID, MM, FOLD>(img + M::oGn, p, N, N * MI, red);
    PK_FOR(e, N * N) {
      const int i = e / N, j = e % N;
      img[M::oJ + e] = J[i * N + j] - J[j * N + i];  // src/pHNN.py:83, no 1/2
    }
    if (G) pack_copy(img + M::oG, G, N * MI);  // row-major (N, MI)
  }
};

template <int HID, int MM, int MI, int MT, int ACT>
struct PackOf<CanonModel<HID, MM, MI, MT, ACT>> {
  using M = CanonModel<HID, MM, MI, MT, ACT>;
  PK_HD static void run(float* img, const phnn_desc* d, const float* p, float* red) {
    PK_FOR(k, M::IMG) img[k] = 0.f;
    PK_SYNC();
    const float* Rd = p; p += 4;
    const float* G = p; p += 4 * MI;
    float* c = img + M::oC;
    if (MT == MASS_CARTPOLE) {
      const float log_a = p[0], b = p[1], log_c = p[2];
      p += 3;
      PK_FOR(k, 1) {
        c[0] = expf(log_a) + 1e-3f;  // src/mass_matrix.py:286-288
        c[1] = b;
        c[2] = expf(log_c) + 1e-3f;
      }
    } else if (MT == MASS_CONSTANT) {
      // L = tril(L_tril) with softplus(diag) + 1e-3; M = L L^T; M^-1 = L^-T L^-1  (src/mass_matrix.py:141-152, 183-194)
      const float l00 = pk_softplus(p[0]) + 1e-3f, l10 = p[2], l11 = pk_softplus(p[3]) + 1e-3f;
      p += 4;
      const float i00 = 1.0f / l00, i11 = 1.0f / l11, i10 = -l10 / (l00 * l11);
      float* cw = img + M::oCW;
      PK_FOR(k, 1) {
        c[0] = l00 * l00;
        c[1] = l00 * l10;
        c[2] = l10 * l10 + l11 * l11;
        cw[0] = i00 * i00 + i10 * i10;
        cw[1] = i10 * i11;
        cw[2] = i11 * i11;
      }
    } else {  // M_net.mlp (already padded to 64): 2 -> 64 -> 64 -> out
      const int nout = MT == MASS_DIAGONAL ? 2 : 3;
      float* dm = img + M::oMn;
      const float* W1 = p; p += 64 * 2;
      const float* b1 = p; p += 64;
      const float* W2 = p; p += 64 * 64;
      const float* b2 = p; p += 64;
      const float* Wo = p; p += (size_t)nout * 64;
      const float* bo = p; p += nout;
      pack_in_frag<64>(dm + LayM::oW1f, W1, 2, 1.0f);
      pack_copy(dm + LayM::oB1, b1, 64);
      pack_rows(dm + LayM::oW2, W2, 64, 64, LayM::LD);
      pack_copy(dm + LayM::oB2, b2, 64);
      pack_rows(dm + LayM::oWo, Wo, nout, 64, LayM::LR);
      pack_copy(dm + LayM::oBo, bo, nout);
      pack_in_frag_T<64>(dm + LayM::oWoTf, Wo, nout);
      pack_cols_as_rows(dm + LayM::oW1T, W1, 64, 2, LayM::LR);
    }
    p = pack_h2<HID, MM, ACT == ACT_TANH>(img + M::oH, p, 4, red);
    PK_FOR(i, 4) {
      c[4 + i] = pk_softplus(Rd[i]) + 1e-4f;  // src/pHNN_canonical.py:162
      // softplus'(raw) = sigmoid(raw) (threshold 20 as torch.nn.functional.softplus): the weight-gradient kernels need it
      c[8 + i] = Rd[i] > 20.f ? 1.0f : (float)(1.0 / (1.0 + exp(-(double)Rd[i])));
    }
    pack_copy(c + 12, G, 4 * MI);  // row-major (4, MI)
    (void)d;
  }
};

template <int N, int HID, int MM, int ACT>
struct PackOf<OdeModel<N, HID, MM, ACT>> {
  using M = OdeModel<N, HID, MM, ACT>;
  PK_HD static void run(float* img, const phnn_desc* d, const float* p, float* red) {
    const int nin = N + 1;
    PK_FOR(k, M::IMG) img[k] = 0.f;
    PK_SYNC();
    const float* W1 = p; p += (size_t)HID * nin;
    const float* b1 = p; p += HID;
    const float* W2 = p; p += (size_t)HID * HID;
    const float* b2 = p; p += HID;
    const float* W3 = p; p += (size_t)HID * HID;
    const float* b3 = p; p += HID;
    const float* W4 = p; p += (size_t)N * HID;
    const float* b4 = p; p += N;
    float S2 = 1.0f, S3 = 1.0f;  // power-of-two scales carried by the f16x2 images
    pack_in_frag<HID>(img + M::oW1f, W1, nin, 1.0f);
    pack_copy(img + M::oB1, b1, HID);
    if (MM == MM_F16X2) {
      S2 = pack_f16x2<HID>(img + M::oW2, W2, 1.0f, red);
      S3 = pack_f16x2<HID>(img + M::oW3, W3, 1.0f, red);
    } else {
      pack_rows(img + M::oW2, W2, HID, HID, M::LD);
      pack_rows(img + M::oW3, W3, HID, HID, M::LD);
    }
    pack_copy(img + M::oB2, b2, HID, S2);
    pack_copy(img + M::oB3, b3, HID, S3);
    pack_rows(img + M::oW4r, W4, N, HID, M::LR);
    pack_copy(img + M::oB4, b4, N);
    pack_in_frag_T<HID>(img + M::oW4f, W4, N);
    const float S23 = S2 * S3;
    PK_FOR(e, HID * 4) {
      const int r = e >> 2, c = e & 3;
      if (c < nin) img[M::oW1T + (size_t)c * M::LR + r] = W1[(size_t)r * nin + c] / S23;
    }
    if (M::WIDE) {  // control column of W1: second k-step fragment (k-slot q = 0) and its replicated-row image
      PK_FOR(e, (HID / 16) * 16) {
        const int nt = e >> 4, lane = e & 15;
        img[M::oW1fu + nt * 64 + lane] = W1[(size_t)(16 * nt + lane) * nin + N];
      }
      PK_FOR(r, HID) img[M::oW1Tu + r] = W1[(size_t)r * nin + N] / S23;
    }
    PK_FOR(k, 1) {
      img[M::oSC + 0] = 2.8853900817779268f / S2;
      img[M::oSC + 1] = 2.8853900817779268f / S3;
    }
    (void)d;
  }
};

// device side (phnn_pack.hip): pad the caller's blob to the kernel width (pad_src[k]: index in the original blob of
// entry k of the padded one, -1 = padding), then pack -- one workgroup of kPackThreads threads
struct PackParams {
  const float* orig;   // caller's blob (device)
  const int* pad_src;  // [n_pad]
  int n_pad;
  float* pblob;        // scratch, n_pad floats
  float* img;          // the handle's image
  phnn_desc pdesc;
};
int phnn_pack_launch(int variant, const PackParams& p, hipStream_t st);  // hipError_t as int; -1: no such variant
