// phnn_kernels.hip.h -- device side of the MI355X (gfx950) rollout engine.
//
// Mapping: one 64-lane wave marches 16 rollouts through the whole horizon.  Every
// dense layer of the small MLPs is evaluated for the 16 rollouts at once on the matrix cores with
// v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fmaf chain), in the orientation
//        D[unit][rollout] = sum_k W[unit][k] * act[k][rollout]
// so that the accumulator a layer leaves in registers (lane (i,q) reg r of tile t = unit 16t+4q+r of
// rollout i) is bit-for-bit the B operand of the next layer's MFMAs: activations never leave registers,
// there are no transposes and no LDS round trips between layers.  Weights are the A operand and are read
// from LDS images staged once per workgroup; one padded row-major image serves both W*a (ds_read_b128,
// rows on lanes) and W^T*g (ds_read_b32, columns on lanes).
//
// Lane geometry inside a wave:  i = lane & 15 (rollout / A-row / D-column),  q = lane >> 4 (k-slot / D-row
// group).  Per-rollout small vectors (state, costate, dH, ...) are held redundantly by the 4 lanes
// (i, q=0..3) of a rollout as f32x4 "all components in every lane".
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/phnn_mpc.h"

#define DEV __device__ __forceinline__
// The K1 -> K2 tape is streamed once each way at large batches: non-temporal stores / loads (worth 3 % each over plain
// ones there).  PHNN_CACHED_TAPE (a translation-unit switch) uses plain accesses instead.
#ifdef PHNN_CACHED_TAPE
#define PHNN_NT_STORE(v, p) (*(p) = (v))
#define PHNN_NT_LOAD(p) (*(p))
#else
#define PHNN_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#define PHNN_NT_LOAD(p) __builtin_nontemporal_load((p))
#endif

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int kTileB = 16;       // rollouts per wave
#ifndef PHNN_KMAXWAVES
#define PHNN_KMAXWAVES 8
#endif
constexpr int kMaxWaves = PHNN_KMAXWAVES;  // waves per workgroup (8 = 2 per SIMD; 12 = 3 per SIMD is an experiment: DESIGN.md section 9)
constexpr int kScrFloats = 16 * 20;  // per-wave LDS scratch: 16 rollouts x (16 + 4 pad) floats

struct Lane {
  int lane, i, q;
  int w;        // split-tile kernels: which quarter of every hidden vector this wave owns (tiles 2w, 2w+1); else 0
  float* xch;   // split-tile kernels: LDS exchange area of the workgroup; else null
};

template <int T>
struct Act {
  f32x4 v[T];
};

DEV f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

DEV f32x4 splat4(float s) { return f32x4{s, s, s, s}; }

// 1 - a^2 (derivative of tanh at its output) as one fused op per element instead of mul + sub
DEV f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
DEV f32x4 dtanh(f32x4 a) { return __builtin_elementwise_fma(-a, a, f32x4{1.0f, 1.0f, 1.0f, 1.0f}); }

DEV float sel4(f32x4 v, int q) { return q == 0 ? v[0] : (q == 1 ? v[1] : (q == 2 ? v[2] : v[3])); }

// tanh in float32.  f32 MFMA and VALU share the SIMD's vector ALUs on gfx950 (SQ_VALU_MFMA_COEXEC_CYCLES = 0
// for this kernel), so every VALU cycle spent here is a cycle the MFMAs do not get: tanh is 384 evaluations
// per rollout-step and was 80 % of the non-MFMA vector time with the two-branch form.
//   128-wide models : 1 - 2/(1 + 2^(2x*log2 e)), 5 instructions, abs. error <= 2.5e-7 everywhere (relative error
//                     grows towards x = 0 but the absolute error is what propagates through the sums)
//   64-wide models  : odd minimax polynomial below 0.4 (rel. 6e-8) + the same formula above (abs. 1e-7): tanh is a
//                     larger share of their arithmetic error budget and a smaller share of their time; the trained
//                     pendulum model (long, large swings) sits at 0.37 of the cost tolerance with it, 0.88 without
DEV float tanh_scaled(float x, float c) {  // tanh(x * c / (2 log2 e)): the pre-activation carries a power-of-two scale
  float e = __builtin_amdgcn_exp2f(x * c);
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

template <bool ACCURATE>
DEV float tanh_dev(float x) {
  if (ACCURATE) {
    float ax = __builtin_fabsf(x);
    float s = x * x;
    float p = -0.007265716325491667f;
    p = __builtin_fmaf(p, s, 0.021598778665065765f);
    p = __builtin_fmaf(p, s, -0.053949106484651566f);
    p = __builtin_fmaf(p, s, 0.13333284854888916f);
    p = __builtin_fmaf(p, s, -0.3333333432674408f);
    float small = __builtin_fmaf(ax * s, p, ax);
    float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);
    float big = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    float r = ax < 0.4f ? small : big;
    return __builtin_copysignf(r, x);
  }
  float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// sin and cos in float32: 3-term Cody-Waite reduction by pi/2 (exact first step through the fma) and
// degree-3 minimax polynomials in r^2; abs. error < 1e-7 for |x| < 3000.  (ocml sincosf carries a
// Payne-Hanek path whose scratch array would put every wave's registers on the stack.)
DEV void sincos_dev(float x, float& sn, float& cs) {
  float k = __builtin_rintf(x * 0.6366197723675814f);
  float r = __builtin_fmaf(k, -1.5707963705062866f, x);
  r = __builtin_fmaf(k, 4.371138828673793e-08f, r);
  r = __builtin_fmaf(k, 1.7763568394002505e-15f, r);
  float r2 = r * r;
  float p = 2.723737452470232e-06f;
  p = __builtin_fmaf(p, r2, -0.00019839986634906381f);
  p = __builtin_fmaf(p, r2, 0.008333331905305386f);
  p = __builtin_fmaf(p, r2, -0.1666666716337204f);
  float s0 = __builtin_fmaf(r * r2, p, r);
  float q = -2.721424721130461e-07f;
  q = __builtin_fmaf(q, r2, 2.479966133250855e-05f);
  q = __builtin_fmaf(q, r2, -0.0013888884568586946f);
  q = __builtin_fmaf(q, r2, 0.0416666679084301f);
  float c0 = __builtin_fmaf(r2 * r2, q, __builtin_fmaf(-0.5f, r2, 1.0f));
  int n = (int)k;
  float ss = (n & 1) ? c0 : s0;
  float cc = (n & 1) ? s0 : c0;
  sn = (n & 2) ? -ss : ss;
  cs = ((n + 1) & 2) ? -cc : cc;
}

// The weight images in LDS never change after staging, so without this the compiler hoists their loads
// out of the time loop (and CSEs them across the evaluations of one step) into registers it does not
// have, and spills.  A compiler-only memory barrier at the start of each dynamics evaluation keeps
// the reads where they are consumed.
DEV void keep_lds_reads_local() { asm volatile("" ::: "memory"); }

template <int T>
DEV void tanh_act(Act<T>& a) {
  constexpr bool ACCURATE = T <= 4;  // hidden width <= 64
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) a.v[t][r] = tanh_dev<ACCURATE>(a.v[t][r]);
}

// H_net / R_net / G_net of the 128-wide models: 2 log2(e) is multiplied into the weights and bias that produce the
// pre-activation when the image is packed (kPreScaled), so tanh is exp2, add, rcp, fma -- one multiply less per value.
template <int T>
constexpr bool kPreScaled = T > 4;

template <int T>
DEV void tanh_act_pre(Act<T>& a) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (kPreScaled<T>) a.v[t][r] = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a.v[t][r]) + 1.0f), 1.0f);
      else a.v[t][r] = tanh_dev<true>(a.v[t][r]);
    }
}

// o[t] = vec[16t + 4q .. +3]  (vector in natural unit order in LDS: bias, output weights)
template <int T>
DEV void load_vec(Act<T>& o, const float* vec, Lane ln) {
  keep_lds_reads_local();
#pragma unroll
  for (int t = 0; t < T; ++t) o.v[t] = *reinterpret_cast<const f32x4*>(vec + 16 * t + 4 * ln.q);
}

template <int T>
DEV void zero_act(Act<T>& o) {
#pragma unroll
  for (int t = 0; t < T; ++t) o.v[t] = splat4(0.f);
}

// in(<=4) -> 16T units, one k-step.  Wf is the fragment image [T][64]: lane (i,q) holds W[16nt+i][q].
template <int T>
DEV void in_layer(Act<T>& o, const float* Wf, Lane ln, float xk) {
  keep_lds_reads_local();
#pragma unroll
  for (int nt = 0; nt < T; ++nt) o.v[nt] = mfma(Wf[nt * 64 + ln.lane], xk, o.v[nt]);
}

// o += W * in : W is a padded row-major image [16*TO][LD], LD = 16*TI + 4.  Rows on lanes: one ds_read_b128
// feeds 4 k-steps; G output tiles (independent accumulation chains) are in flight per group.
// (A hand-pipelined variant with sched_group_barrier, prefetching the next group's fragments, was measured:
// K1 -2.5 %, K2 +3.7 % (more spills), and the IGroupLP solver multiplied compile time by ~15: not kept.)
template <int TO, int TI>
DEV void sq_fwd(Act<TO>& o, const float* W, Lane ln, const Act<TI>& in) {
  constexpr int LD = 16 * TI + 4;
  keep_lds_reads_local();
  const float* base = W + ln.i * LD + 4 * ln.q;
  if (TO == 1) {
    // a single output tile would be one chain of 4*TI dependent MFMAs (40-cycle latency each): use two partial sums
    f32x4 e = o.v[0], d = splat4(0.f);
#pragma unroll
    for (int t = 0; t < TI; ++t) {
      f32x4 a = *reinterpret_cast<const f32x4*>(base + 16 * t);
      e = mfma(a[0], in.v[t][0], e);
      d = mfma(a[1], in.v[t][1], d);
      e = mfma(a[2], in.v[t][2], e);
      d = mfma(a[3], in.v[t][3], d);
    }
    o.v[0] = e + d;
    return;
  }
  constexpr int G = TO < 4 ? TO : 4;
#pragma unroll
  for (int t = 0; t < TI; ++t) {
#pragma unroll
    for (int n0 = 0; n0 < TO; n0 += G) {
      f32x4 a[G];
#pragma unroll
      for (int g = 0; g < G; ++g) a[g] = *reinterpret_cast<const f32x4*>(base + (n0 + g) * 16 * LD + 16 * t);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < G; ++g) o.v[n0 + g] = mfma(a[g][r], in.v[t][r], o.v[n0 + g]);
    }
  }
}

// o += W^T * in : same image [16*TI][LD], LD = 16*TO + 4 (TO = tiles of the columns of W).  Columns on
// lanes: one ds_read_b32 per MFMA (pairs fuse to ds_read2_b32), bank-conflict-free.
template <int TO, int TI>
DEV void sq_bwd(Act<TO>& o, const float* W, Lane ln, const Act<TI>& in) {
  constexpr int LD = 16 * TO + 4;
  keep_lds_reads_local();
  const float* base = W + 4 * ln.q * LD + ln.i;
#pragma unroll
  for (int t = 0; t < TI; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < TO; ++nt) o.v[nt] = mfma(base[(16 * t + r) * LD + 16 * nt], in.v[t][r], o.v[nt]);
}

// ------------------------------------------------------------------------------------------------
// bf16x3 products on the matrix pipe.  f32-input MFMA shares the vector ALUs with VALU; the
// bf16 MFMA v_mfma_f32_16x16x32_bf16 runs on the separate matrix pipe (19 cycles for 8x the FLOPs of a
// 32-cycle f32 MFMA, measured by tools/probe_bf16_split.hip) and co-executes with VALU.  Each f32 operand is
// split exactly into three bf16 pieces x = h + m + l (8+8+8 significant bits); the six products
// Wh*xh, Wh*xm, Wm*xh, Wm*xm, Wh*xl, Wl*xh, accumulated in f32 inside the MFMA smallest first, reproduce the
// f32 product to better than the f32 fma chain does (probe on hardware: max error / sum|w||x| 8.9e-8 vs 1.3e-7).
// Operand registers: for k-step s (32 units = accumulator tiles 2s and 2s+1) lane (i,q) packs its 8 values
// [tile 2s regs 0..3, tile 2s+1 regs 0..3] as the 8 bf16 of its B fragment, i.e. k-slot 8q+j <-> unit
// 32s + (j<4 ? 4q+j : 16+4q+j-4); the weight images are stored with the same permutation of their columns.
// ------------------------------------------------------------------------------------------------
constexpr int MM_F32 = 0, MM_BF16X3 = 1, MM_F16X2 = 2;  // how the hidden x hidden products are evaluated
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

DEV f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

DEV unsigned pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: round-to-nearest-even, a -> low half
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// Priority: low while a wave streams bf16 MFMAs, higher otherwise, so the SIMD partner's VALU instructions take
// the issue slots the MFMA stream leaves (arbitration is priority, then age).  Worth +2 % on the bench workload;
// tools/probe_coexec.hip shows why it cannot be more: on this part an MFMA stream of one wave (bf16 or f32, either
// shape) hides only ~20 % of its time under the partner wave's VALU work -- matrix and vector time nearly add.
DEV void matrix_phase_begin() {
#if !defined(PHNN_NO_SETPRIO) && !defined(PHNN_STATIC_PRIO)
  __builtin_amdgcn_s_setprio(0);
#endif
}
DEV void matrix_phase_end() {
#if !defined(PHNN_NO_SETPRIO) && !defined(PHNN_STATIC_PRIO)
  __builtin_amdgcn_s_setprio(2);
#endif
}
// experiment (PHNN_STATIC_PRIO): no per-phase flips; the second-dispatched half of the workgroup (waves 4-7, the
// arbitration losers by age) runs at priority 1 throughout.  Measured (round 3, same box, 3 x 100 steps): per-phase
// flips 29.26, static 29.35, no s_setprio at all 29.34 M/s -- all within 0.3 %; the flips stay (they were worth 2 % on
// round 1's kernels and cost nothing now)
DEV void static_prio(int wave) {
#ifdef PHNN_STATIC_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
}

template <int T>
struct Split3 {  // three bf16 pieces of an activation vector, as MFMA B fragments per 32-unit k-step
  bf16x8 h[T / 2], m[T / 2], l[T / 2];
};

template <int T>
DEV void split_act(const Act<T>& a, Split3<T>& o) {
#pragma unroll
  for (int s = 0; s < T / 2; ++s) {
    u32x4 H, M, Lo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float x0 = a.v[2 * s + (p >> 1)][2 * (p & 1)], x1 = a.v[2 * s + (p >> 1)][2 * (p & 1) + 1];
      unsigned hb = pk_bf16(x0, x1);
      float r0 = x0 - __builtin_bit_cast(float, hb << 16), r1 = x1 - __builtin_bit_cast(float, hb & 0xffff0000u);
      unsigned mb = pk_bf16(r0, r1);
      float t0 = r0 - __builtin_bit_cast(float, mb << 16), t1 = r1 - __builtin_bit_cast(float, mb & 0xffff0000u);
      H[p] = hb;
      M[p] = mb;
      Lo[p] = pk_bf16(t0, t1);
    }
    o.h[s] = __builtin_bit_cast(bf16x8, H);
    o.m[s] = __builtin_bit_cast(bf16x8, M);
    o.l[s] = __builtin_bit_cast(bf16x8, Lo);
  }
}

// bf16 weight image: 3 parts [HID rows][RS bf16], RS = HID + 16 (288 B rows at HID = 128: conflict-free
// ds_read_b128 row reads, 2-way ds_read_b64_tr_b16 transposed reads), columns in k-slot order.
template <int HID>
struct BfImg {
  static constexpr int RS = HID + 16;            // bf16 elements per row
  static constexpr int PART = HID * RS * 2;      // bytes per part
  static constexpr int FLOATS = 3 * PART / 4;    // size in floats
};

// the six products of one k-step for two output tiles, smallest terms first; the two tiles alternate so that
// dependent MFMAs are two issue slots apart
DEV void mfma6x2(f32x4& o0, f32x4& o1, const bf16x8 (&a)[2][3], bf16x8 xh, bf16x8 xm, bf16x8 xl) {
  o0 = mfma_bf(a[0][2], xh, o0);
  o1 = mfma_bf(a[1][2], xh, o1);
  o0 = mfma_bf(a[0][0], xl, o0);
  o1 = mfma_bf(a[1][0], xl, o1);
  o0 = mfma_bf(a[0][1], xm, o0);
  o1 = mfma_bf(a[1][1], xm, o1);
  o0 = mfma_bf(a[0][1], xh, o0);
  o1 = mfma_bf(a[1][1], xh, o1);
  o0 = mfma_bf(a[0][0], xm, o0);
  o1 = mfma_bf(a[1][0], xm, o1);
  o0 = mfma_bf(a[0][0], xh, o0);
  o1 = mfma_bf(a[1][0], xh, o1);
}

// o += W * in  (rows of W on lanes; one ds_read_b128 per part, tile and k-step).
// (Measured and not kept: k-step-outer order with explicitly double-buffered weight fragments behind
// sched_barrier -- the LDS latency was already covered by the partner wave: K1 -0.4 %, K2 +6 %.)
template <int T>
DEV void sq_fwd_bf(Act<T>& o, const float* Wimg, Lane ln, const Split3<T>& in) {
  using I = BfImg<16 * T>;
  keep_lds_reads_local();
  matrix_phase_begin();
  const char* base = reinterpret_cast<const char*>(Wimg) + ln.i * (I::RS * 2) + ln.q * 16;
#ifdef PHNN_PREFETCH_BF
  {  // fragments of group g + 1 requested before the MFMAs of group g (as sq_fwd_h)
    bf16x8 a[2][2][3];
    auto loadg = [&](int g, bf16x8 (&dst)[2][3]) {
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          dst[gg][p] = *reinterpret_cast<const bf16x8*>(base + p * I::PART + (n0 + gg) * 16 * (I::RS * 2) + s * 64);
    };
    constexpr int NG = (T / 2) * (T / 2);
    loadg(0, a[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) loadg(g + 1, a[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
      mfma6x2(o.v[n0], o.v[n0 + 1], a[g & 1], in.h[s], in.m[s], in.l[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
    matrix_phase_end();
    return;
  }
#endif
#pragma unroll
  for (int n0 = 0; n0 < T; n0 += 2) {
#pragma unroll
    for (int s = 0; s < T / 2; ++s) {
      bf16x8 a[2][3];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[g][p] = *reinterpret_cast<const bf16x8*>(base + p * I::PART + (n0 + g) * 16 * (I::RS * 2) + s * 64);
      mfma6x2(o.v[n0], o.v[n0 + 1], a, in.h[s], in.m[s], in.l[s]);
    }
  }
  matrix_phase_end();
}

// o += W^T * in from the SAME image: the A fragment of (output tile nt, k-step s) is 8 rows x 1 column per lane,
// fetched with two ds_read_b64_tr_b16 (each: 4 rows x 16 columns per 16-lane group, delivered column-major).
// Lane (q, 4a+pp) supplies the address of row 32s+4q+a (+16 for the second read), k-slot columns
// 32(nt>>1)+8pp+4(nt&1) .. +3; lane i of the group receives the column of unit 16nt+i.  EXEC is all ones here.
template <int T>
DEV void sq_bwd_bf(Act<T>& o, const float* Wimg, Lane ln, const Split3<T>& in) {
  using I = BfImg<16 * T>;
  keep_lds_reads_local();
  matrix_phase_begin();
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf4;
  typedef char __attribute__((address_space(3))) * lds_cp;
  const int a4 = (ln.lane & 15) >> 2, pp = ln.lane & 3;
  lds_cp base = (lds_cp) const_cast<char*>(reinterpret_cast<const char*>(Wimg)) + (4 * ln.q + a4) * (I::RS * 2) + 16 * pp;
#ifdef PHNN_PREFETCH_BF
  {
    bf16x8 a[2][2][3];
    auto loadg = [&](int g, bf16x8 (&dst)[2][3]) {
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int nt = n0 + gg;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          lds_cp off = base + p * I::PART + 32 * s * (I::RS * 2) + (64 * (nt >> 1) + 8 * (nt & 1));
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4)off);
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4)(off + 16 * (I::RS * 2)));
          dst[gg][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    };
    constexpr int NG = (T / 2) * (T / 2);
    loadg(0, a[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) loadg(g + 1, a[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
      mfma6x2(o.v[n0], o.v[n0 + 1], a[g & 1], in.h[s], in.m[s], in.l[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
    matrix_phase_end();
    return;
  }
#endif
#pragma unroll
  for (int n0 = 0; n0 < T; n0 += 2) {
#pragma unroll
    for (int s = 0; s < T / 2; ++s) {
      bf16x8 a[2][3];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int nt = n0 + g;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          lds_cp off = base + p * I::PART + 32 * s * (I::RS * 2) + (64 * (nt >> 1) + 8 * (nt & 1));
          bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4)off);
          bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4)(off + 16 * (I::RS * 2)));
          a[g][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
      mfma6x2(o.v[n0], o.v[n0 + 1], a, in.h[s], in.m[s], in.l[s]);
    }
  }
  matrix_phase_end();
}

// ------------------------------------------------------------------------------------------------
// f16x2 products: half the matrix instructions of bf16x3.  x = h + l with h = f16(x), l = f16(x - h) carries 22
// significant bits; the three products Wh*xh, Wh*xl, Wl*xh on v_mfma_f32_16x16x32_f16 give ~2^-22 relative
// accuracy per term provided the operands sit in f16's range: weights are stored times a power of two S (largest
// |w| S in [0.5,1)), tanh outputs and g = w3 (1 - a^2) are O(1) by construction, and the Hessian-vector product is
// linear in its input vector, which is normalised by a power of two per rollout (hnet_hvp).  All scale factors are
// powers of two and are folded into constants of the image (bias * S, tanh constant / S, W1^T / S, w3 / S).
// ------------------------------------------------------------------------------------------------
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

DEV f32x4 mfma_h(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// x - (float)h for a packed pair, one v_fma_mix_f32 each (f16 operand read straight from the packed register:
// replaces v_cvt_f32_f16 + v_sub_f32; the compiler does not form it from the plain expression)
DEV f32x2 residual_h(f16x2 h, f32x2 x) {
#ifdef PHNN_NO_FMAMIX
  return x - __builtin_convertvector(h, f32x2);
#endif
  unsigned hb = __builtin_bit_cast(unsigned, h);
  f32x2 r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hb), "v"(x[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hb), "v"(x[1]));
  return r;
}

template <int T>
struct Split2 {
  f16x8 h[T / 2], l[T / 2];
};

template <int T>
DEV void split_act_h(const Act<T>& a, Split2<T>& o) {
#pragma unroll
  for (int s = 0; s < T / 2; ++s) {
    u32x4 H, Lo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x2 x = {a.v[2 * s + (p >> 1)][2 * (p & 1)], a.v[2 * s + (p >> 1)][2 * (p & 1) + 1]};
      f16x2 hb = __builtin_convertvector(x, f16x2);
      f32x2 r = residual_h(hb, x);
      H[p] = __builtin_bit_cast(unsigned, hb);
      Lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
    o.h[s] = __builtin_bit_cast(f16x8, H);
    o.l[s] = __builtin_bit_cast(f16x8, Lo);
  }
}

template <int HID>
struct HfImg {  // 2 parts [HID rows][RS f16], same row stride and column permutation as BfImg
  static constexpr int RS = HID + 16;
  static constexpr int PART = HID * RS * 2;
  static constexpr int FLOATS = 2 * PART / 4;
};

// f16x2 input layer: in(<=4) -> 16T units with all four split products of W x in ONE v_mfma_f32_16x16x32_f16 per
// tile (8 issue cycles instead of the f32 MFMA's 32).  B fragment of every lane: [hi(x0..3), lo(x0..3)] of its rollout.
// A fragment image [T][64] x 8 halves: lane (i,0) holds [hi(W[u][0..3]), hi(W[u][0..3])], lane (i,1) the same of lo(W),
// lanes q >= 2 zeros -- k-slots 0..7 give W_hi (x_hi + x_lo), 8..15 W_lo (x_hi + x_lo).  f16 denormals are honoured by
// the matrix pipe (tools/probe_f16_denorm.hip), so small inputs keep an absolute error <= 2^-25.
DEV f16x8 in_frag_h(f32x4 x) {
  f32x2 v01 = {x[0], x[1]}, v23 = {x[2], x[3]};
  f16x2 h01 = __builtin_convertvector(v01, f16x2), h23 = __builtin_convertvector(v23, f16x2);
  f32x2 r01 = residual_h(h01, v01), r23 = residual_h(h23, v23);
  u32x4 B = {__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23),
             __builtin_bit_cast(unsigned, __builtin_convertvector(r01, f16x2)),
             __builtin_bit_cast(unsigned, __builtin_convertvector(r23, f16x2))};
  return __builtin_bit_cast(f16x8, B);
}
constexpr int kInFragH(int T) { return T * 64 * 4; }  // floats of the f16 fragment image
template <int T>
DEV void in_layer_h(Act<T>& o, const float* Wh, Lane ln, f16x8 xf) {
  keep_lds_reads_local();
#pragma unroll
  for (int nt = 0; nt < T; ++nt) o.v[nt] = mfma_h(*reinterpret_cast<const f16x8*>(Wh + (nt * 64 + ln.lane) * 4), xf, o.v[nt]);
}
// the input layer of a net in product mode MM: f32 fragment image at oF, f16 fragment image at oFh (f16x2 only).
// SITE: which call site; PHNN_INH_MASK selects the sites that use the f16 form.  Default: everywhere except the hidden
// layer of R_net / G_net inside the adjoint (kInHNet1Adj): with it the headline adjoint kernel spills 120 B per lane
// (measured: every site f16 = K1 -3.8 %, K2 +7 %; this mask = K1 -2.5 %, K2 -0.6 %).
constexpr int kInHFwd = 1, kInHRecomp = 2, kInHHvp = 4, kInHNet1 = 8, kInHNet1Adj = 16;
// (Round 3: with a1 recomputed late the adjoint has the registers for the f16 form at kInHNet1Adj too: K2 -0.9 %, no
// spill in the headline kernel; the mask is now every site.)
#ifndef PHNN_INH_MASK
#define PHNN_INH_MASK (kInHFwd | kInHRecomp | kInHHvp | kInHNet1 | kInHNet1Adj)
#endif
template <int T, int MM, int SITE>
DEV void in_layer_mm(Act<T>& o, const float* Lnet, int oF, int oFh, Lane ln, f32x4 x, int tile0 = 0) {
  if constexpr (MM == MM_F16X2 && (PHNN_INH_MASK & SITE) != 0) in_layer_h<T>(o, Lnet + oFh + tile0 * 256, ln, in_frag_h(x));
  else in_layer<T>(o, Lnet + oF + tile0 * 64, ln, sel4(x, ln.q));
}

DEV void mfma3x2(f32x4& o0, f32x4& o1, const f16x8 (&a)[2][2], f16x8 xh, f16x8 xl) {
  o0 = mfma_h(a[0][1], xh, o0);
  o1 = mfma_h(a[1][1], xh, o1);
  o0 = mfma_h(a[0][0], xl, o0);
  o1 = mfma_h(a[1][0], xl, o1);
  o0 = mfma_h(a[0][0], xh, o0);
  o1 = mfma_h(a[1][0], xh, o1);
}

template <int T>
DEV void sq_fwd_h(Act<T>& o, const float* Wimg, Lane ln, const Split2<T>& in) {
  using I = HfImg<16 * T>;
  keep_lds_reads_local();
  matrix_phase_begin();
  const char* base = reinterpret_cast<const char*>(Wimg) + ln.i * (I::RS * 2) + ln.q * 16;
#ifndef PHNN_NO_PREFETCH_FRAGS
  // The fragments of group g + 1 are requested before the MFMAs of group g (double buffer, pinned by sched_barrier: left
  // to itself the scheduler sinks every fragment load to just before its first use, and every six-MFMA group then waits
  // out a full LDS round trip).  Round 3, same box: K2 1.218 -> 1.185 ms at two waves per SIMD, 1.46 -> 1.36 ms at one;
  // K1 unchanged.  Two groups ahead costs K2 its registers (spills, +7 %).  Same MFMA order per accumulator: bitwise
  // the same results.
  {
    constexpr int PD = 1, NB = PD + 1;
    f16x8 a[NB][2][2];
    auto loadg = [&](int g, f16x8 (&dst)[2][2]) {
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          dst[gg][p] = *reinterpret_cast<const f16x8*>(base + p * I::PART + (n0 + gg) * 16 * (I::RS * 2) + s * 64);
    };
    constexpr int NG = (T / 2) * (T / 2);
#pragma unroll
    for (int g = 0; g < PD && g < NG; ++g) loadg(g, a[g % NB]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PD < NG) loadg(g + PD, a[(g + PD) % NB]);
      __builtin_amdgcn_sched_barrier(0);
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
      mfma3x2(o.v[n0], o.v[n0 + 1], a[g % NB], in.h[s], in.l[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
    matrix_phase_end();
    return;
  }
#endif
#pragma unroll
  for (int n0 = 0; n0 < T; n0 += 2) {
#pragma unroll
    for (int s = 0; s < T / 2; ++s) {
      f16x8 a[2][2];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          a[g][p] = *reinterpret_cast<const f16x8*>(base + p * I::PART + (n0 + g) * 16 * (I::RS * 2) + s * 64);
      mfma3x2(o.v[n0], o.v[n0 + 1], a, in.h[s], in.l[s]);
    }
  }
  matrix_phase_end();
}

template <int T>
DEV void sq_bwd_h(Act<T>& o, const float* Wimg, Lane ln, const Split2<T>& in) {
  using I = HfImg<16 * T>;
  keep_lds_reads_local();
  matrix_phase_begin();
  typedef fp16x4_t __attribute__((address_space(3))) * lds_h4;
  typedef char __attribute__((address_space(3))) * lds_cp;
  const int a4 = (ln.lane & 15) >> 2, pp = ln.lane & 3;
  lds_cp base = (lds_cp) const_cast<char*>(reinterpret_cast<const char*>(Wimg)) + (4 * ln.q + a4) * (I::RS * 2) + 16 * pp;
#ifndef PHNN_NO_PREFETCH_FRAGS
  {
    f16x8 a[2][2][2];
    auto loadg = [&](int g, f16x8 (&dst)[2][2]) {
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int nt = n0 + gg;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          lds_cp off = base + p * I::PART + 32 * s * (I::RS * 2) + (64 * (nt >> 1) + 8 * (nt & 1));
          f16x4 lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)off));
          f16x4 hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(off + 16 * (I::RS * 2))));
          dst[gg][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    };
    constexpr int NG = (T / 2) * (T / 2);
    constexpr int PD = 1, NB = PD + 1;
#pragma unroll
    for (int g = 0; g < PD && g < NG; ++g) loadg(g, a[g % NB]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + PD < NG) loadg(g + PD, a[(g + PD) % NB]);
      __builtin_amdgcn_sched_barrier(0);
      const int n0 = 2 * (g / (T / 2)), s = g % (T / 2);
      mfma3x2(o.v[n0], o.v[n0 + 1], a[g % NB], in.h[s], in.l[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
    matrix_phase_end();
    return;
  }
#endif
#pragma unroll
  for (int n0 = 0; n0 < T; n0 += 2) {
#pragma unroll
    for (int s = 0; s < T / 2; ++s) {
      f16x8 a[2][2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int nt = n0 + g;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          lds_cp off = base + p * I::PART + 32 * s * (I::RS * 2) + (64 * (nt >> 1) + 8 * (nt & 1));
          f16x4 lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)off));
          f16x4 hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(off + 16 * (I::RS * 2))));
          a[g][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
      mfma3x2(o.v[n0], o.v[n0 + 1], a, in.h[s], in.l[s]);
    }
  }
  matrix_phase_end();
}

// 16*TI units -> 4 outputs, every lane receives all 4.  Wt is [4][LR], LR = 16*TI + 8, row c = weights of output c.
// Uses the 16-block v_mfma_f32_4x4x1 (8 cycles): block b = lane>>2 = 4q + (i>>2) multiplies A_b[c][0] = Wt[c = i&3]
// [unit 16t+4q+r] with B_b[0][j = i&3] = the accumulator value of rollout i, i.e. one k per k-slot q and a group of
// four rollouts per block -- exactly the register contents we have.  Each lane then holds, in register c, output c of
// its rollout summed over ITS k-slot's units; the four k-slots are added with two cross-lane steps.
// Association (fixed, part of the bitwise contract between the whole-tile and the split-tile kernels): the units are
// summed in groups of two tiles (32 units: two interleaved chains o0, o1, P = o0 + o1), the group sums are added left
// to right, then the k-slots.  A tile split over four waves (wave w = group w) reproduces exactly this.
template <int TG>
DEV f32x4 to4_group(const float* Wt_group, Lane ln, const f32x4* in_tiles) {  // partial sum of TG tiles, before the k-slot sum
  const float* base = Wt_group + 4 * ln.q;
  f32x4 o0 = splat4(0.f), o1 = splat4(0.f);
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    f32x4 a = *reinterpret_cast<const f32x4*>(base + 16 * t);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if ((r & 1) == 0) o0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[r], in_tiles[t][r], o0, 0, 0, 0);
      else o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[r], in_tiles[t][r], o1, 0, 0, 0);
    }
  }
  return o0 + o1;
}

// Sums over the four lanes of a rollout (lane = 16 q + i): (v_q + v_{q^1}) first, then across the wave halves -- the
// association of v += xor16; v += xor32.  v_permlane16_swap / v_permlane32_swap exchange rows between TWO registers, so a
// copy of the value swapped against itself leaves the two partners in the pair; all in the vector ALU, no LDS round trip
// (ds_bpermute) and no lgkmcnt wait.  (The __builtin_amdgcn_permlane*_swap builtins of this compiler return the first
// result twice when both operands hold the same value -- tools/probe_permlane.hip -- hence the assembly.)
DEV f32x4 to4_kslots(f32x4 o) {  // sum over the four k-slots (lanes q = 0..3 of a rollout)
#ifdef PHNN_BPERMUTE_REDUCE
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = o[c];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    o[c] = v;
  }
  return o;
#else
  f32x4 b;
  asm volatile("v_mov_b32 %4, %0\n\tv_mov_b32 %5, %1\n\tv_mov_b32 %6, %2\n\tv_mov_b32 %7, %3\n\t"
               "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
               "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 0"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]));
  o = o + b;
  asm volatile("v_mov_b32 %4, %0\n\tv_mov_b32 %5, %1\n\tv_mov_b32 %6, %2\n\tv_mov_b32 %7, %3\n\t"
               "v_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
               "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\ts_nop 0"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]));
  return o + b;
#endif
}

template <int TI>
DEV f32x4 to4_rep(const float* Wt, Lane ln, const Act<TI>& in) {
  constexpr int LR = 16 * TI + 8;
  keep_lds_reads_local();
  const float* row = Wt + (ln.i & 3) * LR;
  constexpr int TG = TI >= 2 ? 2 : 1, NG = TI / TG;
  f32x4 tot = to4_group<TG>(row, ln, &in.v[0]);
#pragma unroll
  for (int g = 1; g < NG; ++g) tot = tot + to4_group<TG>(row + 16 * TG * g, ln, &in.v[TG * g]);
  return to4_kslots(tot);
}

// Per-wave stash in HBM (K1 -> K2): one activation vector = T x 64 lanes x float4, i.e. one fully coalesced
// 1 KB store/load per tile.  Streamed once each way, so non-temporal.
template <int T>
DEV void store_act(float* dst, Lane ln, const Act<T>& a) {
#pragma unroll
  for (int t = 0; t < T; ++t) PHNN_NT_STORE(a.v[t], reinterpret_cast<f32x4*>(dst) + t * 64 + ln.lane);
}
template <int T>
DEV void load_act(const float* src, Lane ln, Act<T>& a) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
#ifdef PHNN_PLAIN_STASH_LOADS
    a.v[t] = reinterpret_cast<const f32x4*>(src)[t * 64 + ln.lane];
#else
    a.v[t] = PHNN_NT_LOAD(reinterpret_cast<const f32x4*>(src) + t * 64 + ln.lane);
#endif
  }
}

// 24-bit fixed-point tape for vectors that are tanh OUTPUTS (|a| <= 1): q = rint(a (2^23 - 1)), four values in three
// dwords per lane, one coalesced 768 B store / load per tile.  Absolute error <= 6e-8 -- below the 2.5e-7 of the kernels'
// own tanh -- for 25 % fewer tape bytes.  Used where BOTH march kernels are bound by the tape's HBM stream (the ODEFunc
// model: K1 / K2 of BASELINE config 5 moved 54.5 GB each way at 3.6 TB/s with the matrix pipe mostly idle); the pHNN
// adjoint is bound by vector-instruction issue, where the 14 unpack instructions per float4 would cost more than the bytes.
constexpr float kFix24 = 8388607.0f;
// (a 3-vector type is padded to 16 bytes: the 12-byte lane stride is applied by hand, the access itself is one
// global_store / global_load_dwordx3)
typedef unsigned u32x3 __attribute__((ext_vector_type(3), aligned(4)));
template <int T>
DEV void store_act24(float* dst, Lane ln, const Act<T>& a) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
    unsigned q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = (unsigned)(int)__builtin_rintf(a.v[t][r] * kFix24);
    u32x3 d = {__builtin_amdgcn_perm(q[1], q[0], 0x04020100u),   // q0.b0 q0.b1 q0.b2 q1.b0
               __builtin_amdgcn_perm(q[2], q[1], 0x05040201u),   // q1.b1 q1.b2 q2.b0 q2.b1
               __builtin_amdgcn_perm(q[3], q[2], 0x06050402u)};  // q2.b2 q3.b0 q3.b1 q3.b2
    PHNN_NT_STORE(d, reinterpret_cast<u32x3*>(reinterpret_cast<char*>(dst) + (t * 64 + ln.lane) * 12));
  }
}
template <int T>
DEV void load_act24(const float* src, Lane ln, Act<T>& a) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const u32x3 d = PHNN_NT_LOAD(reinterpret_cast<const u32x3*>(reinterpret_cast<const char*>(src) + (t * 64 + ln.lane) * 12));
    const int q0 = (int)(d[0] << 8) >> 8;
    const int q1 = (int)(__builtin_amdgcn_alignbit(d[1], d[0], 24) << 8) >> 8;
    const int q2 = (int)(__builtin_amdgcn_alignbit(d[2], d[1], 16) << 8) >> 8;
    const int q3 = (int)d[2] >> 8;
    a.v[t] = f32x4{(float)q0, (float)q1, (float)q2, (float)q3} * (1.0f / kFix24);
  }
}

// 16 per-rollout values spread over the 4 lanes of a rollout (lane (i,q) holds values 4q..4q+3) ->
// all 16 in every lane, through the wave's LDS scratch (in-order per wave, no barrier needed).
DEV void gather16(float* scr, Lane ln, f32x4 mine, float (&out)[16]) {
  float* row = scr + ln.i * 20;
  *reinterpret_cast<f32x4*>(row + 4 * ln.q) = mine;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * k);
    out[4 * k + 0] = v[0];
    out[4 * k + 1] = v[1];
    out[4 * k + 2] = v[2];
    out[4 * k + 3] = v[3];
  }
  __builtin_amdgcn_wave_barrier();
}

DEV float reduce_q(float v) {  // sum over the 4 lanes (q = 0..3) of a rollout
#ifdef PHNN_BPERMUTE_REDUCE
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
#else
  float b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(v), "=&v"(b));
  v += b;
  asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(v), "=&v"(b));
  return v + b;
#endif
}

// ------------------------------------------------------------------------------------------------
// LDS image layouts (offsets in floats; every section size is a multiple of 4 floats)
// ------------------------------------------------------------------------------------------------

template <int HID, int MM = MM_F32>
struct LayH2 {  // in(<=4) -> HID -> HID -> 1  (H_net)
  static constexpr int T = HID / 16, LD = HID + 4, LR = HID + 8;
  static constexpr int oW2 = 0;                  // f32: [HID][LD]; bf16x3 / f16x2: BfImg / HfImg (x S for f16x2)
  static constexpr int W2F = MM == MM_F32 ? HID * LD : (MM == MM_BF16X3 ? BfImg<HID>::FLOATS : HfImg<HID>::FLOATS);
  static constexpr int oW1h = oW2 + W2F;         // f16x2: [T][64] x 8 halves, f16 fragment image of W1 (in_layer_h)
  static constexpr int oW1f = oW1h + (MM == MM_F16X2 ? T * 64 * 4 : 0);  // [T][64] f32 fragment image of W1
  static constexpr int oB1 = oW1f + T * 64;      // [HID]
  static constexpr int oB2 = oB1 + HID;          // [HID]   b2 * S
  static constexpr int oW3 = oB2 + HID;          // [HID]   w3 (Hamiltonian value)
  static constexpr int oW3B = oW3 + HID;         // [HID]   w3 * Sb (g2 = w3 (1 - a2^2), fed to the transposed product)
  static constexpr int oW3S = oW3B + HID;        // [HID]   w3 * Sb / S (used by the Hessian-vector product)
  static constexpr int oW1T = oW3S + HID;        // [4][LR] rows c = W1[:,c] / (S Sb)
  static constexpr int oB3 = oW1T + 4 * LR;      // [4] (b3, 2 log2(e) / S, 1 / k1, Sb), k1 = factor folded into W1, b1
  static constexpr int SIZE = oB3 + 4;
};

template <int HID, int MM = MM_F32>
struct LayH1 {  // in(<=4) -> HID -> out(<=16)  (R_net, G_net)
  static constexpr int T = HID / 16, LD = HID + 4, LR = HID + 8;
  static constexpr bool HF = MM == MM_F16X2;    // output layer as f16x2 products (h1_fwd / h1_bwd)
  static constexpr int RS = HID + 16;           // f16 per row of the forward image (same stride as HfImg)
  static constexpr int FPART = 16 * RS * 2;     // bytes per part of the forward image  [16 outputs][RS], k-slot order
  static constexpr int BPART = HID * 16 * 2;    // bytes per part of the transposed image [HID units][16 outputs]
  static constexpr int oV2 = 0;                 // f32: [16][LD];  f16x2: forward image (hi, lo) of Sr * V2 ...
  static constexpr int oV2T = oV2 + 2 * FPART / 4;  // ... then the transposed image (hi, lo) of Sr * V2^T
  static constexpr int V2F = HF ? (2 * FPART + 2 * BPART) / 4 : 16 * LD;
  static constexpr int oV1h = oV2 + V2F;        // f16x2: f16 fragment image of V1 (in_layer_h)
  static constexpr int oV1f = oV1h + (HF ? T * 64 * 4 : 0);  // [T][64]
  static constexpr int oC1 = oV1f + T * 64;     // [HID]
  static constexpr int oC2 = oC1 + HID;         // [16]
  static constexpr int oSc = oC2 + 16;          // [4] (1 / Sr, 0, 0, 0)
  static constexpr int oV1T = oSc + 4;          // [4][LR]
  static constexpr int SIZE = oV1T + 4 * LR;
};

// ------------------------------------------------------------------------------------------------
// Weight-gradient record (training side).  The parameter gradient of lam^T f(x,u;theta) needs, per
// evaluation point, the tapes the VJP already holds in registers.  The adjoint kernels (template flag WG) stream them
// to HBM as one record per (16-point tile, evaluation); k_wgrad_reduce turns records into gradient sums as GEMMs over
// the points.  Record of a model with hidden width 16 T, in floats:
//   [v * T*256, (v+1) * T*256)   big vector v in accumulator layout ([tile t][lane] float4, one coalesced 1 KB store per tile):
//        0 a2            second hidden activation of H_net
//        1 q1 raw        W2^T g2 as the kernel holds it   (= S Sb q1)
//        2 ad2 raw       (1 - a2^2) W2 adot1, un-normalised (= S k1 adot2)
//        3 qd raw        W2^T of the -gdot2/2 the kernel forms, un-normalised (= -S Sb k1 qd / 2)
//        4 hbR           R_net: (V2^T rbar) (1 - h^2)      (pHNN only)
//        5 hbG           G_net: (V2^T gbar) (1 - h^2)      (learned G only)
//   then 16 rollouts x kRecSmall floats: x[4] (H_net input), v[4], lam[4], dH[4], rbar[16], u[4], Hbar, pad[3]
// (S, Sb, k1: power-of-two / tanh-constant scales folded into the image, LayH2::oB3.)
// ------------------------------------------------------------------------------------------------
constexpr int kRecSmall = 40;
template <int T, int NBIG>
struct WRec {
  static constexpr int VEC = T * 256;
  static constexpr int oSmall = NBIG * VEC;
  static constexpr int SIZE = oSmall + 16 * kRecSmall;
};

// The record stream is written once by the adjoint march and read once by k_wgrad_reduce (3.8 GB at the bench shape):
// non-temporal stores and loads.  Round 3, B = 65 536, H = 50: the record-writing adjoint 1.71 -> 1.55 ms, the reduction
// 2.90 -> 2.95 ms, the training pass 5.42 -> 5.31 ms.  (PHNN_CACHED_RECORDS: plain accesses.)
#ifdef PHNN_CACHED_RECORDS
#define PHNN_REC_STORE(v, p) (*(p) = (v))
#define PHNN_REC_LOAD(p) (*(p))
#else
#define PHNN_REC_STORE(v, p) __builtin_nontemporal_store((v), (p))
#define PHNN_REC_LOAD(p) __builtin_nontemporal_load((p))
#endif
template <int T>
DEV void store_rec(float* dst, Lane ln, const Act<T>& a) {
#pragma unroll
  for (int t = 0; t < T; ++t) PHNN_REC_STORE(a.v[t], reinterpret_cast<f32x4*>(dst) + t * 64 + ln.lane);
}
template <int T>
DEV void store_rec_scaled(float* dst, Lane ln, const Act<T>& a, float s) {
#pragma unroll
  for (int t = 0; t < T; ++t) PHNN_REC_STORE(a.v[t] * s, reinterpret_cast<f32x4*>(dst) + t * 64 + ln.lane);
}

// ------------------------------------------------------------------------------------------------
// Activations other than Tanh (src/NN.py:13 defaults to nn.SiLU; src/pHNN.py:41 resolves any nn.* by name;
// src/baseline_node.py:49-58 offers relu, elu, gelu): SiLU, ReLU, ELU and GELU on the all-f32 kernels.  Their outputs are unbounded, so the
// f16 / bf16 split products (which rely on |tanh| <= 1) do not apply, and phi' / phi'' need the PRE-activation: the
// tapes of these variants keep z where the Tanh kernels keep a = tanh(z).  Values = phnn_activation.
// ------------------------------------------------------------------------------------------------
constexpr int ACT_TANH = 0, ACT_SILU = 2, ACT_RELU = 3, ACT_ELU = 4, ACT_GELU = 5;

// standard normal density and distribution (nn.GELU, approximate='none': z Phi(z))
DEV float gelu_pdf(float z) { return 0.3989422804014327f * __builtin_amdgcn_exp2f(z * z * -0.7213475204444817f); }
DEV float gelu_cdf(float z) { return __builtin_fmaf(0.5f, erff(z * 0.7071067811865476f), 0.5f); }

template <int ACT>
DEV void act_eval(float z, float& a, float& d1) {  // phi(z), phi'(z)
  if (ACT == ACT_RELU) {
    a = fmaxf(z, 0.f);
    d1 = z > 0.f ? 1.0f : 0.0f;  // torch: subgradient 0 at 0
  } else if (ACT == ACT_ELU) {  // nn.ELU, alpha = 1: z | expm1(z);  1 | exp(z)
    a = z > 0.f ? z : expm1f(z);
    d1 = z > 0.f ? 1.0f : __builtin_amdgcn_exp2f(z * 1.4426950408889634f);
  } else if (ACT == ACT_GELU) {  // z Phi(z);  Phi(z) + z pdf(z)
    const float c = gelu_cdf(z);
    a = z * c;
    d1 = __builtin_fmaf(z, gelu_pdf(z), c);
  } else {  // SiLU: z s, s (1 + z (1 - s)) with s = sigmoid(z)
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
    a = z * s;
    d1 = __builtin_fmaf(a, 1.0f - s, s);
  }
}
template <int ACT>
DEV float act_d2(float z) {  // phi''(z)
  if (ACT == ACT_RELU) return 0.0f;
  if (ACT == ACT_ELU) return z > 0.f ? 0.0f : __builtin_amdgcn_exp2f(z * 1.4426950408889634f);
  if (ACT == ACT_GELU) return gelu_pdf(z) * __builtin_fmaf(-z, z, 2.0f);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
  return s * (1.0f - s) * __builtin_fmaf(z, 1.0f - 2.0f * s, 2.0f);
}
template <int ACT, int T>
DEV void act_of(const Act<T>& z, Act<T>& a) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float av, d;
      act_eval<ACT>(z.v[t][r], av, d);
      a.v[t][r] = av;
    }
}
template <int ACT, int T>
DEV void mul_d1(Act<T>& g, const Act<T>& z) {  // g *= phi'(z)
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a, d;
      act_eval<ACT>(z.v[t][r], a, d);
      g.v[t][r] *= d;
    }
}

// ------------------------------------------------------------------------------------------------
// H_net: value, gradient and Hessian-vector product (src/pHNN.py:72-73, src/pHNN_canonical.py:208-215)
// ------------------------------------------------------------------------------------------------
template <int HID>
struct HTape {
  Act<HID / 16> a1, a2, q1;  // activations (SiLU / ReLU variants: PRE-activations z1, z2), and q1 = W2^T g2
};

// generic-activation H_net on the all-f32 image (packed WITHOUT the folded tanh constant: S = Sb = k1 = 1)
template <int HID, bool WANT_H, int ACT>
DEV f32x4 hnet_grad_g(const float* L, Lane ln, f32x4 x, HTape<HID>& tp, float& Hval) {
  using Y = LayH2<HID, MM_F32>;
  constexpr int T = Y::T;
  load_vec<T>(tp.a1, L + Y::oB1, ln);
  in_layer<T>(tp.a1, L + Y::oW1f, ln, sel4(x, ln.q));  // z1
  {
    Act<T> a;
    act_of<ACT, T>(tp.a1, a);
    load_vec<T>(tp.a2, L + Y::oB2, ln);
    sq_fwd<T, T>(tp.a2, L + Y::oW2, ln, a);  // z2
  }
  Act<T> g;
  float s = 0.f;
  keep_lds_reads_local();
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f32x4 w3 = *reinterpret_cast<const f32x4*>(L + Y::oW3 + 16 * t + 4 * ln.q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a, d;
      act_eval<ACT>(tp.a2.v[t][r], a, d);
      if (WANT_H) s = __builtin_fmaf(w3[r], a, s);
      g.v[t][r] = w3[r] * d;  // g2 = w3 phi'(z2)
    }
  }
  if (WANT_H) Hval = reduce_q(s) + L[Y::oB3];
  zero_act<T>(tp.q1);
  sq_bwd<T, T>(tp.q1, L + Y::oW2, ln, g);
#pragma unroll
  for (int t = 0; t < T; ++t) g.v[t] = tp.q1.v[t];
  mul_d1<ACT, T>(g, tp.a1);  // g1 = q1 phi'(z1)
  return to4_rep<T>(L + Y::oW1T, ln, g);
}
template <int HID, int ACT>
DEV void hnet_layer1_g(const float* L, Lane ln, f32x4 x, Act<HID / 16>& z1) {
  using Y = LayH2<HID, MM_F32>;
  load_vec<Y::T>(z1, L + Y::oB1, ln);
  in_layer<Y::T>(z1, L + Y::oW1f, ln, sel4(x, ln.q));
}
// Hv = (d^2 H / dx^2) v:  zd1 = W1 v, ad1 = phi'(z1) zd1, zd2 = W2 ad1, gd2 = w3 phi''(z2) zd2, qd1 = W2^T gd2,
// gd1 = qd1 phi'(z1) + q1 phi''(z1) zd1, Hv = W1^T gd1   (oracle/phnn_oracle.c: hnet_hvp).  Consumes tp.q1.
template <int HID, int ACT>
DEV f32x4 hnet_hvp_g(const float* L, Lane ln, HTape<HID>& tp, f32x4 v) {
  using Y = LayH2<HID, MM_F32>;
  constexpr int T = Y::T;
  Act<T> ad1, w;
  zero_act<T>(ad1);
  in_layer<T>(ad1, L + Y::oW1f, ln, sel4(v, ln.q));  // zd1
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = tp.a1.v[t][r], zd = ad1.v[t][r];
      float a, d;
      act_eval<ACT>(z, a, d);
      tp.q1.v[t][r] = tp.q1.v[t][r] * (act_d2<ACT>(z) * zd);  // second term of gd1
      ad1.v[t][r] = d * zd;
    }
  zero_act<T>(w);
  sq_fwd<T, T>(w, L + Y::oW2, ln, ad1);  // zd2
  keep_lds_reads_local();
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f32x4 w3 = *reinterpret_cast<const f32x4*>(L + Y::oW3 + 16 * t + 4 * ln.q);
#pragma unroll
    for (int r = 0; r < 4; ++r) w.v[t][r] = w3[r] * (act_d2<ACT>(tp.a2.v[t][r]) * w.v[t][r]);  // gd2
  }
  Act<T> qd;
  zero_act<T>(qd);
  sq_bwd<T, T>(qd, L + Y::oW2, ln, w);
  mul_d1<ACT, T>(qd, tp.a1);
#pragma unroll
  for (int t = 0; t < T; ++t) qd.v[t] = qd.v[t] + tp.q1.v[t];
  return to4_rep<T>(L + Y::oW1T, ln, qd);
}

// (Measured, round 3, not kept: streaming a2 out right behind its tanh, so that its stores drain underneath the transposed
// product instead of queueing with q1's: K1 +0.8 %, K2 unchanged.)
template <int HID, bool WANT_H, int MM = MM_F32, int SITE = kInHFwd, int ACT = ACT_TANH>
DEV f32x4 hnet_grad(const float* L, Lane ln, f32x4 z, HTape<HID>& tp, float& Hval) {
  if constexpr (ACT != ACT_TANH) {
    static_assert(MM == MM_F32, "SiLU / ReLU run on the all-f32 kernels");
    return hnet_grad_g<HID, WANT_H, ACT>(L, ln, z, tp, Hval);
  }
  using Y = LayH2<HID, MM>;
  constexpr int T = Y::T;
  load_vec<T>(tp.a1, L + Y::oB1, ln);
  in_layer_mm<T, MM, SITE>(tp.a1, L, Y::oW1f, Y::oW1h, ln, z);
  tanh_act_pre<T>(tp.a1);
  load_vec<T>(tp.a2, L + Y::oB2, ln);
  if (MM == MM_BF16X3) {
    Split3<T> sp;
    split_act<T>(tp.a1, sp);
    sq_fwd_bf<T>(tp.a2, L + Y::oW2, ln, sp);
  } else if (MM == MM_F16X2) {
    Split2<T> sp;
    split_act_h<T>(tp.a1, sp);
    sq_fwd_h<T>(tp.a2, L + Y::oW2, ln, sp);
  } else {
    sq_fwd<T, T>(tp.a2, L + Y::oW2, ln, tp.a1);
  }
  if (MM == MM_F16X2 || kPreScaled<T>) {
    // the accumulator holds S * z2; c = 2 log2(e) / S.  128-wide: 2 log2(e) is folded into the image and the f16
    // image needs no further power of two for ordinary weights (max |2.89 w| in [0.5, 1024)), so c == 1 and the
    // multiply disappears; the branch is uniform.
    const float c = L[Y::oB3 + 1];
    if (kPreScaled<T> && c == 1.0f) {
      tanh_act_pre<T>(tp.a2);
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) tp.a2.v[t][r] = tanh_scaled(tp.a2.v[t][r], c);
    }
  } else {
    tanh_act<T>(tp.a2);
  }
  Act<T> g;
  float s = 0.f;
  keep_lds_reads_local();
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (WANT_H) {
      f32x4 w3 = *reinterpret_cast<const f32x4*>(L + Y::oW3 + 16 * t + 4 * ln.q);
#pragma unroll
      for (int r = 0; r < 4; ++r) s = __builtin_fmaf(w3[r], tp.a2.v[t][r], s);
    }
    f32x4 w3b = *reinterpret_cast<const f32x4*>(L + Y::oW3B + 16 * t + 4 * ln.q);
    g.v[t] = w3b * dtanh(tp.a2.v[t]);
  }
  if (WANT_H) Hval = reduce_q(s) + L[Y::oB3];
  zero_act<T>(tp.q1);
  if (MM == MM_BF16X3) {
    Split3<T> sp;
    split_act<T>(g, sp);
    sq_bwd_bf<T>(tp.q1, L + Y::oW2, ln, sp);
  } else if (MM == MM_F16X2) {  // q1 comes out times S; W1^T in the image is divided by S
    Split2<T> sp;
    split_act_h<T>(g, sp);
    sq_bwd_h<T>(tp.q1, L + Y::oW2, ln, sp);
  } else {
    sq_bwd<T, T>(tp.q1, L + Y::oW2, ln, g);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) g.v[t] = tp.q1.v[t] * dtanh(tp.a1.v[t]);
  return to4_rep<T>(L + Y::oW1T, ln, g);
}

// first hidden activation of H_net from its input: cheap (one k-step, 8 f32 MFMAs + tanh) -- K2 recomputes it
// instead of reading it from the stash, which cuts the stash traffic by a third
template <int HID, int MM, int ACT = ACT_TANH>
DEV void hnet_layer1(const float* L, Lane ln, f32x4 z, Act<HID / 16>& a1) {
  if constexpr (ACT != ACT_TANH) return hnet_layer1_g<HID, ACT>(L, ln, z, a1);
  using Y = LayH2<HID, MM>;
  load_vec<Y::T>(a1, L + Y::oB1, ln);
  in_layer_mm<Y::T, MM, kInHRecomp>(a1, L, Y::oW1f, Y::oW1h, ln, z);
  tanh_act_pre<Y::T>(a1);
}

// Hv = (d^2 H / dz^2) v : forward-over-reverse through the kept tape (a1, a2, q1).  Consumes the tape
// (q1 is overwritten) to keep the live register set at five activation vectors.
template <int HID, int MM = MM_F32, bool WG = false, int ACT = ACT_TANH>
DEV f32x4 hnet_hvp(const float* L, Lane ln, HTape<HID>& tp, f32x4 v, float* rec = nullptr) {
  if constexpr (ACT != ACT_TANH) {
    static_assert(!WG, "weight-gradient kernels exist for Tanh models only");
    return hnet_hvp_g<HID, ACT>(L, ln, tp, v);
  }
  using Y = LayH2<HID, MM>;
  constexpr int T = Y::T;
  float unscale = 1.0f;
  if (MM == MM_F16X2) {  // Hv is linear in v: bring max|v_i| into [0.5,1) by a power of two (exact), undo at the end
    float mx = fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    int e = 0;
    (void)__builtin_frexpf(mx, &e);
    e = (mx > 0.f && mx < 3.0e38f) ? e : 0;
    v = v * __builtin_ldexpf(1.0f, -e);
    unscale = __builtin_ldexpf(1.0f, e);
  }
  Act<T> ad1, w;
  zero_act<T>(ad1);
  in_layer_mm<T, MM, kInHHvp>(ad1, L, Y::oW1f, Y::oW1h, ln, v);
#pragma unroll
  for (int t = 0; t < T; ++t) ad1.v[t] = dtanh(tp.a1.v[t]) * ad1.v[t];
  // gdot1 = qdot1*(1-a1^2) + q1*(-2 a1 adot1) and gdot2 = w3 (-2 a2 (1-a2^2) zdot2): both carry a factor -2, so the
  // rest of this function works with -1/2 of the true quantities and the factor is restored on the final 4-vector.
  // Second term of gdot1 folded now, adot1 dies after the product.
#pragma unroll
  for (int t = 0; t < T; ++t) tp.q1.v[t] = tp.q1.v[t] * (tp.a1.v[t] * ad1.v[t]);
  zero_act<T>(w);
  if (MM == MM_BF16X3) {
    Split3<T> sp;
    split_act<T>(ad1, sp);
    sq_fwd_bf<T>(w, L + Y::oW2, ln, sp);
  } else if (MM == MM_F16X2) {
    Split2<T> sp;
    split_act_h<T>(ad1, sp);
    sq_fwd_h<T>(w, L + Y::oW2, ln, sp);
  } else {
    sq_fwd<T, T>(w, L + Y::oW2, ln, ad1);
  }
  keep_lds_reads_local();
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f32x4 w3 = *reinterpret_cast<const f32x4*>(L + Y::oW3S + 16 * t + 4 * ln.q);  // w3 / S: w holds S * zdot2
    f32x4 a2 = tp.a2.v[t];
    f32x4 ad2 = dtanh(a2) * w.v[t];
    if (WG) PHNN_REC_STORE(ad2 * unscale, reinterpret_cast<f32x4*>(rec + 2 * T * 256) + t * 64 + ln.lane);
    w.v[t] = w3 * (a2 * ad2);  // -gdot2 / 2
  }
  Act<T> qd;
  zero_act<T>(qd);
  if (MM == MM_BF16X3) {
    Split3<T> sp;
    split_act<T>(w, sp);
    sq_bwd_bf<T>(qd, L + Y::oW2, ln, sp);
  } else if (MM == MM_F16X2) {
    Split2<T> sp;
    split_act_h<T>(w, sp);
    sq_bwd_h<T>(qd, L + Y::oW2, ln, sp);
  } else {
    sq_bwd<T, T>(qd, L + Y::oW2, ln, w);
  }
  if (WG) store_rec_scaled<T>(rec + 3 * T * 256, ln, qd, unscale);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    qd.v[t] = __builtin_elementwise_fma(qd.v[t], dtanh(tp.a1.v[t]), tp.q1.v[t]);
  }
  f32x4 Hv = to4_rep<T>(L + Y::oW1T, ln, qd);
  return Hv * (-2.0f * unscale * L[Y::oB3 + 2]);  // oB3[2]: 1 / (scale folded into W1)
}

// HID -> 16 output layer of R_net / G_net on the hi/lo fragments of the hidden activations (f16x2): lane (i,q) gets outputs 4q..4q+3
template <int HID, int MM>
DEV f32x4 h1_out_hf(const float* L, Lane ln, const Split2<HID / 16>& sp) {
  using Y = LayH1<HID, MM>;
  constexpr int T = Y::T;
  keep_lds_reads_local();
  const char* base = reinterpret_cast<const char*>(L + Y::oV2) + ln.i * (Y::RS * 2) + ln.q * 16;
  f32x4 o0 = splat4(0.f), o1 = splat4(0.f), o2 = splat4(0.f);
#pragma unroll
  for (int s = 0; s < T / 2; ++s) {
    f16x8 ah = *reinterpret_cast<const f16x8*>(base + s * 64);
    f16x8 al = *reinterpret_cast<const f16x8*>(base + Y::FPART + s * 64);
    o0 = mfma_h(al, sp.h[s], o0);
    o1 = mfma_h(ah, sp.l[s], o1);
    o2 = mfma_h(ah, sp.h[s], o2);
  }
  const float inv = L[Y::oSc];
  f32x4 c2 = *reinterpret_cast<const f32x4*>(L + Y::oC2 + 4 * ln.q);
  return ((o0 + o1) + o2) * inv + c2;
}

// B operands of the transposed output layer (f16x2): obar normalised per rollout by a power of two, split hi/lo and
// stacked along K (see h1_bwd).  unscale = 2^e / Sr.
template <int HID, int MM>
DEV void h1_bwd_operands(const float* L, Lane ln, const float (&obar)[16], f16x8& b1, f16x8& b2, float& unscale) {
  using Y = LayH1<HID, MM>;
  float mx = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) mx = fmaxf(mx, __builtin_fabsf(obar[k]));
  int e = 0;
  (void)__builtin_frexpf(mx, &e);
  e = (mx > 0.f && mx < 3.0e38f) ? e : 0;
  const float sc = __builtin_ldexpf(1.0f, -e);
  unscale = __builtin_ldexpf(1.0f, e) * L[Y::oSc];
  const bool second = (ln.q & 1) != 0, lo_half = ln.q >= 2;
  u32x4 B1, B2;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 v = {(second ? obar[8 + 2 * p] : obar[2 * p]) * sc, (second ? obar[9 + 2 * p] : obar[2 * p + 1]) * sc};
    f16x2 hb2 = __builtin_convertvector(v, f16x2);
    f32x2 r = residual_h(hb2, v);
    unsigned hbits = __builtin_bit_cast(unsigned, hb2), lbits = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    B1[p] = lo_half ? lbits : hbits;
    B2[p] = lo_half ? 0u : hbits;
  }
  b1 = __builtin_bit_cast(f16x8, B1);
  b2 = __builtin_bit_cast(f16x8, B2);
}

// one-hidden-layer net in(<=4) -> HID -> out(<=16): forward keeps the hidden activations.
// MM_F16X2: the HID -> 16 output layer runs as 3 x T/2 v_mfma_f32_16x16x32_f16 (three independent chains) on the
// hi/lo split of the hidden activations instead of 2T dependent-pair f32 MFMAs of 32 cycles each.
// the 16 outputs of R_net per rollout in the K1 -> K2 tape: lane (i,q) keeps out[4q .. 4q+3] of rollout i (every lane
// holds all 16 after gather16), one coalesced 1 KB store / load per wave
DEV void store_rf(float* dst, Lane ln, f32x4 o) {  // o = outputs 4q .. 4q+3 of rollout i, as the output layer leaves them
  if (ln.w == 0) PHNN_NT_STORE(o, reinterpret_cast<f32x4*>(dst) + ln.i * 4 + ln.q);
}
DEV void load_rf(const float* src, Lane ln, float (&rf)[16]) {
  const f32x4* p = reinterpret_cast<const f32x4*>(src) + ln.i * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x4 v = PHNN_NT_LOAD(p + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) rf[4 * k + e] = v[e];
  }
}
// hidden layer only (the adjoint with a tape: the outputs come from there)
template <int HID, int MM = MM_F32, int SITE = kInHNet1, int ACT = ACT_TANH>
DEV void h1_hidden(const float* L, Lane ln, f32x4 x, Act<HID / 16>& h) {
  using Y = LayH1<HID, MM>;
  constexpr int T = Y::T;
  load_vec<T>(h, L + Y::oC1, ln);
  in_layer_mm<T, MM, SITE>(h, L, Y::oV1f, Y::oV1h, ln, x);
  if constexpr (ACT == ACT_TANH) tanh_act_pre<T>(h);  // SiLU / ReLU: h keeps the PRE-activation (h1_bwd needs phi'(z))
}

template <int HID, int MM = MM_F32, int SITE = kInHNet1, int ACT = ACT_TANH>
DEV void h1_fwd(const float* L, float* scr, Lane ln, f32x4 x, Act<HID / 16>& h, float (&out)[16], float* rf_stash = nullptr) {
  using Y = LayH1<HID, MM>;
  constexpr int T = Y::T;
  load_vec<T>(h, L + Y::oC1, ln);
  in_layer_mm<T, MM, SITE>(h, L, Y::oV1f, Y::oV1h, ln, x);
  if constexpr (ACT != ACT_TANH) {  // h stays z; the output layer takes phi(z)
    static_assert(MM == MM_F32, "SiLU / ReLU run on the all-f32 kernels");
    Act<T> a;
    act_of<ACT, T>(h, a);
    Act<1> og;
    og.v[0] = *reinterpret_cast<const f32x4*>(L + Y::oC2 + 4 * ln.q);
    sq_fwd<1, T>(og, L + Y::oV2, ln, a);
    if (rf_stash) store_rf(rf_stash, ln, og.v[0]);
    gather16(scr, ln, og.v[0], out);
    return;
  }
  tanh_act_pre<T>(h);
  Act<1> o;
  if (Y::HF) {
    Split2<T> sp;
    split_act_h<T>(h, sp);
    o.v[0] = h1_out_hf<HID, MM>(L, ln, sp);
  } else {
    o.v[0] = *reinterpret_cast<const f32x4*>(L + Y::oC2 + 4 * ln.q);
    sq_fwd<1, T>(o, L + Y::oV2, ln, h);
  }
  if (rf_stash) store_rf(rf_stash, ln, o.v[0]);
  gather16(scr, ln, o.v[0], out);
}

// xbar = (d net / d x)^T obar, obar given as all 16 values in every lane.
// MM_F16X2: obar is normalised per rollout by a power of two (the map is linear), split hi/lo and STACKED along the
// K = 32 of one MFMA: k-slots 0..15 carry hi(obar), 16..31 lo(obar), against [V2^T hi | V2^T hi]; a second MFMA adds
// V2^T lo x hi(obar).  Two 16-cycle MFMAs per tile of hidden units instead of four 32-cycle f32 ones.
template <int HID, int MM = MM_F32, bool WG = false, int ACT = ACT_TANH>
DEV f32x4 h1_bwd(const float* L, Lane ln, const Act<HID / 16>& h, const float (&obar)[16], float* rec = nullptr) {
  using Y = LayH1<HID, MM>;
  constexpr int T = Y::T;
  Act<T> hb;
  if constexpr (ACT != ACT_TANH) {  // h holds the pre-activation
    static_assert(MM == MM_F32 && !WG, "SiLU / ReLU: all-f32 kernels, no weight-gradient records");
    Act<1> ob;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      ob.v[0][r] = ln.q == 0 ? obar[r] : (ln.q == 1 ? obar[4 + r] : (ln.q == 2 ? obar[8 + r] : obar[12 + r]));
    zero_act<T>(hb);
    sq_bwd<T, 1>(hb, L + Y::oV2, ln, ob);
    mul_d1<ACT, T>(hb, h);
    return to4_rep<T>(L + Y::oV1T, ln, hb);
  }
  float unscale = 1.0f;
  if (Y::HF) {
    f16x8 b1, b2;
    h1_bwd_operands<HID, MM>(L, ln, obar, b1, b2, unscale);
    keep_lds_reads_local();
    const char* base = reinterpret_cast<const char*>(L + Y::oV2T) + ln.i * 32 + (ln.q & 1) * 16;
#pragma unroll
    for (int nt = 0; nt < T; ++nt) {
      f16x8 ah = *reinterpret_cast<const f16x8*>(base + nt * 512);
      f16x8 al = *reinterpret_cast<const f16x8*>(base + Y::BPART + nt * 512);
      hb.v[nt] = mfma_h(ah, b1, mfma_h(al, b2, splat4(0.f)));
    }
  } else {
    Act<1> ob;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      ob.v[0][r] = ln.q == 0 ? obar[r] : (ln.q == 1 ? obar[4 + r] : (ln.q == 2 ? obar[8 + r] : obar[12 + r]));
    zero_act<T>(hb);
    sq_bwd<T, 1>(hb, L + Y::oV2, ln, ob);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) hb.v[t] = hb.v[t] * dtanh(h.v[t]);
  if (WG) store_rec_scaled<T>(rec, ln, hb, Y::HF ? unscale : 1.0f);
  f32x4 xb = to4_rep<T>(L + Y::oV1T, ln, hb);
  return Y::HF ? xb * unscale : xb;
}

// S = sym(R_raw) = (R_raw + R_raw^T) / 2 (src/pHNN.py:77-80) from the 16 outputs of R_net: the diagonal is exact
// ((x + x) / 2 == x) and each off-diagonal pair is formed once.
template <int N>
DEV void sym_from_rf(const float (&rf)[16], float (&S)[N][N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    S[i][i] = rf[i * N + i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) S[i][j] = S[j][i] = (rf[i * N + j] + rf[j * N + i]) * 0.5f;
  }
}
// Cotangent of R_raw from the dissipation term -S S^T dH:  Sbar = -(lam (S^T dH)^T + dH (S^T lam)^T),
// R_raw_bar = (Sbar + Sbar^T) / 2 -- symmetric: the upper triangle is formed once and mirrored, the diagonal is -m_ii.
// Shared by the whole-tile and the split-tile adjoint (bitwise the same by construction).
template <int N>
DEV void rbar_from(f32x4 lam, f32x4 dH, const float (&Stl)[N], const float (&StdH)[N], float (&rbar)[16]) {
  float m[N][N];
#pragma unroll
  for (int k = 0; k < 16; ++k) rbar[k] = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) m[i][j] = __builtin_fmaf(lam[i], StdH[j], dH[i] * Stl[j]);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    rbar[i * N + i] = -m[i][i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) rbar[i * N + j] = rbar[j * N + i] = (m[i][j] + m[j][i]) * -0.5f;
  }
}

// ------------------------------------------------------------------------------------------------
// Model: pHNN (src/pHNN.py:52-100)
// ------------------------------------------------------------------------------------------------
template <int N_, int HID_, bool FIXG_, int MM_ = MM_F32, int MI_ = 1, int ACT_ = ACT_TANH>
struct PhnnModel {
  static constexpr int N = N_, HID = HID_, T = HID / 16, MM = MM_, MI = MI_, ACT = ACT_;  // MI = input_dim m (controls per step)
  static constexpr bool FIXG = FIXG_, SPLIT = false;
  static_assert(N_ * MI_ <= 16 && MI_ <= 4, "G_net output n*m <= 16, m <= 4");
  static constexpr int SCR = kScrFloats;  // per-wave LDS scratch (exchange of the 16 R_net / G_net outputs)
  static constexpr int oH = 0;
  static constexpr int oR = oH + LayH2<HID, MM>::SIZE;
  static constexpr int oGn = oR + LayH1<HID, MM>::SIZE;
  static constexpr int oJ = oGn + (FIXG ? 0 : LayH1<HID, MM>::SIZE);  // [16] J - J^T, row-major N x N
  static constexpr int oG = oJ + 16;                               // [16] G_fixed, row-major N x MI
  static constexpr int IMG = oG + 16;

  // floats one wave stashes per step for the adjoint: a2, q1 (T x 256 each) + dH (16 x 4) + the 16 outputs of R_net
  // (16 x 16; K2 then needs only R_net's hidden layer); a1 is recomputed
  static constexpr int oStashRf = 2 * T * 256 + 64;
  static constexpr int STASH = oStashRf + 256;

  // dx = (Jeff - S S^T) dH + G u, with S = sym(R_raw).  stash != null: keep the H_net tape for K2.
  template <bool WANT_H, bool ST = false>
  DEV static f32x4 f(const float* L, float* scr, Lane ln, f32x4 x, f32x4 u, float& Hval, float* stash = nullptr) {
    keep_lds_reads_local();
    HTape<HID> tp;
    f32x4 dH = hnet_grad<HID, WANT_H, MM, kInHFwd, ACT>(L + oH, ln, x, tp, Hval);
    if (ST) {
      store_act<T>(stash, ln, tp.a2);
      store_act<T>(stash + T * 256, ln, tp.q1);
      if (ln.q == 0) PHNN_NT_STORE(dH, reinterpret_cast<f32x4*>(stash + 2 * T * 256) + ln.i);
    }
    Act<T> hR;
    float rf[16];
    h1_fwd<HID, MM, kInHNet1, ACT>(L + oR, scr, ln, x, hR, rf, ST ? stash + oStashRf : nullptr);
    float G[N * MI];  // G(x) row-major (N, MI): G_fixed buffer or G_net(x).view(n, m)  (src/pHNN.py:86-92)
    if (FIXG) {
#pragma unroll
      for (int i = 0; i < N * MI; ++i) G[i] = L[oG + i];
    } else {
      Act<T> hG;
      float gf[16];
      h1_fwd<HID, MM, kInHNet1, ACT>(L + oGn, scr, ln, x, hG, gf);
#pragma unroll
      for (int i = 0; i < N * MI; ++i) G[i] = gf[i];
    }
    return combine(L, rf, dH, G, u);
  }

  DEV static f32x4 combine(const float* L, const float (&rf)[16], f32x4 dH, const float (&G)[N * MI], f32x4 u) {
    float S[N][N], StdH[N];
    sym_from_rf<N>(rf, S);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) a = __builtin_fmaf(S[i][k], dH[i], a);
      StdH[k] = a;
    }
    f32x4 dx = splat4(0.f);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) acc = __builtin_fmaf(L[oJ + i * N + j], dH[j], acc);
#pragma unroll
      for (int k = 0; k < N; ++k) acc = __builtin_fmaf(-S[i][k], StdH[k], acc);
      if (MI == 1) {
        dx[i] = __builtin_fmaf(G[i], u[0], acc);
      } else {  // acc + sum_k G[i][k] u_k, the G u product formed first as bmm does (src/pHNN.py:97)
        float gu = 0.f;
#pragma unroll
        for (int k = 0; k < MI; ++k) gu = __builtin_fmaf(G[i * MI + k], u[k], gu);
        dx[i] = acc + gu;
      }
    }
    return dx;
  }

  // weight-gradient record of one evaluation (WRec): a2, q1, ad2, qd, hbR [, hbG] + the per-rollout small vectors
  // hbR / hbG are part of the record only when the output layers are not f16x2 images: with those the reduction
  // recomputes them from rbar / gbar and V2 (four f32 MFMAs per record and wave instead of 1 KB of record per tile)
  static constexpr bool RECHB = !LayH1<HID, MM>::HF;
  static constexpr int NBIG = 4 + (RECHB ? (FIXG ? 1 : 2) : 0);
  using Rec = WRec<T, NBIG>;

  // xbar = (df/dx)^T lam, ubar = (df/du)^T lam at (x,u); recomputes the forward tape it needs.
  // WG: also writes the weight-gradient record of this evaluation to `rec` (Hbar = cotangent on H, adds Hbar dH to xbar).
  template <bool ST = false, bool WG = false>
  DEV static void vjp(const float* L, float* scr, Lane ln, f32x4 x, f32x4 u, f32x4 lam, f32x4& xbar, f32x4& ubar,
                      const float* stash = nullptr, float* rec = nullptr, float Hbar = 0.f) {
    keep_lds_reads_local();
    HTape<HID> tp;
    float Hdummy;
    f32x4 dH;
    float rf[16];
    if (ST) {  // tape written by K1: the loads fly while a1 is recomputed and the R_net part below runs.  Loads return
      // in issue order (vmcnt): the small vectors the R_net part needs first are requested first, the big ones after
      dH = PHNN_NT_LOAD(reinterpret_cast<const f32x4*>(stash + 2 * T * 256) + ln.i);
      load_rf(stash + oStashRf, ln, rf);
      load_act<T>(stash, ln, tp.a2);
      load_act<T>(stash + T * 256, ln, tp.q1);
      // (a1 is recomputed further down, right before the Hessian-vector product: it is not live through the R_net part,
      // which is where this kernel's register peak sits.  K2 1.179 -> 1.154 ms.  Issuing a2's loads later too: no change.)
    } else {
      dH = hnet_grad<HID, false, MM, kInHRecomp, ACT>(L + oH, ln, x, tp, Hdummy);
    }
    if (WG && !ST) {  // with the tapes of K1 the reduction reads a2, q1 from there (their record slots stay unwritten)
      store_rec<T>(rec, ln, tp.a2);
      store_rec<T>(rec + Rec::VEC, ln, tp.q1);
    }
    f32x4 xb = splat4(0.f);
    float S[N][N], Stl[N], StdH[N];
    {
      Act<T> hR;
      if (ST) {  // R_net's outputs came with the tape: only its hidden layer is re-evaluated
        h1_hidden<HID, MM, kInHNet1Adj, ACT>(L + oR, ln, x, hR);
      } else {
        h1_fwd<HID, MM, kInHNet1Adj, ACT>(L + oR, scr, ln, x, hR, rf);
      }
      sym_from_rf<N>(rf, S);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          a = __builtin_fmaf(S[i][k], lam[i], a);
          c = __builtin_fmaf(S[i][k], dH[i], c);
        }
        Stl[k] = a;
        StdH[k] = c;
      }
      // dissipation term: Sbar = -(lam (S^T dH)^T + dH (S^T lam)^T), R_raw_bar = (Sbar + Sbar^T)/2
      float rbar[16];
      rbar_from<N>(lam, dH, Stl, StdH, rbar);
      xb += h1_bwd<HID, MM, WG && RECHB, ACT>(L + oR, ln, hR, rbar, WG && RECHB ? rec + 4 * Rec::VEC : nullptr);
      if (WG && ln.q == 0) {
        f32x4* sm = reinterpret_cast<f32x4*>(rec + Rec::oSmall + ln.i * kRecSmall);
#pragma unroll
        for (int k = 0; k < 4; ++k) sm[4 + k] = f32x4{rbar[4 * k], rbar[4 * k + 1], rbar[4 * k + 2], rbar[4 * k + 3]};
      }
    }
    ubar = splat4(0.f);
    if (FIXG) {
#pragma unroll
      for (int k = 0; k < MI; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) ubar[k] = __builtin_fmaf(L[oG + i * MI + k], lam[i], ubar[k]);
    } else {
      Act<T> hG;
      float gf[16], gbar[16];
      h1_fwd<HID, MM, kInHNet1Adj, ACT>(L + oGn, scr, ln, x, hG, gf);
#pragma unroll
      for (int k = 0; k < 16; ++k) gbar[k] = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < MI; ++k) {
          ubar[k] = __builtin_fmaf(gf[i * MI + k], lam[i], ubar[k]);
          gbar[i * MI + k] = lam[i] * u[k];
        }
      xb += h1_bwd<HID, MM, WG && RECHB, ACT>(L + oGn, ln, hG, gbar, WG && RECHB ? rec + 5 * Rec::VEC : nullptr);
    }
    // v = A^T lam, A = Jeff - S S^T
    f32x4 v = splat4(0.f);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) acc = __builtin_fmaf(L[oJ + i * N + j], lam[i], acc);
#pragma unroll
      for (int k = 0; k < N; ++k) acc = __builtin_fmaf(-S[j][k], Stl[k], acc);
      v[j] = acc;
    }
    if (WG && ln.q == 0) {
      f32x4* sm = reinterpret_cast<f32x4*>(rec + Rec::oSmall + ln.i * kRecSmall);
      sm[0] = x;
      sm[1] = v;
      sm[2] = lam;
      sm[3] = dH;
      sm[8] = u;
      sm[9] = f32x4{Hbar, 0.f, 0.f, 0.f};
    }
    if (ST) hnet_layer1<HID, MM, ACT>(L + oH, ln, x, tp.a1);
    xbar = xb + hnet_hvp<HID, MM, WG, ACT>(L + oH, ln, tp, v, rec);
    if (WG) xbar = xbar + Hbar * dH;
  }
};

// ------------------------------------------------------------------------------------------------
// M_net.mlp of the general MassMatrixNetwork (src/mass_matrix.py:59-98): in(q_dim = 2) -> 64 -> 64 -> out (2 diagonal
// log-entries or 3 Cholesky entries), all-f32 MFMA layers (exact f32; the net is small and not on the headline path).
// ------------------------------------------------------------------------------------------------
struct LayM {
  static constexpr int HID = 64, T = 4, LD = HID + 4, LR = HID + 8;
  static constexpr int oW1f = 0;                 // [T][64] fragment image of W1 (64 x 2)
  static constexpr int oB1 = oW1f + T * 64;      // [64]
  static constexpr int oW2 = oB1 + HID;          // [64][LD]
  static constexpr int oB2 = oW2 + HID * LD;     // [64]
  static constexpr int oWo = oB2 + HID;          // [4][LR] rows c = Wout[c,:]
  static constexpr int oBo = oWo + 4 * LR;       // [4]
  static constexpr int oWoTf = oBo + 4;          // [T][64] fragment image of Wout^T (64 x out)
  static constexpr int oW1T = oWoTf + T * 64;    // [4][LR] rows c = W1[:,c]
  static constexpr int SIZE = oW1T + 4 * LR;
};

struct MTape {
  Act<4> a1, a2;
  f32x4 o;
};

DEV void mnet_fwd(const float* L, Lane ln, f32x4 q, MTape& tp) {
  using Y = LayM;
  keep_lds_reads_local();
  load_vec<4>(tp.a1, L + Y::oB1, ln);
  in_layer<4>(tp.a1, L + Y::oW1f, ln, sel4(q, ln.q));
  tanh_act<4>(tp.a1);
  load_vec<4>(tp.a2, L + Y::oB2, ln);
  sq_fwd<4, 4>(tp.a2, L + Y::oW2, ln, tp.a1);
  tanh_act<4>(tp.a2);
  tp.o = to4_rep<4>(L + Y::oWo, ln, tp.a2) + *reinterpret_cast<const f32x4*>(L + Y::oBo);
}

DEV f32x4 mnet_bwd(const float* L, Lane ln, const MTape& tp, f32x4 ob) {  // qbar = (d o / d q)^T ob
  using Y = LayM;
  keep_lds_reads_local();
  Act<4> d, e;
  zero_act<4>(d);
  in_layer<4>(d, L + Y::oWoTf, ln, sel4(ob, ln.q));
#pragma unroll
  for (int t = 0; t < 4; ++t) d.v[t] = d.v[t] * dtanh(tp.a2.v[t]);
  zero_act<4>(e);
  sq_bwd<4, 4>(e, L + Y::oW2, ln, d);
#pragma unroll
  for (int t = 0; t < 4; ++t) e.v[t] = e.v[t] * dtanh(tp.a1.v[t]);
  return to4_rep<4>(L + Y::oW1T, ln, e);
}

DEV float softplus_dev(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch softplus, threshold 20
DEV float sigmoid_sp_dev(float x) { return x > 20.f ? 1.0f : 1.0f / (1.0f + expf(-x)); }

constexpr int MASS_CARTPOLE = 0, MASS_CONSTANT = 1, MASS_DIAGONAL = 2, MASS_FULL = 3;

// M (m00, m01, m11) and M^-1 (w00, w01, w11) of MassMatrixNetwork's diagonal / full types from the mlp outputs o
template <int MT>
DEV void mass_from_outputs(f32x4 o, float (&m)[3], float (&w)[3]) {
  if (MT == MASS_DIAGONAL) {  // M = diag(exp(o) + 1e-3)  (src/mass_matrix.py:157-161, 196-200)
    m[0] = expf(o[0]) + 1e-3f;
    m[2] = expf(o[1]) + 1e-3f;
    m[1] = 0.f;
    w[0] = 1.0f / m[0];
    w[2] = 1.0f / m[2];
    w[1] = 0.f;
  } else {  // M = L L^T, L = [[softplus(o0) + 1e-3, 0], [o1, softplus(o2) + 1e-3]]; M^-1 = inverse of the 2x2 matrix
    const float l00 = softplus_dev(o[0]) + 1e-3f, l10 = o[1], l11 = softplus_dev(o[2]) + 1e-3f;
    m[0] = l00 * l00;
    m[1] = l00 * l10;
    m[2] = l10 * l10 + l11 * l11;
    const float det = m[0] * m[2] - m[1] * m[1];
    w[0] = m[2] / det;
    w[1] = -m[1] / det;
    w[2] = m[0] / det;
  }
}

// cotangent of the mlp outputs from the full 2x2 cotangent Mb of M (row-major)
template <int MT>
DEV f32x4 mass_outputs_bar(f32x4 o, const float (&Mb)[4]) {
  if (MT == MASS_DIAGONAL) return f32x4{Mb[0] * expf(o[0]), Mb[3] * expf(o[1]), 0.f, 0.f};
  const float l00 = softplus_dev(o[0]) + 1e-3f, l10 = o[1], l11 = softplus_dev(o[2]) + 1e-3f;
  const float s00 = 2.0f * Mb[0], s01 = Mb[1] + Mb[2], s11 = 2.0f * Mb[3];  // Lb = (Mb + Mb^T) L, lower triangle
  return f32x4{(s00 * l00 + s01 * l10) * sigmoid_sp_dev(o[0]), s01 * l00 + s11 * l10, (s11 * l11) * sigmoid_sp_dev(o[2]), 0.f};
}

// ------------------------------------------------------------------------------------------------
// Model: canonical pHNN with the cart-pole mass matrix (src/pHNN_canonical.py:172-273,
// src/mass_matrix.py:270-362, src/coordinate_transforms.py:20-130)
// ------------------------------------------------------------------------------------------------
template <int HID_, int MM_ = MM_F32, int MI_ = 1, int MT_ = MASS_CARTPOLE, int ACT_ = ACT_TANH>
struct CanonModel {
  static constexpr int N = 4, HID = HID_, T = HID / 16, MM = MM_, MI = MI_, MT = MT_, ACT = ACT_;  // MT: mass matrix type
  static constexpr bool SPLIT = false;
  static constexpr int SCR = 0;  // no per-wave LDS scratch needed
  static constexpr int oH = 0;
  static constexpr int oC = oH + LayH2<HID, MM>::SIZE;  // a, b, c, 0 | Rd[4] | sigmoid(R_diag_raw)[4] | G[4][MI] (16 slots)
  static constexpr int oCG = oC + 12;
  static constexpr int oCW = oC + 28;  // MT = constant: M^-1 entries (w00, w01, w11, 0); M itself sits in a, b, c
  static constexpr int oMn = oC + 32;  // MT = diagonal / full: LayM image of M_net.mlp
  static constexpr int IMG = oMn + (MT_ >= MASS_DIAGONAL ? LayM::SIZE : 0);

  // M(q) and M^-1(q) for the MassMatrixNetwork types (MT >= 1)
  DEV static void mass_eval(const float* L, Lane ln, f32x4 y, float (&m)[3], float (&w)[3], MTape& mt) {
    if (MT == MASS_CONSTANT) {
      m[0] = L[oC + 0]; m[1] = L[oC + 1]; m[2] = L[oC + 2];
      w[0] = L[oCW + 0]; w[1] = L[oCW + 1]; w[2] = L[oCW + 2];
    } else {
      mnet_fwd(L + oMn, ln, f32x4{y[0], y[1], 0.f, 0.f}, mt);
      mass_from_outputs<MT>(mt.o, m, w);
    }
  }

  DEV static float Base_Gu(const float* L, int row, f32x4 u) {  // (G u)_row
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MI; ++k) s = __builtin_fmaf(L[oCG + row * MI + k], u[k], s);
    return s;
  }
  DEV static f32x4 Base_Gt(const float* L, float dpb0, float dpb1) {  // G^T [0, 0, dpb0, dpb1]: only the dp rows reach the output
    f32x4 ub = splat4(0.f);
#pragma unroll
    for (int k = 0; k < MI; ++k) ub[k] = L[oCG + 2 * MI + k] * dpb0 + L[oCG + 3 * MI + k] * dpb1;
    return ub;
  }

  static constexpr int STASH = 2 * T * 256 + 64;

  template <bool WANT_H, bool ST = false>
  DEV static f32x4 f(const float* L, float* scr, Lane ln, f32x4 y, f32x4 u, float& Hval, float* stash = nullptr) {
    keep_lds_reads_local();
    if constexpr (MT != MASS_CARTPOLE) {  // general MassMatrixNetwork: no det fudge, M^-1 as the reference forms it
      float m[3], w[3];
      MTape mt;
      mass_eval(L, ln, y, m, w, mt);
      f32x4 z = {y[0], y[1], m[0] * y[2] + m[1] * y[3], m[1] * y[2] + m[2] * y[3]};
      HTape<HID> tp;
      f32x4 dH = hnet_grad<HID, WANT_H, MM, kInHFwd, ACT>(L + oH, ln, z, tp, Hval);
      if (ST) {
        store_act<T>(stash, ln, tp.a2);
        store_act<T>(stash + T * 256, ln, tp.q1);
        if (ln.q == 0) PHNN_NT_STORE(dH, reinterpret_cast<f32x4*>(stash + 2 * T * 256) + ln.i);
      }
      float dp0 = (-dH[0] - L[oC + 6] * dH[2]) + Base_Gu(L, 2, u);
      float dp1 = (-dH[1] - L[oC + 7] * dH[3]) + Base_Gu(L, 3, u);
      return f32x4{w[0] * z[2] + w[1] * z[3], w[1] * z[2] + w[2] * z[3], w[0] * dp0 + w[1] * dp1, w[1] * dp0 + w[2] * dp1};
    }
    float a = L[oC + 0], b = L[oC + 1], c = L[oC + 2];
    float sn, cs;
    sincos_dev(y[1], sn, cs);
    float bc = b * cs;
    f32x4 z = {y[0], y[1], a * y[2] + bc * y[3], bc * y[2] + c * y[3]};
    HTape<HID> tp;
    f32x4 dH = hnet_grad<HID, WANT_H, MM, kInHFwd, ACT>(L + oH, ln, z, tp, Hval);
    if (ST) {
      store_act<T>(stash, ln, tp.a2);
      store_act<T>(stash + T * 256, ln, tp.q1);
      if (ln.q == 0) PHNN_NT_STORE(dH, reinterpret_cast<f32x4*>(stash + 2 * T * 256) + ln.i);
    }
    float dp0 = (-dH[0] - L[oC + 6] * dH[2]) + Base_Gu(L, 2, u);
    float dp1 = (-dH[1] - L[oC + 7] * dH[3]) + Base_Gu(L, 3, u);
    float det = (a * c - bc * bc) + 1e-6f;
    float mi00 = c / det, mi01 = -bc / det, mi11 = a / det;
    return f32x4{mi00 * z[2] + mi01 * z[3], mi01 * z[2] + mi11 * z[3], mi00 * dp0 + mi01 * dp1,
                 mi01 * dp0 + mi11 * dp1};
  }

  static constexpr int NBIG = 4;
  using Rec = WRec<T, NBIG>;

  // WG: also writes the weight-gradient record (a2, q1, ad2, qd; small: z, v, lam, dH, the two R_diag cotangents)
  template <bool ST = false, bool WG = false>
  DEV static void vjp(const float* L, float* scr, Lane ln, f32x4 y, f32x4 u, f32x4 lam, f32x4& ybar, f32x4& ubar,
                      const float* stash = nullptr, float* rec = nullptr, float Hbar = 0.f) {
    keep_lds_reads_local();
    if constexpr (MT != MASS_CARTPOLE) {
      // WG (MassMatrixNetwork): the record carries what H_net and R_diag_raw need, as below, plus -- per evaluation and
      // rollout -- q and the cotangent Mb of the 2 x 2 matrix M(q) (small vectors 5, 6).  The mass network's own
      // parameter gradient is then ONE autograd pass of the caller's MassMatrixNetwork module over those points
      // (phnn_mpc_amd/models.py: mass_param_grads): it is a 2 -> 64 -> 64 -> 3 net outside the hot loop.
      float m[3], w[3];
      MTape mt;
      mass_eval(L, ln, y, m, w, mt);
      f32x4 z = {y[0], y[1], m[0] * y[2] + m[1] * y[3], m[1] * y[2] + m[2] * y[3]};
      HTape<HID> tp;
      float Hdummy;
      f32x4 dH;
      if (ST) {
        dH = PHNN_NT_LOAD(reinterpret_cast<const f32x4*>(stash + 2 * T * 256) + ln.i);  // needed first: requested first
        load_act<T>(stash, ln, tp.a2);
        load_act<T>(stash + T * 256, ln, tp.q1);
        hnet_layer1<HID, MM, ACT>(L + oH, ln, z, tp.a1);
      } else {
        dH = hnet_grad<HID, false, MM, kInHRecomp, ACT>(L + oH, ln, z, tp, Hdummy);
      }
      const float Rd2 = L[oC + 6], Rd3 = L[oC + 7];
      const float dp0 = (-dH[0] - Rd2 * dH[2]) + Base_Gu(L, 2, u);
      const float dp1 = (-dH[1] - Rd3 * dH[3]) + Base_Gu(L, 3, u);
      const float pb0 = lam[0] * w[0] + lam[1] * w[1], pb1 = lam[0] * w[1] + lam[1] * w[2];      // W lam_q
      const float dpb0 = lam[2] * w[0] + lam[3] * w[1], dpb1 = lam[2] * w[1] + lam[3] * w[2];    // W lam_v
      f32x4 v = {-dpb0, -dpb1, -Rd2 * dpb0, -Rd3 * dpb1};
      ubar = Base_Gt(L, dpb0, dpb1);
      if (WG) {
        if (!ST) {
          store_rec<T>(rec, ln, tp.a2);
          store_rec<T>(rec + Rec::VEC, ln, tp.q1);
        }
        if (ln.q == 0) {
          f32x4* sm = reinterpret_cast<f32x4*>(rec + Rec::oSmall + ln.i * kRecSmall);
          sm[0] = z;
          sm[1] = v;
          sm[2] = lam;
          sm[3] = dH;
          sm[4] = f32x4{0.f, 0.f, -dpb0 * dH[2], -dpb1 * dH[3]};  // R_diag_raw rows 2, 3 (as the cart-pole branch)
          sm[8] = u;
          sm[9] = f32x4{Hbar, 0.f, 0.f, 0.f};
        }
      }
      f32x4 zb = hnet_hvp<HID, MM, WG, ACT>(L + oH, ln, tp, v, rec);
      if (WG) zb = zb + Hbar * dH;
      zb[2] += pb0;
      zb[3] += pb1;
      ybar = f32x4{zb[0], zb[1], zb[2] * m[0] + zb[3] * m[1], zb[2] * m[1] + zb[3] * m[2]};
      if (MT >= MASS_DIAGONAL || WG) {  // M(q) carries gradient: p = M qdot and the two uses of M^-1 (d W = -W dM W)
        const float W4[4] = {w[0], w[1], w[1], w[2]};
        const float Wb[4] = {lam[0] * z[2] + lam[2] * dp0, lam[0] * z[3] + lam[2] * dp1,
                             lam[1] * z[2] + lam[3] * dp0, lam[1] * z[3] + lam[3] * dp1};
        float Tm[4], Um[4], Mb[4];
        Tm[0] = W4[0] * Wb[0] + W4[1] * Wb[2]; Tm[1] = W4[0] * Wb[1] + W4[1] * Wb[3];
        Tm[2] = W4[2] * Wb[0] + W4[3] * Wb[2]; Tm[3] = W4[2] * Wb[1] + W4[3] * Wb[3];
        Um[0] = Tm[0] * W4[0] + Tm[1] * W4[2]; Um[1] = Tm[0] * W4[1] + Tm[1] * W4[3];
        Um[2] = Tm[2] * W4[0] + Tm[3] * W4[2]; Um[3] = Tm[2] * W4[1] + Tm[3] * W4[3];
        Mb[0] = zb[2] * y[2] - Um[0];
        Mb[1] = zb[2] * y[3] - Um[1];
        Mb[2] = zb[3] * y[2] - Um[2];
        Mb[3] = zb[3] * y[3] - Um[3];
        if (WG && ln.q == 0) {
          f32x4* sm = reinterpret_cast<f32x4*>(rec + Rec::oSmall + ln.i * kRecSmall);
          sm[5] = f32x4{Mb[0], Mb[1], Mb[2], Mb[3]};
          sm[6] = f32x4{y[0], y[1], 0.f, 0.f};
        }
        if (MT >= MASS_DIAGONAL) {
          f32x4 qb = mnet_bwd(L + oMn, ln, mt, mass_outputs_bar<MT>(mt.o, Mb));
          ybar[0] += qb[0];
          ybar[1] += qb[1];
        }
      }
      return;
    }
    float a = L[oC + 0], b = L[oC + 1], c = L[oC + 2];
    float sn, cs;
    sincos_dev(y[1], sn, cs);
    float bc = b * cs;
    f32x4 z = {y[0], y[1], a * y[2] + bc * y[3], bc * y[2] + c * y[3]};
    HTape<HID> tp;
    float Hdummy;
    f32x4 dH;
    if (ST) {
      dH = PHNN_NT_LOAD(reinterpret_cast<const f32x4*>(stash + 2 * T * 256) + ln.i);  // needed first: requested first
      load_act<T>(stash, ln, tp.a2);
      load_act<T>(stash + T * 256, ln, tp.q1);
      hnet_layer1<HID, MM, ACT>(L + oH, ln, z, tp.a1);
    } else {
      dH = hnet_grad<HID, false, MM, kInHRecomp, ACT>(L + oH, ln, z, tp, Hdummy);
    }
    float Rd2 = L[oC + 6], Rd3 = L[oC + 7];
    float dp0 = (-dH[0] - Rd2 * dH[2]) + Base_Gu(L, 2, u);
    float dp1 = (-dH[1] - Rd3 * dH[3]) + Base_Gu(L, 3, u);
    float det = (a * c - bc * bc) + 1e-6f;
    float rdet = 1.0f / det;
    float mi00 = c * rdet, mi01 = -bc * rdet, mi11 = a * rdet;
    float pb0 = lam[0] * mi00 + lam[1] * mi01, pb1 = lam[0] * mi01 + lam[1] * mi11;
    float dpb0 = lam[2] * mi00 + lam[3] * mi01, dpb1 = lam[2] * mi01 + lam[3] * mi11;
    float mb00 = lam[0] * z[2] + lam[2] * dp0;
    float mb01 = lam[0] * z[3] + lam[1] * z[2] + lam[2] * dp1 + lam[3] * dp0;
    float mb11 = lam[1] * z[3] + lam[3] * dp1;
    f32x4 v = {-dpb0, -dpb1, -Rd2 * dpb0, -Rd3 * dpb1};
    ubar = Base_Gt(L, dpb0, dpb1);
    if (WG) {
      if (!ST) {
        store_rec<T>(rec, ln, tp.a2);
        store_rec<T>(rec + Rec::VEC, ln, tp.q1);
      }
      if (ln.q == 0) {
        f32x4* sm = reinterpret_cast<f32x4*>(rec + Rec::oSmall + ln.i * kRecSmall);
        sm[0] = z;
        sm[1] = v;
        sm[2] = lam;
        sm[3] = dH;
        // d(lam^T f)/d Rd_{2,3}: dp_i = -dH_i - Rd_{2+i} dH_{2+i} + ...  (softplus' is applied by the reduce kernel)
        sm[4] = f32x4{0.f, 0.f, -dpb0 * dH[2], -dpb1 * dH[3]};
        sm[8] = u;
        sm[9] = f32x4{Hbar, 0.f, 0.f, 0.f};
      }
    }
    f32x4 zb = hnet_hvp<HID, MM, WG, ACT>(L + oH, ln, tp, v, rec);
    if (WG) zb = zb + Hbar * dH;
    zb[2] += pb0;
    zb[3] += pb1;
    float bcb = zb[2] * y[3] + zb[3] * y[2];
    float detb = (-(mb00 * c + mb11 * a) + mb01 * bc) * (rdet * rdet);
    bcb += -mb01 * rdet;
    bcb += -2.0f * bc * detb;
    ybar = f32x4{zb[0], zb[1] + bcb * (-b * sn), zb[2] * a + zb[3] * bc, zb[2] * bc + zb[3] * c};
  }
};

// ------------------------------------------------------------------------------------------------
// Split-tile models: FOUR waves share one 16-rollout tile (small batches, single plants).
//
// With fewer than ~2 tiles per CU the whole-tile kernels leave one wave per CU marching alone (every vector
// instruction issued by a single wave, three SIMDs idle).  Here the hidden units are split instead: wave w of a
// 4-wave workgroup owns accumulator tiles 2w, 2w+1 of every 128-wide vector -- exactly k-step s = w of the next
// hidden x hidden product, whose B fragment lane (i,q) builds from tiles 2s, 2s+1 of the SAME lane.  Exchange between
// layers: each wave writes the hi/lo fragments of its own k-step (2 x 16 B per lane) to LDS, one barrier, every wave
// reads the four k-steps (8 x 16 B per lane, conflict-free) and runs its 2 output tiles x 4 k-steps x 3 products =
// 24 MFMAs instead of 96; tanh, splits and the element-wise adjoint work shrink by four as well.  The 128 -> 4
// reductions leave one partial per wave (to4_group), summed in the fixed order of to4_rep; the small 128 -> 16 output
// layer of R_net runs redundantly on all four waves from the exchanged fragments.  Every sum keeps the association of
// the whole-tile kernels, so results are BITWISE identical to them (tests compare slices of a large batch, run by the
// whole-tile kernels, with the same rollouts run alone by these).  The stash format is the same too.
// Exchange area (floats): three fragment regions [part 2][k-step 4][lane 64] x 16 B, two partial-sum sets
// [wave 4][lane 64] x 16 B.  HID = 128, f16x2 products only.
// ------------------------------------------------------------------------------------------------
constexpr int kXR0 = 0, kXR1 = 2048, kXR2 = 4096, kXP0 = 6144, kXP1 = 7168, kXchFloats = 8192;
using ActW = Act<2>;

DEV void xch_put(float* region, Lane ln, const Split2<2>& sp) {  // this wave's k-step (= ln.w)
  f16x8* p = reinterpret_cast<f16x8*>(region);
  p[ln.w * 64 + ln.lane] = sp.h[0];
  p[(4 + ln.w) * 64 + ln.lane] = sp.l[0];
}
DEV void xch_get(const float* region, Lane ln, Split2<8>& o) {
  const f16x8* p = reinterpret_cast<const f16x8*>(region);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    o.h[s] = p[s * 64 + ln.lane];
    o.l[s] = p[(4 + s) * 64 + ln.lane];
  }
}
DEV void xch_put_partial(float* set, Lane ln, f32x4 v) { reinterpret_cast<f32x4*>(set)[ln.w * 64 + ln.lane] = v; }
DEV f32x4 xch_sum_partials(const float* set, Lane ln) {  // ((P0 + P1) + P2) + P3, then the k-slots: to4_rep's order
  const f32x4* p = reinterpret_cast<const f32x4*>(set) + ln.lane;
  return to4_kslots(((p[0] + p[64]) + p[128]) + p[192]);
}

DEV void tanh_pre_w(ActW& a) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) a.v[t][r] = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a.v[t][r]) + 1.0f), 1.0f);
}

// o (tiles 2w, 2w+1) += rows of W times the full vector given as four k-step fragments; same chain per tile as sq_fwd_h
DEV void sq_fwd_h_w(ActW& o, const float* Wimg, Lane ln, const Split2<8>& in) {
  using I = HfImg<128>;
  keep_lds_reads_local();
  matrix_phase_begin();
  const char* base = reinterpret_cast<const char*>(Wimg) + ln.i * (I::RS * 2) + ln.q * 16 + (2 * ln.w) * 16 * (I::RS * 2);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f16x8 a[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[g][p] = *reinterpret_cast<const f16x8*>(base + p * I::PART + g * 16 * (I::RS * 2) + s * 64);
    mfma3x2(o.v[0], o.v[1], a, in.h[s], in.l[s]);
  }
  matrix_phase_end();
}

// o (tiles 2w, 2w+1) += columns of W (W^T product) times the full vector; same chain per tile as sq_bwd_h
DEV void sq_bwd_h_w(ActW& o, const float* Wimg, Lane ln, const Split2<8>& in) {
  using I = HfImg<128>;
  keep_lds_reads_local();
  matrix_phase_begin();
  typedef fp16x4_t __attribute__((address_space(3))) * lds_h4;
  typedef char __attribute__((address_space(3))) * lds_cp;
  const int a4 = (ln.lane & 15) >> 2, pp = ln.lane & 3;
  lds_cp base = (lds_cp) const_cast<char*>(reinterpret_cast<const char*>(Wimg)) + (4 * ln.q + a4) * (I::RS * 2) + 16 * pp + 64 * ln.w;
  // (requesting all four k-steps' fragments up front -- one wave per SIMD here, nothing else covers the LDS round trips --
  // was measured in round 3: single-plant solve 3.68 -> 3.74 ms, no gain; the step is bound by its barriers)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f16x8 a[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        lds_cp off = base + p * I::PART + 32 * s * (I::RS * 2) + 8 * g;
        f16x4 lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)off));
        f16x4 hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(off + 16 * (I::RS * 2))));
        a[g][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    mfma3x2(o.v[0], o.v[1], a, in.h[s], in.l[s]);
  }
  matrix_phase_end();
}

// H_net of a split tile: forward to (a1, a2, q1) on the own tiles and the wave's partial of dH (to4_group).
// Barriers: two inside (after the a1 and after the g2 fragments).  `pre_barrier` runs between the first fragment write
// and the first barrier (R_net's first layer shares that barrier).
struct HTapeW {
  ActW a1, a2, q1;
};

template <int SITE = kInHFwd, class PreBarrier>
DEV f32x4 hnet_grad_w(const float* L, Lane ln, f32x4 z, HTapeW& tp, PreBarrier pre_barrier) {
  using Y = LayH2<128, MM_F16X2>;
  const int t0 = 2 * ln.w;
  load_vec<2>(tp.a1, L + Y::oB1 + 16 * t0, ln);
  in_layer_mm<2, MM_F16X2, SITE>(tp.a1, L, Y::oW1f, Y::oW1h, ln, z, t0);
  tanh_pre_w(tp.a1);
  {
    Split2<2> sp;
    split_act_h<2>(tp.a1, sp);
    xch_put(ln.xch + kXR0, ln, sp);
  }
  pre_barrier();
  __syncthreads();
  {
    Split2<8> all;
    xch_get(ln.xch + kXR0, ln, all);
    load_vec<2>(tp.a2, L + Y::oB2 + 16 * t0, ln);
    sq_fwd_h_w(tp.a2, L + Y::oW2, ln, all);
  }
  const float c = L[Y::oB3 + 1];
  if (c == 1.0f) {
    tanh_pre_w(tp.a2);
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tp.a2.v[t][r] = tanh_scaled(tp.a2.v[t][r], c);
  }
  ActW g;
  keep_lds_reads_local();
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x4 w3b = *reinterpret_cast<const f32x4*>(L + Y::oW3B + 16 * (t0 + t) + 4 * ln.q);
    g.v[t] = w3b * dtanh(tp.a2.v[t]);
  }
  {
    Split2<2> sp;
    split_act_h<2>(g, sp);
    xch_put(ln.xch + kXR1, ln, sp);
  }
  __syncthreads();
  {
    Split2<8> all;
    xch_get(ln.xch + kXR1, ln, all);
    zero_act<2>(tp.q1);
    sq_bwd_h_w(tp.q1, L + Y::oW2, ln, all);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) g.v[t] = tp.q1.v[t] * dtanh(tp.a1.v[t]);
  keep_lds_reads_local();
  return to4_group<2>(L + Y::oW1T + (ln.i & 3) * Y::LR + 16 * t0, ln, g.v);
}

// Hessian-vector product of a split tile: the wave's partial (before the partial sum, the k-slot sum and the final
// scale); `scale` receives that final factor.  Two barriers inside.  Consumes tp.q1 like hnet_hvp.
DEV f32x4 hnet_hvp_w(const float* L, Lane ln, HTapeW& tp, f32x4 v, float& scale) {
  using Y = LayH2<128, MM_F16X2>;
  const int t0 = 2 * ln.w;
  float mx = fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
  int e = 0;
  (void)__builtin_frexpf(mx, &e);
  e = (mx > 0.f && mx < 3.0e38f) ? e : 0;
  v = v * __builtin_ldexpf(1.0f, -e);
  const float unscale = __builtin_ldexpf(1.0f, e);
  ActW ad1, w;
  zero_act<2>(ad1);
  in_layer_mm<2, MM_F16X2, kInHHvp>(ad1, L, Y::oW1f, Y::oW1h, ln, v, t0);
#pragma unroll
  for (int t = 0; t < 2; ++t) ad1.v[t] = dtanh(tp.a1.v[t]) * ad1.v[t];
#pragma unroll
  for (int t = 0; t < 2; ++t) tp.q1.v[t] = tp.q1.v[t] * (tp.a1.v[t] * ad1.v[t]);
  {
    Split2<2> sp;
    split_act_h<2>(ad1, sp);
    xch_put(ln.xch + kXR0, ln, sp);
  }
  __syncthreads();
  {
    Split2<8> all;
    xch_get(ln.xch + kXR0, ln, all);
    zero_act<2>(w);
    sq_fwd_h_w(w, L + Y::oW2, ln, all);
  }
  keep_lds_reads_local();
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x4 w3 = *reinterpret_cast<const f32x4*>(L + Y::oW3S + 16 * (t0 + t) + 4 * ln.q);
    f32x4 a2 = tp.a2.v[t];
    f32x4 ad2 = dtanh(a2) * w.v[t];
    w.v[t] = w3 * (a2 * ad2);
  }
  {
    Split2<2> sp;
    split_act_h<2>(w, sp);
    xch_put(ln.xch + kXR1, ln, sp);
  }
  __syncthreads();
  ActW qd;
  {
    Split2<8> all;
    xch_get(ln.xch + kXR1, ln, all);
    zero_act<2>(qd);
    sq_bwd_h_w(qd, L + Y::oW2, ln, all);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) qd.v[t] = __builtin_elementwise_fma(qd.v[t], dtanh(tp.a1.v[t]), tp.q1.v[t]);
  scale = -2.0f * unscale * L[Y::oB3 + 2];
  keep_lds_reads_local();
  return to4_group<2>(L + Y::oW1T + (ln.i & 3) * Y::LR + 16 * t0, ln, qd.v);
}

template <int N_>
struct PhnnSplit {  // PhnnModel<N_, 128, fixed G, f16x2> with the tile split over four waves
  using Base = PhnnModel<N_, 128, true, MM_F16X2>;
  static constexpr int N = N_, HID = 128, T = 8, MM = MM_F16X2;
  static constexpr bool FIXG = true, SPLIT = true;
  static constexpr int SCR = Base::SCR, IMG = Base::IMG, STASH = Base::STASH, MI = 1;
  static constexpr int oH = Base::oH, oR = Base::oR, oJ = Base::oJ, oG = Base::oG;
  using YR = LayH1<128, MM_F16X2>;

  // R_net hidden layer on the own tiles; fragments to region 2
  template <int SITE = kInHNet1>
  DEV static void rnet_hidden(const float* L, Lane ln, f32x4 x, ActW& hR) {  // own tiles only, nothing exchanged
    const int t0 = 2 * ln.w;
    load_vec<2>(hR, L + oR + YR::oC1 + 16 * t0, ln);
    in_layer_mm<2, MM_F16X2, SITE>(hR, L + oR, YR::oV1f, YR::oV1h, ln, x, t0);
    tanh_pre_w(hR);
  }
  template <int SITE = kInHNet1>
  DEV static void rnet_layer1(const float* L, Lane ln, f32x4 x, ActW& hR) {
    rnet_hidden<SITE>(L, ln, x, hR);
    Split2<2> sp;
    split_act_h<2>(hR, sp);
    xch_put(ln.xch + kXR2, ln, sp);
  }
  // R_net outputs from the exchanged fragments (after a barrier): all 16 in every lane
  DEV static void rnet_out(const float* L, float* scr, Lane ln, float (&rf)[16], float* rf_stash = nullptr) {
    Split2<8> all;
    xch_get(ln.xch + kXR2, ln, all);
    f32x4 o = h1_out_hf<128, MM_F16X2>(L + oR, ln, all);
    if (rf_stash) store_rf(rf_stash, ln, o);  // wave 0 writes (ln.w)
    gather16(scr, ln, o, rf);
  }

  template <bool WANT_H, bool ST = false>
  DEV static f32x4 f(const float* L, float* scr, Lane ln, f32x4 x, f32x4 u, float& Hval, float* stash = nullptr) {
    static_assert(!WANT_H, "the split-tile kernels are rollout kernels");
    keep_lds_reads_local();
    HTapeW tp;
    ActW hR;
    f32x4 P = hnet_grad_w(L + oH, ln, x, tp, [&]() { rnet_layer1(L, ln, x, hR); });
    xch_put_partial(ln.xch + kXP0, ln, P);
    float rf[16];
    rnet_out(L, scr, ln, rf, ST ? stash + Base::oStashRf : nullptr);
    if (ST) {
      store_act<2>(stash + 2 * ln.w * 256, ln, tp.a2);
      store_act<2>(stash + T * 256 + 2 * ln.w * 256, ln, tp.q1);
    }
    __syncthreads();
    f32x4 dH = xch_sum_partials(ln.xch + kXP0, ln);
    if (ST && ln.q == 0 && ln.w == 0) PHNN_NT_STORE(dH, reinterpret_cast<f32x4*>(stash + 2 * T * 256) + ln.i);
    float G[N];
#pragma unroll
    for (int i = 0; i < N; ++i) G[i] = L[oG + i];
    return Base::combine(L, rf, dH, G, u);
  }

  template <bool ST = false, bool WG = false>
  DEV static void vjp(const float* L, float* scr, Lane ln, f32x4 x, f32x4 u, f32x4 lam, f32x4& xbar, f32x4& ubar,
                      const float* stash = nullptr, float* rec = nullptr, float Hbar = 0.f) {
    static_assert(!WG, "weight-gradient records come from the whole-tile kernels");
    keep_lds_reads_local();
    const int t0 = 2 * ln.w;
    HTapeW tp;
    ActW hR;
    f32x4 dH;
    float rf[16];
    if (ST) {
      // small vectors first (vmcnt returns loads in issue order; the R_net part needs them first)
      dH = PHNN_NT_LOAD(reinterpret_cast<const f32x4*>(stash + 2 * T * 256) + ln.i);
      load_rf(stash + Base::oStashRf, ln, rf);  // R_net's outputs come with the tape: no fragment exchange, no barrier
      load_act<2>(stash + t0 * 256, ln, tp.a2);
      load_act<2>(stash + T * 256 + t0 * 256, ln, tp.q1);
      using Y = LayH2<128, MM_F16X2>;
      load_vec<2>(tp.a1, L + oH + Y::oB1 + 16 * t0, ln);
      in_layer_mm<2, MM_F16X2, kInHRecomp>(tp.a1, L + oH, Y::oW1f, Y::oW1h, ln, x, t0);
      tanh_pre_w(tp.a1);
      rnet_hidden<kInHNet1Adj>(L, ln, x, hR);
    } else {
      f32x4 P = hnet_grad_w<kInHRecomp>(L + oH, ln, x, tp, [&]() { rnet_layer1<kInHNet1Adj>(L, ln, x, hR); });
      xch_put_partial(ln.xch + kXP0, ln, P);
      rnet_out(L, scr, ln, rf);
      __syncthreads();
      dH = xch_sum_partials(ln.xch + kXP0, ln);
    }
    float S[N][N], Stl[N], StdH[N];
    sym_from_rf<N>(rf, S);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        a = __builtin_fmaf(S[i][k], lam[i], a);
        c = __builtin_fmaf(S[i][k], dH[i], c);
      }
      Stl[k] = a;
      StdH[k] = c;
    }
    float rbar[16];
    rbar_from<N>(lam, dH, Stl, StdH, rbar);
    // transposed output layer of R_net on the own tiles, then the wave's partial of V1^T (.)
    float unscaleR;
    f32x4 PR;
    {
      f16x8 b1, b2;
      h1_bwd_operands<128, MM_F16X2>(L + oR, ln, rbar, b1, b2, unscaleR);
      keep_lds_reads_local();
      const char* base = reinterpret_cast<const char*>(L + oR + YR::oV2T) + ln.i * 32 + (ln.q & 1) * 16;
      ActW hb;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f16x8 ah = *reinterpret_cast<const f16x8*>(base + (t0 + t) * 512);
        f16x8 al = *reinterpret_cast<const f16x8*>(base + YR::BPART + (t0 + t) * 512);
        hb.v[t] = mfma_h(ah, b1, mfma_h(al, b2, splat4(0.f)));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) hb.v[t] = hb.v[t] * dtanh(hR.v[t]);
      keep_lds_reads_local();
      PR = to4_group<2>(L + oR + YR::oV1T + (ln.i & 3) * YR::LR + 16 * t0, ln, hb.v);
    }
    ubar = splat4(0.f);
#pragma unroll
    for (int i = 0; i < N; ++i) ubar[0] = __builtin_fmaf(L[oG + i], lam[i], ubar[0]);
    f32x4 v = splat4(0.f);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) acc = __builtin_fmaf(L[oJ + i * N + j], lam[i], acc);
#pragma unroll
      for (int k = 0; k < N; ++k) acc = __builtin_fmaf(-S[j][k], Stl[k], acc);
      v[j] = acc;
    }
    float scaleH;
    f32x4 PH = hnet_hvp_w(L + oH, ln, tp, v, scaleH);
    xch_put_partial(ln.xch + kXP0, ln, PR);
    xch_put_partial(ln.xch + kXP1, ln, PH);
    __syncthreads();
    f32x4 xb = splat4(0.f);
    xb += xch_sum_partials(ln.xch + kXP0, ln) * unscaleR;
    xbar = xb + xch_sum_partials(ln.xch + kXP1, ln) * scaleH;
  }
};

template <int DUMMY = 0>
struct CanonSplit {  // CanonModel<128, f16x2> with the tile split over four waves
  using Base = CanonModel<128, MM_F16X2>;
  static constexpr int N = 4, HID = 128, T = 8, MM = MM_F16X2;
  static constexpr bool SPLIT = true;
  static constexpr int SCR = 0, IMG = Base::IMG, STASH = Base::STASH, oH = Base::oH, oC = Base::oC, MI = 1;
  DEV static float Base_Gu(const float* L, int row, f32x4 u) { return Base::Base_Gu(L, row, u); }
  DEV static f32x4 Base_Gt(const float* L, float dpb0, float dpb1) { return Base::Base_Gt(L, dpb0, dpb1); }

  template <bool WANT_H, bool ST = false>
  DEV static f32x4 f(const float* L, float* scr, Lane ln, f32x4 y, f32x4 u, float& Hval, float* stash = nullptr) {
    static_assert(!WANT_H, "the split-tile kernels are rollout kernels");
    keep_lds_reads_local();
    float a = L[oC + 0], b = L[oC + 1], c = L[oC + 2];
    float sn, cs;
    sincos_dev(y[1], sn, cs);
    float bc = b * cs;
    f32x4 z = {y[0], y[1], a * y[2] + bc * y[3], bc * y[2] + c * y[3]};
    HTapeW tp;
    f32x4 P = hnet_grad_w(L + oH, ln, z, tp, []() {});
    xch_put_partial(ln.xch + kXP0, ln, P);
    if (ST) {
      store_act<2>(stash + 2 * ln.w * 256, ln, tp.a2);
      store_act<2>(stash + T * 256 + 2 * ln.w * 256, ln, tp.q1);
    }
    __syncthreads();
    f32x4 dH = xch_sum_partials(ln.xch + kXP0, ln);
    if (ST && ln.q == 0 && ln.w == 0) PHNN_NT_STORE(dH, reinterpret_cast<f32x4*>(stash + 2 * T * 256) + ln.i);
    float dp0 = (-dH[0] - L[oC + 6] * dH[2]) + Base_Gu(L, 2, u);
    float dp1 = (-dH[1] - L[oC + 7] * dH[3]) + Base_Gu(L, 3, u);
    float det = (a * c - bc * bc) + 1e-6f;
    float mi00 = c / det, mi01 = -bc / det, mi11 = a / det;
    return f32x4{mi00 * z[2] + mi01 * z[3], mi01 * z[2] + mi11 * z[3], mi00 * dp0 + mi01 * dp1,
                 mi01 * dp0 + mi11 * dp1};
  }

  template <bool ST = false, bool WG = false>
  DEV static void vjp(const float* L, float* scr, Lane ln, f32x4 y, f32x4 u, f32x4 lam, f32x4& ybar, f32x4& ubar,
                      const float* stash = nullptr, float* rec = nullptr, float Hbar = 0.f) {
    static_assert(!WG, "weight-gradient records come from the whole-tile kernels");
    keep_lds_reads_local();
    const int t0 = 2 * ln.w;
    float a = L[oC + 0], b = L[oC + 1], c = L[oC + 2];
    float sn, cs;
    sincos_dev(y[1], sn, cs);
    float bc = b * cs;
    f32x4 z = {y[0], y[1], a * y[2] + bc * y[3], bc * y[2] + c * y[3]};
    HTapeW tp;
    f32x4 dH;
    if (ST) {
      dH = PHNN_NT_LOAD(reinterpret_cast<const f32x4*>(stash + 2 * T * 256) + ln.i);  // needed first: requested first
      load_act<2>(stash + t0 * 256, ln, tp.a2);
      load_act<2>(stash + T * 256 + t0 * 256, ln, tp.q1);
      using Y = LayH2<128, MM_F16X2>;
      load_vec<2>(tp.a1, L + oH + Y::oB1 + 16 * t0, ln);
      in_layer_mm<2, MM_F16X2, kInHRecomp>(tp.a1, L + oH, Y::oW1f, Y::oW1h, ln, z, t0);
      tanh_pre_w(tp.a1);
    } else {
      f32x4 P = hnet_grad_w<kInHRecomp>(L + oH, ln, z, tp, []() {});
      xch_put_partial(ln.xch + kXP0, ln, P);
      __syncthreads();
      dH = xch_sum_partials(ln.xch + kXP0, ln);
    }
    float Rd2 = L[oC + 6], Rd3 = L[oC + 7];
    float dp0 = (-dH[0] - Rd2 * dH[2]) + Base_Gu(L, 2, u);
    float dp1 = (-dH[1] - Rd3 * dH[3]) + Base_Gu(L, 3, u);
    float det = (a * c - bc * bc) + 1e-6f;
    float rdet = 1.0f / det;
    float mi00 = c * rdet, mi01 = -bc * rdet, mi11 = a * rdet;
    float pb0 = lam[0] * mi00 + lam[1] * mi01, pb1 = lam[0] * mi01 + lam[1] * mi11;
    float dpb0 = lam[2] * mi00 + lam[3] * mi01, dpb1 = lam[2] * mi01 + lam[3] * mi11;
    float mb00 = lam[0] * z[2] + lam[2] * dp0;
    float mb01 = lam[0] * z[3] + lam[1] * z[2] + lam[2] * dp1 + lam[3] * dp0;
    float mb11 = lam[1] * z[3] + lam[3] * dp1;
    f32x4 v = {-dpb0, -dpb1, -Rd2 * dpb0, -Rd3 * dpb1};
    ubar = Base_Gt(L, dpb0, dpb1);
    float scaleH;
    f32x4 PH = hnet_hvp_w(L + oH, ln, tp, v, scaleH);
    xch_put_partial(ln.xch + kXP1, ln, PH);
    __syncthreads();
    f32x4 zb = xch_sum_partials(ln.xch + kXP1, ln) * scaleH;
    zb[2] += pb0;
    zb[3] += pb1;
    float bcb = zb[2] * y[3] + zb[3] * y[2];
    float detb = (-(mb00 * c + mb11 * a) + mb01 * bc) * (rdet * rdet);
    bcb += -mb01 * rdet;
    bcb += -2.0f * bc * detb;
    ybar = f32x4{zb[0], zb[1] + bcb * (-b * sn), zb[2] * a + zb[3] * bc, zb[2] * bc + zb[3] * c};
  }
};

// ------------------------------------------------------------------------------------------------
// Model: ODEFunc MLP [x,u] -> HID -> HID -> HID -> n  (src/baseline_node.py:60-116), n + m <= 4
// ------------------------------------------------------------------------------------------------
template <int N_, int HID_, int MM_ = MM_F32, int ACT_ = ACT_TANH>
struct OdeModel {
  static constexpr int N = N_, HID = HID_, T = HID / 16, LD = HID + 4, LR = HID + 8, MM = MM_, MI = 1, ACT = ACT_;
  static_assert(ACT_ == ACT_TANH || MM_ == MM_F32, "SiLU / ReLU run on the all-f32 kernels");
  static constexpr bool SPLIT = false;
  static constexpr int SCR = 0;  // no per-wave LDS scratch needed
  static constexpr int WF = MM == MM_F16X2 ? HfImg<HID>::FLOATS : HID * LD;  // one hidden x hidden image
  static constexpr int oW1f = 0;                   // [T][64]   W1 (HID x (n+m))
  static constexpr int oB1 = oW1f + T * 64;        // [HID]
  static constexpr int oW2 = oB1 + HID;            // f32: [HID][LD]; f16x2: HfImg of S2 * W2
  static constexpr int oB2 = oW2 + WF;             // [HID]  b2 * S2
  static constexpr int oW3 = oB2 + HID;            // like oW2, scale S3
  static constexpr int oB3 = oW3 + WF;             // [HID]  b3 * S3
  static constexpr int oW4r = oB3 + HID;           // [4][LR] rows c = W4[c,:]
  static constexpr int oB4 = oW4r + 4 * LR;        // [4]
  static constexpr int oW4f = oB4 + 4;             // [T][64] fragment image of W4^T (HID x n)
  static constexpr int oW1T = oW4f + T * 64;       // [4][LR] rows c = W1[:,c] / (S2 S3)
  static constexpr int oSC = oW1T + 4 * LR;        // [4] (2 log2 e / S2, 2 log2 e / S3, 0, 0)
  // n + m = 5 (the reference's default cart-pole ODEFunc(4,1)): the control column of W1 gets its own fragment
  // image (second k-step of the first layer) and its own replicated-row image (ubar)
  static constexpr bool WIDE = N + 1 > 4;
  static constexpr int oW1fu = oSC + 4;                    // [T][64]  lane (i, q=0) holds W1[16nt+i][N]
  static constexpr int oW1Tu = oW1fu + (WIDE ? T * 64 : 0);  // [4][LR] row 0 = W1[:,N] / (S2 S3)
  static constexpr int IMG = oW1Tu + (WIDE ? 4 * LR : 0);
  static_assert(MM != MM_BF16X3, "ODEFunc has an f32 and an f16x2 variant");
  static_assert(N <= 4, "state dimension up to 4");

  struct Tape {
    Act<T> a1, a2, a3;  // activations; SiLU / ReLU variants: the PRE-activations z1, z2, z3
  };

  DEV static void layer1(const float* L, Lane ln, f32x4 x, float u, Act<T>& a1) {
    f32x4 in = x;
    if (!WIDE) in[N & 3] = u;
    load_vec<T>(a1, L + oB1, ln);
    in_layer<T>(a1, L + oW1f, ln, sel4(in, ln.q));
    if (WIDE) in_layer<T>(a1, L + oW1fu, ln, ln.q == 0 ? u : 0.f);
    if constexpr (ACT == ACT_TANH) tanh_act<T>(a1);
  }

  // hidden -> hidden layer: o = tanh(b + W a)
  DEV static void hidden(const float* L, Lane ln, int oW, int oB, float c, const Act<T>& a, Act<T>& o) {
    load_vec<T>(o, L + oB, ln);
    if constexpr (ACT != ACT_TANH) {  // a holds z of the previous layer; o becomes z of this one
      Act<T> act;
      act_of<ACT, T>(a, act);
      sq_fwd<T, T>(o, L + oW, ln, act);
      return;
    }
    if (MM == MM_F16X2) {
      Split2<T> sp;
      split_act_h<T>(a, sp);
      sq_fwd_h<T>(o, L + oW, ln, sp);
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.v[t][r] = tanh_scaled(o.v[t][r], c);
    } else {
      sq_fwd<T, T>(o, L + oW, ln, a);
      tanh_act<T>(o);
    }
  }

  // o = W^T d (times the image's scale in f16x2 mode)
  DEV static void hidden_T(const float* L, Lane ln, int oW, const Act<T>& d, Act<T>& o) {
    zero_act<T>(o);
    if (MM == MM_F16X2) {
      Split2<T> sp;
      split_act_h<T>(d, sp);
      sq_bwd_h<T>(o, L + oW, ln, sp);
    } else {
      sq_bwd<T, T>(o, L + oW, ln, d);
    }
  }

  DEV static f32x4 fwd(const float* L, Lane ln, f32x4 x, float u, Tape& tp) {
    keep_lds_reads_local();
    layer1(L, ln, x, u, tp.a1);
    hidden(L, ln, oW2, oB2, L[oSC + 0], tp.a1, tp.a2);
    hidden(L, ln, oW3, oB3, L[oSC + 1], tp.a2, tp.a3);
    f32x4 b4 = *reinterpret_cast<const f32x4*>(L + oB4);
    if constexpr (ACT != ACT_TANH) {
      Act<T> act;
      act_of<ACT, T>(tp.a3, act);
      return to4_rep<T>(L + oW4r, ln, act) + b4;
    }
    return to4_rep<T>(L + oW4r, ln, tp.a3) + b4;
  }

  // floats one wave stashes per step: a2, a3 as 24-bit fixed point (store_act24; a1 is recomputed from (x,u))
  // (SiLU / ReLU: the pre-activations z2, z3 are unbounded: plain float32, T x 256 floats each)
  static constexpr int VEC24 = ACT == ACT_TANH ? T * 192 : T * 256;
  static constexpr int STASH = 2 * VEC24;

  template <bool WANT_H, bool ST = false>
  DEV static f32x4 f(const float* L, float* scr, Lane ln, f32x4 x, f32x4 uv, float& Hval, float* stash = nullptr) {
    const float u = uv[0];
    Tape tp;
    if (WANT_H) Hval = 0.f;
    f32x4 dx = fwd(L, ln, x, u, tp);
    if (ST) {
      if constexpr (ACT == ACT_TANH) {
        store_act24<T>(stash, ln, tp.a2);
        store_act24<T>(stash + VEC24, ln, tp.a3);
      } else {
        store_act<T>(stash, ln, tp.a2);
        store_act<T>(stash + VEC24, ln, tp.a3);
      }
    }
    return dx;
  }

  template <bool ST = false, bool WG = false>
  DEV static void vjp(const float* L, float* scr, Lane ln, f32x4 x, f32x4 uv, f32x4 lam, f32x4& xbar, f32x4& ubar4,
                      const float* stash = nullptr, float* rec = nullptr, float Hbar = 0.f) {
    static_assert(!WG, "ODEFunc has no weight-gradient kernels");
    const float u = uv[0];
    float ubar;
    Tape tp;
    if (ST) {
      if constexpr (ACT == ACT_TANH) {
        load_act24<T>(stash + VEC24, ln, tp.a3);  // the backward sweep meets a3 first
        load_act24<T>(stash, ln, tp.a2);
      } else {
        load_act<T>(stash + VEC24, ln, tp.a3);
        load_act<T>(stash, ln, tp.a2);
      }
      keep_lds_reads_local();
      layer1(L, ln, x, u, tp.a1);
    } else {
      (void)fwd(L, ln, x, u, tp);
    }
    float unscale = 1.0f;
    if (MM == MM_F16X2) {  // the backward chain is linear in lam: normalise by a power of two (exact)
      float mx = fmaxf(fmaxf(__builtin_fabsf(lam[0]), __builtin_fabsf(lam[1])), fmaxf(__builtin_fabsf(lam[2]), __builtin_fabsf(lam[3])));
      int e = 0;
      (void)__builtin_frexpf(mx, &e);
      e = (mx > 0.f && mx < 3.0e38f) ? e : 0;
      lam = lam * __builtin_ldexpf(1.0f, -e);
      unscale = __builtin_ldexpf(1.0f, e);
    }
    Act<T> d, e;
    zero_act<T>(d);
    in_layer<T>(d, L + oW4f, ln, sel4(lam, ln.q));
    if constexpr (ACT != ACT_TANH) {
      mul_d1<ACT, T>(d, tp.a3);
      hidden_T(L, ln, oW3, d, e);
      mul_d1<ACT, T>(e, tp.a2);
      hidden_T(L, ln, oW2, e, d);
      mul_d1<ACT, T>(d, tp.a1);
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) d.v[t] = d.v[t] * dtanh(tp.a3.v[t]);
      hidden_T(L, ln, oW3, d, e);
#pragma unroll
      for (int t = 0; t < T; ++t) e.v[t] = e.v[t] * dtanh(tp.a2.v[t]);
      hidden_T(L, ln, oW2, e, d);
#pragma unroll
      for (int t = 0; t < T; ++t) d.v[t] = d.v[t] * dtanh(tp.a1.v[t]);
    }
    f32x4 inb = to4_rep<T>(L + oW1T, ln, d);
    if (MM == MM_F16X2) inb = inb * unscale;
    if (WIDE) {
      ubar = to4_rep<T>(L + oW1Tu, ln, d)[0] * unscale;
    } else {
      ubar = inb[N & 3];
      inb[N & 3] = 0.f;
    }
    xbar = inb;
    ubar4 = f32x4{ubar, 0.f, 0.f, 0.f};
  }
};

// ------------------------------------------------------------------------------------------------
// stage cost (src/mpc_controller.py:75-114, src/mpc_controller_canonical.py:91-120)
// ------------------------------------------------------------------------------------------------
// The soft state barrier (src/mpc_controller.py:96-107; off in every shipped configuration) sits behind a REAL branch:
// left to if-conversion its ~25 vector instructions ran in every step of every rollout and were masked out afterwards.
// The empty asm has side effects as far as the compiler knows, so the block cannot be speculated.
DEV void no_speculation() { asm volatile("" ::: "memory"); }

template <int N>
DEV float state_cost(const phnn_cost& c, f32x4 x) {
  float e[N], cost = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = x[i] - c.x_target[i];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) s = __builtin_fmaf(e[i], c.Q[i * N + j], s);
    cost = __builtin_fmaf(s, e[j], cost);
  }
  if (c.has_x_min) {
    no_speculation();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float v = fmaxf(c.x_min[i] - x[i], 0.f);
      s = __builtin_fmaf(v, v, s);
    }
    cost = __builtin_fmaf(c.barrier_weight, s, cost);
  }
  if (c.has_x_max) {
    no_speculation();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float v = fmaxf(x[i] - c.x_max[i], 0.f);
      s = __builtin_fmaf(v, v, s);
    }
    cost = __builtin_fmaf(c.barrier_weight, s, cost);
  }
  return cost;
}

// Qs = Q + Q^T from the host (RollParams::Qs; the same float32 sum the kernel used to form per step)
template <int N>
DEV f32x4 state_cost_grad(const phnn_cost& c, const float (&Qs)[16], f32x4 x) {
  float e[N];
  f32x4 g = splat4(0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) e[i] = x[i] - c.x_target[i];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) s = __builtin_fmaf(Qs[i * N + j], e[j], s);
    g[i] = s;
  }
  if (c.has_x_min) {
    no_speculation();
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = __builtin_fmaf(-2.0f * c.barrier_weight, fmaxf(c.x_min[i] - x[i], 0.f), g[i]);
  }
  if (c.has_x_max) {
    no_speculation();
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = __builtin_fmaf(2.0f * c.barrier_weight, fmaxf(x[i] - c.x_max[i], 0.f), g[i]);
  }
  return g;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct RollParams {
  const float* img;    // packed LDS image (global)
  const float* x0;     // (B,N)
  const float* u;      // (B,H)   (m = 1)
  const float* traj_in;  // (B,H+1,N)  K2
  const float* traj_bar;  // (B,H+1,N) or null: cotangent on the trajectory (K2)
  const float* cost_bar;  // (B) or null (= 1): cotangent on the cost (K2)
  float* cost;         // (B)
  float* traj;         // (B,H+1,N) or null
  float* grad_u;       // (B,H)
  float* grad_x0;      // (B,N) or null
  float* stash;        // K1 -> K2 tape workspace (Euler only) or null: [tile][t][M::STASH] floats
  const float* dx_bar;  // (B,H,N) or null: cotangent on the per-step derivatives f(x_t,u_t) (training losses on dX_pred)
  float* dx_out;       // (B,H,N) or null: K1 also returns f(x_t,u_t) of every step (first stage)
  float* wrec;         // weight-gradient records [tile][t][stage][M::Rec::SIZE] (adjoint kernels built with WG)
  int no_cost;         // 1: no stage cost at all (training rollouts): cost cotangent 0, controls used as given
  long long B;
  int H;
  float dt, half_dt, sixth_dt;
  phnn_cost c;
  float Qs[16];        // Q + Q^T (row-major n x n), formed on the host in float32: the adjoint used to rebuild it every step
};

struct PointParams {
  const float* img;
  const float* x;    // (B,N)
  const float* u;    // (B)
  const float* lam;  // (B,N)  vjp only
  float* dx;         // (B,N)  forward: dx ; vjp: xbar
  float* Hout;       // (B)    forward: H (nullable) ; vjp: ubar
  long long B;
  const float* Hbar;  // (B) or null: cotangent on H (WG vjp only)
  float* wrec;        // weight-gradient records [tile][M::Rec::SIZE] (WG vjp only)
};

template <int IMG>
DEV void stage_image(float* lds, const float* img) {
  const f32x4* src = reinterpret_cast<const f32x4*>(img);
  f32x4* dst = reinterpret_cast<f32x4*>(lds);
  for (int k = threadIdx.x; k < IMG / 4; k += blockDim.x) dst[k] = src[k];
  __syncthreads();
}

template <int N>
DEV f32x4 load_state(const float* p) {
  if (N == 4) return *reinterpret_cast<const f32x4*>(p);
  f32x4 x = splat4(0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = p[i];
  return x;
}
template <int N>
DEV void store_state(float* p, f32x4 x) {
  if (N == 4) {
    *reinterpret_cast<f32x4*>(p) = x;
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = x[i];
  }
}

DEV float clamp_u(const phnn_cost& c, float u) { return c.has_u_bounds ? fminf(fmaxf(u, c.u_min), c.u_max) : u; }

// controls of one step: (B,H,MI) row-major, up = this rollout's (H,MI) block
template <int MI>
DEV f32x4 load_u(const float* up, int t) {
  f32x4 u = splat4(0.f);
#pragma unroll
  for (int k = 0; k < MI; ++k) u[k] = up[t * MI + k];
  return u;
}
template <int MI>
DEV f32x4 clamp_u4(const phnn_cost& c, f32x4 u) {
#pragma unroll
  for (int k = 0; k < MI; ++k) u[k] = clamp_u(c, u[k]);
  return u;
}
// u^T R u in the order of ((u @ R) * u).sum() (src/mpc_controller_canonical.py:116-118); m = 1: (u R) u
template <int MI>
DEV float control_cost(const phnn_cost& c, f32x4 u) {
  if (MI == 1) return (u[0] * c.R[0]) * u[0];
  float cost = 0.f;
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) s = __builtin_fmaf(u[i], c.R[i * MI + j], s);
    cost = __builtin_fmaf(s, u[j], cost);
  }
  return cost;
}

// K1 -> K2 stash of one rollout step of one 16-rollout tile.  Euler: the tape of the one dynamics evaluation
// (M::STASH floats).  RK4: four stage slots, each the tape of that stage's evaluation followed by the stage state
// (16 rollouts x float4), so that K2 runs four tape-reading VJPs and no forward evaluation at all.
template <class M, int INTEG>
struct StashStep {
  static constexpr int SLOT = M::STASH + 64;
  static constexpr int FLOATS = INTEG == PHNN_INTEG_EULER ? M::STASH : 4 * SLOT;
};
DEV void store_stage(float* dst, Lane ln, f32x4 y) {  // ln.w: wave within a split tile (0 for whole-tile models)
  if (ln.q == 0 && ln.w == 0) PHNN_NT_STORE(y, reinterpret_cast<f32x4*>(dst) + ln.i);
}
DEV f32x4 load_stage(const float* src, Lane ln) {
  return PHNN_NT_LOAD(reinterpret_cast<const f32x4*>(src) + ln.i);
}

// Lane / tile bookkeeping shared by the march kernels.  One wave = 16 rollouts; split-tile models: the workgroup's four
// waves share tile blockIdx.x (all four reach every barrier together).
struct TileCtx {
  Lane ln;
  float* scr;
  long long tile, b;
  bool valid, writer;
};
template <class M>
DEV bool tile_ctx(TileCtx& c, float* lds, long long B) {  // false: no tile for this wave
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  static_prio(wave);
  c.ln.lane = threadIdx.x & 63;
  c.ln.i = c.ln.lane & 15;
  c.ln.q = c.ln.lane >> 4;
  c.ln.w = M::SPLIT ? wave : 0;
  c.ln.xch = M::SPLIT ? lds + M::IMG + nwaves * M::SCR : nullptr;
  c.scr = lds + M::IMG + wave * M::SCR;
  c.tile = M::SPLIT ? (long long)blockIdx.x : (long long)blockIdx.x * nwaves + wave;
  if (c.tile * kTileB >= B) return false;
  c.b = c.tile * kTileB + c.ln.i;
  c.valid = c.b < B;
  if (!c.valid) c.b = B - 1;
  c.writer = c.valid && c.ln.q == 0 && (!M::SPLIT || wave == 0);
  return true;
}

// K1: forward march of one tile.
template <class M, int INTEG, bool STASH>
DEV void fwd_march(const RollParams& p, const float* L, const TileCtx& tc) {
  constexpr int N = M::N;
  const Lane ln = tc.ln;
  float* scr = tc.scr;
  const long long tile = tc.tile, b = tc.b;
  const bool writer = tc.writer;
  f32x4 x = load_state<N>(p.x0 + b * N);
  if (p.traj && writer) store_state<N>(p.traj + (b * (p.H + 1)) * N, x);
  float cost = state_cost<N>(p.c, x);
  constexpr int MI = M::MI;
  const float* up = p.u + b * p.H * MI;
  float Hd;
  for (int t = 0; t < p.H; ++t) {
    f32x4 u = clamp_u4<MI>(p.c, load_u<MI>(up, t));
    if (MI == 1) cost = __builtin_fmaf(u[0] * p.c.R[0], u[0], cost);
    else cost += control_cost<MI>(p.c, u);
    float* sl = STASH ? p.stash + (tile * p.H + t) * (long long)StashStep<M, INTEG>::FLOATS : nullptr;
    f32x4 k1 = M::template f<false, STASH>(L, scr, ln, x, u, Hd, sl);
    if (p.dx_out && writer) store_state<N>(p.dx_out + (b * p.H + t) * N, k1);
    if (INTEG == PHNN_INTEG_EULER) {
      x = x + p.dt * k1;
    } else {
      // stages 2..4 as a rolled loop (one copy of f in the instruction stream; nothing of one stage is hoisted into
      // another).  x + dt/6 (((k1 + 2 k2) + 2 k3) + k4), the association of src/integrators.py:97-109.
      constexpr int SLOT = StashStep<M, INTEG>::SLOT;  // one stage: the tape of f at the stage state + the stage state
      f32x4 acc = k1, kp = k1;
#pragma unroll 1
      for (int s = 1; s < 4; ++s) {
        const f32x4 y = x + (s == 3 ? p.dt : p.half_dt) * kp;
        float* ss = STASH ? sl + s * SLOT : nullptr;
        if (STASH) store_stage(ss + M::STASH, ln, y);
        kp = M::template f<false, STASH>(L, scr, ln, y, u, Hd, ss);
        acc = acc + (s == 3 ? 1.0f : 2.0f) * kp;
      }
      x = x + p.sixth_dt * acc;
    }
    cost += state_cost<N>(p.c, x);
    if (p.traj && writer) store_state<N>(p.traj + (b * (p.H + 1) + t + 1) * N, x);
  }
  if (writer && p.cost) p.cost[b] = cost;
}

template <class M, int INTEG, bool STASH>
__global__ __launch_bounds__(64 * kMaxWaves) void k_rollout_fwd(RollParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_image<M::IMG>(lds, p.img);
  TileCtx tc;
  if (!tile_ctx<M>(tc, lds, p.B)) return;
  fwd_march<M, INTEG, STASH>(p, lds, tc);
}

// K2: adjoint march over the states K1 stored.  WG: every dynamics VJP also emits its weight-gradient record
// (training side: k_wgrad_reduce sums them into d loss / d theta).
template <class M, int INTEG, bool STASH, bool WG = false>
DEV void grad_march(const RollParams& p, const float* L, const TileCtx& tc) {
  constexpr int N = M::N;
  const Lane ln = tc.ln;
  float* scr = tc.scr;
  const long long tile = tc.tile, b = tc.b;
  const bool valid = tc.valid, writer = tc.writer;
  const float* tr = p.traj_in + (b * (p.H + 1)) * N;
  constexpr int MI = M::MI;
  const float* up = p.u + b * p.H * MI;
  const float cb = p.no_cost ? 0.0f : (p.cost_bar ? p.cost_bar[b] : 1.0f);
  const float* tb = p.traj_bar ? p.traj_bar + (b * (p.H + 1)) * N : nullptr;
  const float* db = p.dx_bar ? p.dx_bar + (b * p.H) * N : nullptr;
  // a padding lane (rollout index beyond the batch) must not contribute to the weight gradient: its cotangents are zeroed
  const float live = valid ? 1.0f : 0.0f;
  f32x4 lam = cb * state_cost_grad<N>(p.c, p.Qs, load_state<N>(tr + (long long)p.H * N));
  if (tb) lam = lam + load_state<N>(tb + (long long)p.H * N);
  if (WG) lam = lam * live;
  constexpr int STAGES = INTEG == PHNN_INTEG_EULER ? 1 : 4;
  float Hd;
  for (int t = p.H - 1; t >= 0; --t) {
    f32x4 x = load_state<N>(tr + (long long)t * N);
    f32x4 uraw = load_u<MI>(up, t);
    f32x4 u = p.no_cost ? uraw : clamp_u4<MI>(p.c, uraw);
    f32x4 dxb = splat4(0.f);
    if (INTEG == PHNN_INTEG_EULER && db) dxb = load_state<N>(db + (long long)t * N) * live;
    f32x4 xb, ub, utot;
    float* rec = nullptr;
    if constexpr (WG) rec = p.wrec + ((tile * p.H + t) * STAGES) * (long long)M::Rec::SIZE;
    if constexpr (INTEG == PHNN_INTEG_EULER) {
      if constexpr (WG) {
        M::template vjp<STASH, true>(L, scr, ln, x, u, p.dt * lam + dxb, xb, ub,
                                     STASH ? p.stash + (tile * p.H + t) * (long long)M::STASH : nullptr, rec);
      } else {
        M::template vjp<STASH>(L, scr, ln, x, u, p.dt * lam + dxb, xb, ub,
                               STASH ? p.stash + (tile * p.H + t) * (long long)M::STASH : nullptr);
      }
      lam = lam + xb;
      utot = ub;
    } else {
      // RK4, stages as rolled loops (one copy of f / vjp in the instruction stream, no cross-stage hoisting of tape
      // loads: that is what made the unrolled form spill).  STASH: the stage states and tapes come from K1's stash.
      constexpr int SLOT = StashStep<M, INTEG>::SLOT;
      const float* sl = STASH ? p.stash + (tile * p.H + t) * (long long)StashStep<M, INTEG>::FLOATS : nullptr;
      f32x4 y2 = x, y3 = x, y4 = x;
      if constexpr (!STASH) {
        f32x4 y = x;
#pragma unroll 1
        for (int s = 0; s < 3; ++s) {
          const f32x4 k = M::template f<false>(L, scr, ln, y, u, Hd);
          y = x + (s == 2 ? p.dt : p.half_dt) * k;
          y2 = s == 0 ? y : y2;
          y3 = s == 1 ? y : y3;
          y4 = s == 2 ? y : y4;
        }
      }
      // Registers are full inside vjp: what rides through the four calls is kept to lam, the running sums and (in
      // recompute mode) the stage states; the step's state, controls and dX cotangent are re-read where a stage
      // needs them (L2 hits; vjp opens with a compiler memory barrier, so the loads stay where they are written).
      f32x4 ysum = splat4(0.f), ybn = ysum;
      utot = splat4(0.f);
#pragma unroll 1
      for (int s = 3; s >= 0; --s) {
        f32x4 y;
        if constexpr (STASH) y = s == 0 ? load_state<N>(tr + (long long)t * N) : load_stage(sl + s * SLOT + M::STASH, ln);
        else y = s == 0 ? load_state<N>(tr + (long long)t * N) : (s == 1 ? y2 : (s == 2 ? y3 : y4));
        const f32x4 us = p.no_cost ? load_u<MI>(up, t) : clamp_u4<MI>(p.c, load_u<MI>(up, t));
        // cotangent on k_s: dt/6 (1,2,2,1) lam + the next stage's state cotangent times its step factor
        f32x4 in = ((s == 0 || s == 3) ? p.sixth_dt : 2.0f * p.sixth_dt) * lam;
        if (s != 3) in = in + (s == 2 ? p.dt : p.half_dt) * ybn;
        if (s == 0 && db) in = in + load_state<N>(db + (long long)t * N) * live;
        f32x4 yb;
        if constexpr (WG) M::template vjp<STASH, true>(L, scr, ln, y, us, in, yb, ub, STASH ? sl + s * SLOT : nullptr, rec + s * M::Rec::SIZE);
        else M::template vjp<STASH>(L, scr, ln, y, us, in, yb, ub, STASH ? sl + s * SLOT : nullptr);
        utot = s == 3 ? ub : utot + ub;
        ybn = yb;
        ysum = s == 3 ? yb : ysum + yb;  // ((yb4 + yb3) + yb2) + yb1
      }
      lam = lam + ysum;
      x = load_state<N>(tr + (long long)t * N);
      uraw = load_u<MI>(up, t);
      u = p.no_cost ? uraw : clamp_u4<MI>(p.c, uraw);
    }
    lam = lam + cb * state_cost_grad<N>(p.c, p.Qs, x);
    if (tb) lam = lam + load_state<N>(tb + (long long)t * N) * (WG ? live : 1.0f);
#pragma unroll
    for (int k = 0; k < MI; ++k) {
      float g;
      if (MI == 1) {
        g = __builtin_fmaf(cb * (2.0f * p.c.R[0]), u[0], utot[0]);
      } else {  // d(u^T R u)/du_k = sum_j (R[k][j] + R[j][k]) u_j
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < MI; ++j) s = __builtin_fmaf(p.c.R[k * MI + j] + p.c.R[j * MI + k], u[j], s);
        g = __builtin_fmaf(cb, s, utot[k]);
      }
      if (!p.no_cost && p.c.has_u_bounds && !(uraw[k] >= p.c.u_min && uraw[k] <= p.c.u_max)) g = 0.f;
      if (writer && p.grad_u) p.grad_u[(b * p.H + t) * MI + k] = g;
    }
  }
  if (p.grad_x0 && writer) store_state<N>(p.grad_x0 + b * N, lam);
}

template <class M, int INTEG, bool STASH, bool WG = false>
__global__ __launch_bounds__(64 * kMaxWaves) void k_rollout_grad(RollParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_image<M::IMG>(lds, p.img);
  TileCtx tc;
  if (!tile_ctx<M>(tc, lds, p.B)) return;
  grad_march<M, INTEG, STASH, WG>(p, lds, tc);
}

// (Measured, round 3, and removed: a fused solve kernel -- `iters` x (forward march, adjoint march, Adam step) for one
// tile inside one workgroup, one launch instead of 3 x iters, the image staged once.  Bit-identical results, but a
// single-plant solve took 3.54 ms against 3.48 ms as separate launches: back-to-back launches on one stream are already
// pipelined, the solve is the serial chain of its time steps and nothing else.  fwd_march / grad_march stay factored out.)

// model(x,u) -> (dx, H)
template <class M>
__global__ __launch_bounds__(64 * kMaxWaves) void k_model_forward(PointParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int N = M::N;
  stage_image<M::IMG>(lds, p.img);
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  Lane ln;
  ln.lane = threadIdx.x & 63;
  ln.i = ln.lane & 15;
  ln.q = ln.lane >> 4;
  ln.w = 0;
  ln.xch = nullptr;
  float* scr = lds + M::IMG + wave * M::SCR;
  const long long ntiles = (p.B + kTileB - 1) / kTileB;
  for (long long tile = (long long)blockIdx.x * nwaves + wave; tile < ntiles; tile += (long long)gridDim.x * nwaves) {
    long long b = tile * kTileB + ln.i;
    const bool valid = b < p.B;
    if (!valid) b = p.B - 1;
    f32x4 x = load_state<N>(p.x + b * N);
    float Hval = 0.f;
    f32x4 dx = M::template f<true>(lds, scr, ln, x, load_u<M::MI>(p.u + b * M::MI, 0), Hval);
    if (valid && ln.q == 0) {
      store_state<N>(p.dx + b * N, dx);
      if (p.Hout) p.Hout[b] = Hval;
    }
  }
}

template <class M, bool WG = false>
__global__ __launch_bounds__(64 * kMaxWaves) void k_model_vjp(PointParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int N = M::N;
  stage_image<M::IMG>(lds, p.img);
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  Lane ln;
  ln.lane = threadIdx.x & 63;
  ln.i = ln.lane & 15;
  ln.q = ln.lane >> 4;
  ln.w = 0;
  ln.xch = nullptr;
  float* scr = lds + M::IMG + wave * M::SCR;
  const long long ntiles = (p.B + kTileB - 1) / kTileB;
  for (long long tile = (long long)blockIdx.x * nwaves + wave; tile < ntiles; tile += (long long)gridDim.x * nwaves) {
    long long b = tile * kTileB + ln.i;
    const bool valid = b < p.B;
    if (!valid) b = p.B - 1;
    f32x4 x = load_state<N>(p.x + b * N);
    f32x4 lam = load_state<N>(p.lam + b * N);
    f32x4 xb, ub;
    const f32x4 u = load_u<M::MI>(p.u + b * M::MI, 0);
    if constexpr (WG) {  // padding lanes carry zero cotangents: they add nothing to the weight gradient
      const float live = valid ? 1.0f : 0.0f;
      M::template vjp<false, true>(lds, scr, ln, x, u, lam * live, xb, ub, nullptr,
                                   p.wrec + tile * (long long)M::Rec::SIZE, p.Hbar ? p.Hbar[b] * live : 0.0f);
    } else {
      M::vjp(lds, scr, ln, x, u, lam, xb, ub);
    }
    if (valid && ln.q == 0) {
      store_state<N>(p.dx + b * N, xb);
#pragma unroll
      for (int k = 0; k < M::MI; ++k) p.Hout[b * M::MI + k] = ub[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient reduction (training side): records -> d loss / d theta.
//
// Each gradient is a sum over evaluation points of outer products of per-point vectors (oracle/phnn_oracle.c
// hnet_wgrad / mlp_wgrad state the algebra):
//   W2bar = sum_p  gdot2* (x) a1 + g2 (x) adot1          b2bar = sum gdot2*        W3bar = sum adot2*
//   W1bar = sum_p  gdot1* (x) x  + g1 (x) v              b1bar = sum gdot1*        b3bar = sum Hbar
//   (x* = x + Hbar-weighted value term;  v = A^T lam;  dots = tangents along v, all rebuilt from the record)
//   R_net: V2bar = sum rbar (x) hR, c2bar = sum rbar, V1bar = sum hbR (x) x, c1bar = sum hbR     (G_net alike)
//   Jbar  = sum lam dH^T - dH lam^T
// Mapping: one workgroup = T waves, wave w owns hidden units 16w..16w+15 of every vector, so all element-wise
// algebra and every vector-shaped sum is lane-local (lane (i,q) reg r = unit 16w+4q+r of rollout i); sums over the 16
// rollouts of a record are taken once, at the end.  The one matrix-shaped gradient, W2bar, is a GEMM with K = points:
// the factor vectors are transposed through LDS (rollout index -> MFMA k index) and accumulated with exact-f32
// v_mfma_f32_16x16x4_f32, wave w producing rows 16w.. of W2bar for all 16T columns (T tiles = 4T registers).
// Workgroups stride over the records; each writes its partial gradient to its own slab row (padded-blob layout),
// k_wgrad_finish sums the rows in a fixed order: bitwise reproducible, no atomics.
// ------------------------------------------------------------------------------------------------
struct WgradParams {
  const float* img;
  const float* rec;   // [n_rec][Rec::SIZE]
  long long n_rec;
  float* slab;        // [gridDim.x][PP]
  int PP;
  const float* tapes;  // TAPES kernels: K1's stash, slot of record r at tapes + r * tape_stride (a2, q1 lead the slot)
  int tape_stride;
};

template <int T>
DEV f32x4 tanh4_model(f32x4 z) {  // the model's own tanh of a pre-activation (tanh_act_pre)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (kPreScaled<T>) z[r] = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z[r]) + 1.0f), 1.0f);
    else z[r] = tanh_dev<true>(z[r]);
  }
  return z;
}

DEV float sum16(float v) {  // over the 16 rollout lanes (i) that share a q
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

// offsets of the padded weight blob (include/phnn_mpc.h order) -- what a slab row is laid out as
template <int N, int HID>
struct BlobMlp1 {  // in(N) -> HID -> OUT
  int oV1, oC1, oV2, oC2, size;
  __host__ __device__ constexpr BlobMlp1(int base, int OUT)
      : oV1(base), oC1(base + HID * N), oV2(base + HID * N + HID), oC2(base + HID * N + HID + OUT * HID),
        size(HID * N + HID + OUT * HID + OUT) {}
};
template <int N, int HID>
struct BlobH {  // in(N) -> HID -> HID -> 1
  int oW1, oB1, oW2, oB2, oW3, oB3, size;
  __host__ __device__ constexpr BlobH(int base)
      : oW1(base), oB1(base + HID * N), oW2(base + HID * N + HID), oB2(base + HID * N + HID + HID * HID),
        oW3(base + HID * N + 2 * HID + HID * HID), oB3(base + HID * N + 3 * HID + HID * HID),
        size(HID * N + 3 * HID + HID * HID + 1) {}
};

template <class M>
struct BlobOf;
template <int N, int HID, bool FIXG, int MM, int MI>
struct BlobOf<PhnnModel<N, HID, FIXG, MM, MI>> {
  static constexpr int oJ = 0, oGfix = N * N;
  static constexpr BlobMlp1<N, HID> R{N * N + (FIXG ? N * MI : 0), N * N};
  static constexpr BlobH<N, HID> H{R.oV1 + R.size};
  static constexpr BlobMlp1<N, HID> G{H.oW1 + H.size, N * MI};
  static constexpr int SIZE = H.oW1 + H.size + (FIXG ? 0 : G.size);
};
template <int HID, int MM, int MI>
struct BlobOf<CanonModel<HID, MM, MI, MASS_CARTPOLE>> {
  static constexpr int oRd = 0;  // R_diag_raw (4) | G (4 MI) | log_a, b, log_c | H_net
  static constexpr BlobH<4, HID> H{4 + 4 * MI + 3};
  static constexpr int SIZE = H.oW1 + H.size;
};

// MassMatrixNetwork variants: the mass block (L_tril, or the 64-wide padded M_net.mlp) sits where log_a, b, log_c do;
// the kernels leave it untouched (its gradient comes from the module's own autograd pass over the recorded points)
template <int HID, int MM, int MI, int MT>
struct BlobOf<CanonModel<HID, MM, MI, MT>> {
  static constexpr int oRd = 0;
  static constexpr int MLP_OUT = MT == MASS_DIAGONAL ? 2 : 3;
  static constexpr int MASS = MT == MASS_CONSTANT ? 4 : (64 * 2 + 64) + (64 * 64 + 64) + (MLP_OUT * 64 + MLP_OUT);
  static constexpr BlobH<4, HID> H{4 + 4 * MI + MASS};
  static constexpr int SIZE = H.oW1 + H.size;
};

template <class M>
struct IsPhnn { static constexpr bool value = false, gnet = false; };
template <int N, int HID, bool FIXG, int MM, int MI>
struct IsPhnn<PhnnModel<N, HID, FIXG, MM, MI>> { static constexpr bool value = true, gnet = !FIXG; };

// geometry of k_wgrad_reduce (shared with the host's LDS size).  Staged constants: only the small vectors of each net
// (H_net: W1 fragments, b1 ... b3; R_net / G_net: V1 fragments, c1, c2, scale).  Exchange buffer (floats).
template <class M>
struct WgGeom {
  static constexpr bool PHNN = IsPhnn<M>::value;
  static constexpr bool GNET = IsPhnn<M>::gnet;
  using YH = LayH2<M::HID, M::MM>;
  using Y1 = LayH1<M::HID, M::MM>;
  static constexpr int H0 = YH::oW1f, HN = YH::SIZE - YH::oW1f;          // staged range of the H_net image
  static constexpr int R0 = Y1::oV1f, RN = Y1::oSc + 4 - Y1::oV1f;        // staged range of an R_net / G_net image
  static constexpr int KEEP = HN + (PHNN ? RN : 0) + (GNET ? RN : 0);
  static constexpr int PARTW = M::HID * 16;                               // dwords per bf16 part: [HID rows][32 k] bf16
  static constexpr int LDH = (PHNN ? 16 : 0) + (GNET ? 16 : 0) + 4;        // floats per row of the hidden-activation array
  static constexpr int oXA = 3 * PARTW, oXH = 6 * PARTW, oXO = oXH + M::HID * LDH;
  static constexpr int XCH = oXO + 32 * 20;
  static constexpr int LDS_FLOATS = KEEP + 2 * XCH;
};

// f32 pair -> three packed bf16 pairs (hi, mid, lo; x = hi + mid + lo to 2^-24), low half = first value
DEV void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pk_bf16(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = pk_bf16(r0, r1);
  const float t0 = r0 - __builtin_bit_cast(float, m << 16), t1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = pk_bf16(t0, t1);
}

// A fragments of V2^T for the reduction's recomputation of hb (f16x2 images only): lane (i,q) of wave w gets
// V2[o = 4q + s][unit 16w + i], s = 0..3, rebuilt from the hi/lo halves of the transposed image (exact to 2^-22).
template <int HID, int MM>
DEV f32x4 v2_frag(const float* img_net, int w, Lane ln) {
  using Y = LayH1<HID, MM>;
  const _Float16* bw = reinterpret_cast<const _Float16*>(img_net + Y::oV2T);
  const int at = (16 * w + ln.i) * 16 + 4 * ln.q;
  const float inv = img_net[Y::oSc];
  f32x4 v;
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) v[s2] = ((float)bw[at + s2] + (float)bw[Y::BPART / 2 + at + s2]) * inv;
  return v;
}

// one-hidden-layer net (R_net / G_net) part of a record, for the units of this lane: the vector-shaped sums.  (The
// output layer's matrix-shaped gradient V2bar = sum obar (x) h goes through the MFMA path with W2bar.)
template <int N, int HID, int MM, int NOUT>
struct H1Acc {
  f32x4 c1 = {0, 0, 0, 0};
  f32x4 V1[4] = {};  // [c] : units on the vector
  f32x4 V2 = {0, 0, 0, 0};  // MFMA accumulator: rows o = 4q + r, column = this lane's unit 16w + i
  f32x4 c2 = {0, 0, 0, 0};  // outputs 4q .. 4q+3 of this lane's rollout

  // hidden activation of this lane's 4 units (recomputed from x); accumulates c1bar, V1bar, c2bar.  ov = the output
  // cotangents 4q .. 4q+3 of this lane's rollout (the values that also go to the exchange as A operand rows).
  // hb = (V2^T obar) (1 - h^2) of these units: from the record (RECHB models), else recomputed -- v2 = this lane's
  // A fragments of V2^T (v2[s] = V2[o = 4q + s][unit 16w + i], see v2_frag), B = ov: k-step s, k-slot q <-> o = 4q + s.
  template <bool RECOMPUTE>
  DEV f32x4 add(const float* Lh1, int w, Lane ln, f32x4 x, f32x4 hb, f32x4 ov, f32x4 v2) {
    using Y = LayH1<HID, MM>;
    f32x4 c = *reinterpret_cast<const f32x4*>(Lh1 + Y::oC1 + 16 * w + 4 * ln.q);
    f32x4 h = tanh4_model<HID / 16>(mfma(Lh1[Y::oV1f + w * 64 + ln.lane], sel4(x, ln.q), c));
    if constexpr (RECOMPUTE) {
      f32x4 d = splat4(0.f);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) d = mfma(v2[s2], ov[s2], d);
      hb = d * dtanh(h);
    }
    c1 += hb;
#pragma unroll
    for (int k = 0; k < N; ++k) V1[k] = fma4(hb, splat4(x[k]), V1[k]);
    c2 += ov;
    return h;
  }

  DEV void write(float* row, const BlobMlp1<N, HID>& B, int w, Lane ln) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = 16 * w + 4 * ln.q + r;
      float s = sum16(c1[r]);
      if (ln.i == 0) row[B.oC1 + unit] = s;
#pragma unroll
      for (int k = 0; k < N; ++k) {
        float t = sum16(V1[k][r]);
        if (ln.i == 0) row[B.oV1 + unit * N + k] = t;
      }
      const int o = 4 * ln.q + r;  // V2bar[o][16w + i]
      if (o < NOUT) row[B.oV2 + o * HID + 16 * w + ln.i] = V2[r];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = sum16(c2[e]);
      if (w == 0 && ln.i == 0 && 4 * ln.q + e < NOUT) row[B.oC2 + 4 * ln.q + e] = t;
    }
  }
};

// one wave's slice of a record: its unit tile of every big vector + what its lane needs of the small vectors of its
// rollout: x, v, lam, dH, the quarter 4q .. 4q+3 of rbar, u and Hbar (25 dwords instead of the block's 40)
template <int NB>
struct RecSlice {
  f32x4 big[NB], x, v, lam, dH, rv, uu;
  float Hbar;
  // tape != null: a2, q1 (big vectors 0, 1) come from K1's stash slot of this record
  DEV void load(const float* R, int vec4, int oSmall, int w, Lane ln, const float* tape = nullptr) {
    const f32x4* bg = reinterpret_cast<const f32x4*>(R) + w * 64 + ln.lane;
    const f32x4* tg = reinterpret_cast<const f32x4*>(tape) + w * 64 + ln.lane;
#pragma unroll
    for (int k = 0; k < NB; ++k) big[k] = (tape && k < 2) ? PHNN_REC_LOAD(tg + k * vec4) : PHNN_REC_LOAD(bg + k * vec4);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(R + oSmall + ln.i * kRecSmall);
    x = s4[0];
    v = s4[1];
    lam = s4[2];
    dH = s4[3];
    rv = s4[4 + ln.q];
    uu = s4[8];
    Hbar = R[oSmall + ln.i * kRecSmall + 36];
  }
};

template <class M, bool TAPES = false>
__global__ __launch_bounds__(64 * M::T) void k_wgrad_reduce(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int N = M::N, HID = M::HID, T = M::T, MM = M::MM;
  using Y = LayH2<HID, MM>;
  using Rec = typename M::Rec;
  using BL = BlobOf<M>;
  using GE = WgGeom<M>;
  constexpr bool PHNN = GE::PHNN, GNET = GE::GNET;
  static_assert(M::oH == 0 && Y::oW2 == 0, "the W2 image leads the model image");
  // stage the small constants of each net (the 128 x 128 images are not needed here)
  for (int k = threadIdx.x; k < GE::HN; k += blockDim.x) lds[k] = p.img[M::oH + GE::H0 + k];
  if constexpr (PHNN) {
    for (int k = threadIdx.x; k < GE::RN; k += blockDim.x) lds[GE::HN + k] = p.img[M::oR + GE::R0 + k];
    if constexpr (GNET)
      for (int k = threadIdx.x; k < GE::RN; k += blockDim.x) lds[GE::HN + GE::RN + k] = p.img[M::oGn + GE::R0 + k];
  }
  const float* L = lds - GE::H0;                       // L[Y::o...] valid for the staged H_net range
  const float* LR = lds + GE::HN - GE::R0;             // LR[LayH1::o...] valid for the staged R_net range
  const float* LG = lds + GE::HN + GE::RN - GE::R0;    // G_net
  const int w = threadIdx.x >> 6;
  Lane ln;
  ln.lane = threadIdx.x & 63;
  ln.i = ln.lane & 15;
  ln.q = ln.lane >> 4;
  ln.w = 0;
  ln.xch = nullptr;
  // Exchange buffers (x2, ping-pong).  The matrix-shaped gradient W2bar = sum_points gdot2* (x) a1 + g2 (x) adot1 is a
  // GEMM with K = points; per record K = 32 (16 rollouts x 2 terms), one v_mfma_f32_16x16x32_bf16 k-depth.  The
  // factors go through LDS as three bf16 pieces each (x = hi + mid + lo to 2^-24; bf16 keeps the f32 exponent range,
  // cotangents can be arbitrarily small) and every 16 x 16 output tile takes the six significant cross products:
  //   XG [3 parts][HID rows][32 k] bf16   k = 2 rollout + term: (gdot2*, g2)      A operand (rows of this wave)
  //   XA [3 parts][HID rows][32 k] bf16   (a1, adot1)                               B operand (all rows)
  //   XH [HID rows][16 (+16) k] f32       hR [| hG], k at position (k & 3) * 4 + (k >> 2)   B operands of the V2bars
  //   XO [32 rows][16 k] f32              rbar (rows 0..15) | gbar (rows 16..31)            A operands of the V2bars
  // Bank layout of the bf16 arrays (64-byte rows, no padding): rows of a 16-row group are stored permuted, row 4q + r
  // at slot s = 4r + q, and the 16-byte chunk c of a row at chunk c ^ ((s >> 2) & 3).  Writes (one dword per lane):
  // the four q-lanes of a rollout hit four adjacent rows = all 64 banks once.  Reads (ds_read_b128, 16 lanes per
  // pass): the 16 rows of a pass fall on 16 different 16-byte bank groups.  (Plain 64-byte rows: 4-way conflicts on
  // one side or the other -- measured 1.4 ms of the kernel's 3.4.)
  constexpr int LDH = GE::LDH, LDO = 20, PARTW = GE::PARTW;
  float* X = lds + GE::KEEP;
  __syncthreads();
  const float S = 2.8853900817779268f / L[Y::oB3 + 1], k1inv = L[Y::oB3 + 2], Sb = L[Y::oB3 + 3];
  const float c_ad2 = k1inv / S, c_q1 = 1.0f / (S * Sb), c_qd = -2.0f * k1inv / (S * Sb);
  const f32x4 b1v = *reinterpret_cast<const f32x4*>(L + Y::oB1 + 16 * w + 4 * ln.q);
  const f32x4 w3v = *reinterpret_cast<const f32x4*>(L + Y::oW3 + 16 * w + 4 * ln.q);
  const float w1f = L[Y::oW1f + w * 64 + ln.lane];
  constexpr bool HBREC = PHNN && !LayH1<HID, MM>::HF;  // hbR / hbG come with the record; else they are recomputed
  f32x4 v2R = splat4(0.f), v2G = splat4(0.f);
  if constexpr (PHNN && !HBREC) {
    v2R = v2_frag<HID, MM>(p.img + M::oR, w, ln);
    if constexpr (GNET) v2G = v2_frag<HID, MM>(p.img + M::oGn, w, ln);
  }

  f32x4 accW2[T];
#pragma unroll
  for (int t = 0; t < T; ++t) accW2[t] = splat4(0.f);
  f32x4 aW3 = splat4(0.f), aB2 = splat4(0.f), aB1 = splat4(0.f), aW1[4] = {};
  float aB3 = 0.f, aJ[N] = {}, aRd[2] = {};
  H1Acc<N, HID, MM, PHNN ? N * N : 1> accR;
  H1Acc<N, HID, MM, N * M::MI> accG;

  // Records are streamed once.  Per record: (1) "stage" -- the element-wise algebra of this wave's units (vector-shaped
  // sums stay in registers) and the transposition of the GEMM factors into exchange buffer `buf`; workgroup barrier;
  // (2) "gemm" -- the matrix-shaped sums from that buffer.  The loop is software-pipelined: an iteration runs gemm of
  // the previous record and stage of the current one (different buffers, no dependence, one basic block -- MFMA / LDS
  // reads and VALU overlap), then the barrier; the slice of the record after that (NBIG + 10 float4 per lane) is
  // already in flight, so its HBM round trip is covered too.
  constexpr int NB = Rec::oSmall / Rec::VEC;
  const int p16 = (ln.i & 3) * 4 + (ln.i >> 2);  // position of k = rollout i in a 16-k segment
  auto stage = [&](const RecSlice<NB>& cur, int buf) {
    const f32x4 a2 = cur.big[0], q1r = cur.big[1], ad2r = cur.big[2], qdr = cur.big[3];
    const f32x4 x = cur.x, v = cur.v, lam = cur.lam, dH = cur.dH, uu = cur.uu;
    const float Hbar = cur.Hbar;
    // H_net factors of this lane's 4 units
    const f32x4 a1 = tanh4_model<T>(mfma(w1f, sel4(x, ln.q), b1v));
    const f32x4 d1 = dtanh(a1), d2 = dtanh(a2);
    const f32x4 ad1 = d1 * mfma(w1f, sel4(v, ln.q), splat4(0.f)) * k1inv;
    const f32x4 ad2 = ad2r * c_ad2;
    const f32x4 g2 = w3v * d2;
    const f32x4 gd2 = fma4(-2.0f * w3v, a2 * ad2, Hbar * g2);
    const f32x4 q1 = q1r * c_q1;
    const f32x4 g1 = q1 * d1;
    const f32x4 gd1 = fma4(qdr * c_qd, d1, fma4(-2.0f * q1, a1 * ad1, Hbar * g1));
    aW3 += fma4(splat4(Hbar), a2, ad2);
    aB2 += gd2;
    aB1 += gd1;
#pragma unroll
    for (int k = 0; k < N; ++k) aW1[k] = fma4(gd1, splat4(x[k]), fma4(g1, splat4(v[k]), aW1[k]));
    unsigned* XG = reinterpret_cast<unsigned*>(X + buf * GE::XCH);
    unsigned* XA = XG + GE::oXA;
    float* XH = X + buf * GE::XCH + GE::oXH;
    float* XO = X + buf * GE::XCH + GE::oXO;
    f32x4 hR = splat4(0.f), hG = splat4(0.f);
    // per-rollout sums: every wave forms them (no branch in the loop body), wave 0 writes them at the end
    aB3 += Hbar;
    if (PHNN) {  // lane (i,q) keeps row q of Jbar
      const float lq = sel4(lam, ln.q), hq = sel4(dH, ln.q);
#pragma unroll
      for (int j = 0; j < N; ++j) aJ[j] = __builtin_fmaf(lq, dH[j], __builtin_fmaf(-lam[j], hq, aJ[j]));
    } else {
      aRd[0] += cur.rv[2];  // lanes q = 0 hold small vector 4 (the R_diag cotangents); only lane 0's sum is written
      aRd[1] += cur.rv[3];
    }
    if constexpr (PHNN) {
      // output cotangents 4q .. 4q+3 of this lane's rollout: rbar = small vectors 4.., gbar[i][k] = lam[i] u[k]
      // (every wave writes the same values to XO: identical stores, no branch)
      f32x4 gv = splat4(0.f);
      const f32x4 rv = (4 * ln.q < N * N) ? cur.rv : splat4(0.f);
      hR = accR.template add<!HBREC>(LR, w, ln, x, HBREC ? cur.big[NB - (GNET ? 2 : 1)] : splat4(0.f), rv, v2R);
#pragma unroll
      for (int e = 0; e < 4; ++e) XO[(4 * ln.q + e) * LDO + p16] = rv[e];
      if constexpr (GNET) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < M::MI; ++k) gv[e] = (4 * ln.q + e == i * M::MI + k) ? lam[i] * uu[k] : gv[e];
        hG = accG.template add<!HBREC>(LG, w, ln, x, HBREC ? cur.big[NB - 1] : splat4(0.f), gv, v2G);
#pragma unroll
        for (int e = 0; e < 4; ++e) XO[(16 + 4 * ln.q + e) * LDO + p16] = gv[e];
      }
    }
    // transpose the factors through LDS: rollout index -> k index
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      // unit 16w + 4q + rr, k = 2i, 2i + 1: row slot 4rr + q, 16-byte chunk (i >> 2) ^ rr
      const int slot = (16 * w + 4 * rr + ln.q) * 16 + ((((ln.i >> 2) ^ rr) << 2) | (ln.i & 3));
      unsigned h, m, l;
      split3_pair(gd2[rr], g2[rr], h, m, l);
      XG[slot] = h;
      XG[PARTW + slot] = m;
      XG[2 * PARTW + slot] = l;
      split3_pair(a1[rr], ad1[rr], h, m, l);
      XA[slot] = h;
      XA[PARTW + slot] = m;
      XA[2 * PARTW + slot] = l;
      const int unit = 16 * w + 4 * ln.q + rr;
      if (PHNN) XH[unit * LDH + p16] = hR[rr];
      if (GNET) XH[unit * LDH + 16 + p16] = hG[rr];
    }
  };
  auto gemm = [&](int buf) {
    const unsigned* XG = reinterpret_cast<const unsigned*>(X + buf * GE::XCH);
    const unsigned* XA = XG + GE::oXA;
    const float* XH = X + buf * GE::XCH + GE::oXH;
    const float* XO = X + buf * GE::XCH + GE::oXO;
    // rows 16w.. of W2bar: A = XG[16w + i][8q .. 8q+7], B = XA[16nt + i][8q .. 8q+7]
    const int rslot = ((ln.i & 3) * 4 + (ln.i >> 2)) * 16 + ((ln.q ^ (ln.i & 3)) << 2);  // row i of a group, chunk of k = 8q
    bf16x8 ag[3];
#pragma unroll
    for (int part = 0; part < 3; ++part)
      ag[part] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(XG + part * PARTW + 16 * w * 16 + rslot));
    // column tiles in pairs, the next pair's fragments requested before this pair's MFMAs (LDS latency, not LDS
    // bandwidth, is what this phase waits on); the two tiles of a pair alternate so that dependent MFMAs are two apart
    auto bfrag = [&](int nt, bf16x8 (&bb)[3]) {
#pragma unroll
      for (int part = 0; part < 3; ++part)
        bb[part] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(XA + part * PARTW + 16 * nt * 16 + rslot));
    };
    bf16x8 b0[3], b1[3];
    bfrag(0, b0);
    bfrag(1, b1);
#pragma unroll
    for (int np = 0; np < T; np += 2) {
      bf16x8 c0[3], c1[3];
      if (np + 2 < T) {
        bfrag(np + 2, c0);
        bfrag(np + 3, c1);
      }
      f32x4 o0 = accW2[np], o1 = accW2[np + 1];  // smallest terms first
      o0 = mfma_bf(ag[2], b0[0], o0);
      o1 = mfma_bf(ag[2], b1[0], o1);
      o0 = mfma_bf(ag[0], b0[2], o0);
      o1 = mfma_bf(ag[0], b1[2], o1);
      o0 = mfma_bf(ag[1], b0[1], o0);
      o1 = mfma_bf(ag[1], b1[1], o1);
      o0 = mfma_bf(ag[1], b0[0], o0);
      o1 = mfma_bf(ag[1], b1[0], o1);
      o0 = mfma_bf(ag[0], b0[1], o0);
      o1 = mfma_bf(ag[0], b1[1], o1);
      o0 = mfma_bf(ag[0], b0[0], o0);
      o1 = mfma_bf(ag[0], b1[0], o1);
      accW2[np] = o0;
      accW2[np + 1] = o1;
      if (np + 2 < T) {
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          b0[part] = c0[part];
          b1[part] = c1[part];
        }
      }
    }
    if constexpr (PHNN) {  // V2bar[o][16w + i] += sum_k obar[o][k] h[16w + i][k]   (k-step s <-> k = 4s + q <-> position 4q + s)
      const f32x4 oa = *reinterpret_cast<const f32x4*>(XO + ln.i * LDO + 4 * ln.q);
      const f32x4 hb4 = *reinterpret_cast<const f32x4*>(XH + (16 * w + ln.i) * LDH + 4 * ln.q);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) accR.V2 = mfma(oa[s2], hb4[s2], accR.V2);
      if constexpr (GNET) {
        const f32x4 og = *reinterpret_cast<const f32x4*>(XO + (16 + ln.i) * LDO + 4 * ln.q);
        const f32x4 hg4 = *reinterpret_cast<const f32x4*>(XH + (16 * w + ln.i) * LDH + 16 + 4 * ln.q);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) accG.V2 = mfma(og[s2], hg4[s2], accG.V2);
      }
    }
  };
  const long long G = gridDim.x, last = p.n_rec - 1;
  // PHNN_WG_DEBUG_{CACHED,NOGEMM,NOSTAGE,NOBARRIER}: timing diagnostics only (wrong results) -- the record stream served
  // from L2, the loop without its GEMM phase / element-wise stage / barrier: profiles/r03_wgrad_reduce_diagnostics.txt
#ifdef PHNN_WG_DEBUG_CACHED
  auto slice_of = [&](long long r) { return p.rec + ((r < last ? r : last) & 255) * Rec::SIZE; };
  auto tape_of = [&](long long r) { return TAPES ? p.tapes + ((r < last ? r : last) & 255) * (long long)p.tape_stride : nullptr; };
#else
  auto slice_of = [&](long long r) { return p.rec + (r < last ? r : last) * Rec::SIZE; };  // clamped: a prefetch past the end re-reads the last record
  auto tape_of = [&](long long r) { return TAPES ? p.tapes + (r < last ? r : last) * (long long)p.tape_stride : nullptr; };
#endif
  if ((long long)blockIdx.x < p.n_rec) {
    long long r = blockIdx.x;
    RecSlice<NB> cur, nxt;
    cur.load(slice_of(r), Rec::VEC / 4, Rec::oSmall, w, ln, tape_of(r));
    nxt.load(slice_of(r + G), Rec::VEC / 4, Rec::oSmall, w, ln, tape_of(r + G));
    stage(cur, 0);
    __syncthreads();
    int buf = 0;
    for (r += G; r < p.n_rec; r += G, buf ^= 1) {
      cur = nxt;
      nxt.load(slice_of(r + G), Rec::VEC / 4, Rec::oSmall, w, ln, tape_of(r + G));
      // (measured and not kept: stage before gemm +4 %; the two waves of a SIMD in opposite orders: spills, +50 %;
      // sched_group_barrier MFMA / VALU interleaving: no change)
#ifndef PHNN_WG_DEBUG_NOGEMM
      gemm(buf);
#endif
#ifndef PHNN_WG_DEBUG_NOSTAGE
      stage(cur, buf ^ 1);
#else
      aB3 += cur.big[0][0] + cur.big[1][0] + cur.big[2][0] + cur.big[3][0] + cur.x[0] + cur.v[0] + cur.lam[0] + cur.dH[0] + cur.rv[0] + cur.uu[0] + cur.Hbar;
#endif
#ifndef PHNN_WG_DEBUG_NOBARRIER
      __syncthreads();
#endif
    }
    gemm(buf);
  }
  // ---- write this workgroup's partial gradient (every entry of its row that carries a gradient)
  float* row = p.slab + (long long)blockIdx.x * p.PP;
  constexpr BlobH<N, HID> BH = BL::H;
#pragma unroll
  for (int nt = 0; nt < T; ++nt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) row[BH.oW2 + (16 * w + 4 * ln.q + rr) * HID + 16 * nt + ln.i] = accW2[nt][rr];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int unit = 16 * w + 4 * ln.q + rr;
    float s3 = sum16(aW3[rr]), s2 = sum16(aB2[rr]), s1 = sum16(aB1[rr]);
    if (ln.i == 0) {
      row[BH.oW3 + unit] = s3;
      row[BH.oB2 + unit] = s2;
      row[BH.oB1 + unit] = s1;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float t = sum16(aW1[k][rr]);
      if (ln.i == 0) row[BH.oW1 + unit * N + k] = t;
    }
  }
  {
    float t = sum16(aB3);
    if (w == 0 && ln.lane == 0) row[BH.oB3] = t;
  }
  if constexpr (PHNN) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float t = sum16(aJ[j]);
      if (w == 0 && ln.i == 0 && ln.q < N) row[BL::oJ + ln.q * N + j] = t;
    }
    accR.write(row, BL::R, w, ln);
    if constexpr (!M::FIXG) accG.write(row, BL::G, w, ln);
  } else {
    // R_diag_raw: only rows 2, 3 reach the output; softplus' = sigmoid(raw) (image constants oC[8..11])
    float t2 = sum16(aRd[0]), t3 = sum16(aRd[1]);
    if (w == 0 && ln.lane == 0) {
      row[BL::oRd + 0] = 0.f;  // rows 0, 1 of R multiply the dq rows the reference discards: zero gradient
      row[BL::oRd + 1] = 0.f;
      row[BL::oRd + 2] = t2 * p.img[M::oC + 10];
      row[BL::oRd + 3] = t3 * p.img[M::oC + 11];
    }
  }
}

#ifdef PHNN_WGRAD_UNIT  // non-template kernel: defined in phnn_wgrad.hip only
// out[j] (+)= sum over slab rows of the padded-blob entry map[j] (map[j] < 0: a buffer / autograd constant -> 0).
// Fixed summation order: bitwise reproducible.
__global__ void k_wgrad_finish(const float* slab, int rows, int PP, const int* map, int P, float* out, int accumulate) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  const int src = map[j];
  float s = 0.f;
  if (src >= 0)
    for (int g = 0; g < rows; ++g) s += slab[(long long)g * PP + src];
  out[j] = accumulate ? out[j] + s : s;
}
#endif  // PHNN_WGRAD_UNIT

// K3: Adam on the controls + best-iterate tracking (torch.optim.Adam single-tensor order)
struct AdamParams {
  float* u;
  const float* g;
  float* m;
  float* v;
  long long count, per;
  float w1, w2, b2, bc2s, step_neg, eps;
  const float* cost;
  float* best_cost;
  float* best_u;
  float u_min, u_max;
  int has_u_bounds;
};

#ifndef PHNN_ADJOINT_UNIT  // the non-template kernels belong to phnn_mpc.hip only
__global__ void k_adam(AdamParams p) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.count) return;
  float u = p.u[idx];
  if (p.best_cost) {
    long long b = idx / p.per;
    // strict '<' against the value before this step's update (src/mpc_controller_canonical.py:212)
    if (p.cost[b] < p.best_cost[b]) p.best_u[idx] = p.has_u_bounds ? fminf(fmaxf(u, p.u_min), p.u_max) : u;
  }
  float g = p.g[idx], m = p.m[idx], v = p.v[idx];
  m = __builtin_fmaf(p.w1, g - m, m);
  v = v * p.b2 + (p.w2 * g) * g;
  float denom = sqrtf(v) / p.bc2s + p.eps;
  u = u + (p.step_neg * m) / denom;
  p.u[idx] = u;
  p.m[idx] = m;
  p.v[idx] = v;
}

// second pass of the best-iterate tracking: best_cost = min(best_cost, cost) (after k_adam used the old value)
__global__ void k_best_cost(const float* cost, float* best_cost, long long B) {
  long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && cost[b] < best_cost[b]) best_cost[b] = cost[b];
}

// Plant: the reference's ground-truth cart-pole (src/cartpole_simulator.py:63-112), float64, one thread per plant.
// Operation order as in the reference (compiled with -ffp-contract=off: no fused multiply-adds).
struct PlantParams {
  phnn_plant pl;
  double* state;        // (B,4) in/out
  const float* action;  // action[b * stride]
  long long stride, B;
  int has_u_bounds;
  float u_min, u_max;
  float* state_f32;
  int* done_step;
  const int* step_dev;
  int step_host;
  double* log_states;   // (T+1,B,4)
  float* log_controls;  // (T,B)
};

__global__ void k_plant_step(PlantParams p) {
  long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.B) return;
  const int step = p.step_dev ? *p.step_dev : p.step_host;
  float uf = p.action[b * p.stride];
  if (p.has_u_bounds) uf = fminf(fmaxf(uf, p.u_min), p.u_max);
  const double force = (double)uf;
  const double polemass_length = p.pl.masspole * p.pl.length, total_mass = p.pl.masspole + p.pl.masscart;
  double x = p.state[4 * b + 0], theta = p.state[4 * b + 1], x_dot = p.state[4 * b + 2], theta_dot = p.state[4 * b + 3];
  const double costheta = cos(theta), sintheta = sin(theta);
  const double temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
  const double thetaacc = (p.pl.gravity * sintheta - costheta * temp) /
                          (p.pl.length * (4.0 / 3.0 - p.pl.masspole * (costheta * costheta) / total_mass));
  const double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
  x = x + p.pl.dt * x_dot;
  theta = theta + p.pl.dt * theta_dot;
  x_dot = x_dot + p.pl.dt * xacc;
  theta_dot = theta_dot + p.pl.dt * thetaacc;
  p.state[4 * b + 0] = x;
  p.state[4 * b + 1] = theta;
  p.state[4 * b + 2] = x_dot;
  p.state[4 * b + 3] = theta_dot;
  if (p.state_f32) {
    p.state_f32[4 * b + 0] = (float)x;
    p.state_f32[4 * b + 1] = (float)theta;
    p.state_f32[4 * b + 2] = (float)x_dot;
    p.state_f32[4 * b + 3] = (float)theta_dot;
  }
  const bool done = fabs(x) > p.pl.x_limit || fabs(theta) > p.pl.theta_limit;
  if (p.done_step && done && p.done_step[b] < 0) p.done_step[b] = step;
  if (p.log_states) {
    double* row = p.log_states + ((long long)(step + 1) * p.B + b) * 4;
    row[0] = x;
    row[1] = theta;
    row[2] = x_dot;
    row[3] = theta_dot;
  }
  if (p.log_controls) p.log_controls[(long long)step * p.B + b] = uf;
}

// warm start: dst[b,t,:] = src[b,t+1,:], last step zero (src/mpc_controller_canonical.py:252-255); advances the step counter
__global__ void k_shift_controls(const float* src, float* dst, long long B, int H, int m, int* step_dev) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0 && step_dev) *step_dev += 1;  // k_plant_step of this control step has completed (stream order)
  const long long per = (long long)H * m;
  if (idx >= B * per) return;
  const long long r = idx % per;
  dst[idx] = r < per - m ? src[idx + m] : 0.0f;
}
#endif  // PHNN_ADJOINT_UNIT
