// probe_mfma_group.hip -- the fragment-load / MFMA group of the 128x128 products (sq_fwd_h) as the compiler emitted it
// in the packed-f32 build, replayed in isolation with exactly known answers: four ds_read_b128 of weight fragments, six
// v_mfma_f32_16x16x32_f16 on two accumulators, the next group's reads overwriting the fragment registers right behind
// the last MFMA.  A long s_nop behind the 5th MFMA of a group (where the real kernel becomes non-repeatable,
// tools/debug/asm_edit_at.py) is inserted in some iterations / waves.  8 waves per workgroup = two per SIMD.
//   LDS holds f16 constant ALPHA in every element; B = BETA everywhere; so each MFMA adds 32 ALPHA BETA to every element.
// usage: probe_mfma_group [variant]   variant bit 0: delay behind MFMA 5, bit 1: partner waves 4-7 run without delay,
//                                     bit 2: delay behind MFMA 4 instead, bit 3: no overwrite of fragments by the next reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define MF "v_mfma_f32_16x16x32_f16 "
#define GROUP(DELAY5, DELAY4)                                             \
  "ds_read_b128 v[86:89], %[ad]\n"                                        \
  "ds_read_b128 v[90:93], %[ad] offset:4096\n"                            \
  "ds_read_b128 v[94:97], %[ad] offset:1024\n"                            \
  "ds_read_b128 v[98:101], %[ad] offset:5120\n"                           \
  "s_waitcnt lgkmcnt(2)\n"                                                \
  MF "v[78:81], v[90:93], v[54:57], v[78:81]\n"                           \
  "s_waitcnt lgkmcnt(0)\n"                                                \
  MF "v[82:85], v[98:101], v[54:57], v[82:85]\n"                          \
  MF "v[78:81], v[86:89], v[58:61], v[78:81]\n"                           \
  MF "v[82:85], v[94:97], v[58:61], v[82:85]\n"                           \
  DELAY4                                                                  \
  MF "v[78:81], v[86:89], v[54:57], v[78:81]\n"                           \
  DELAY5                                                                  \
  MF "v[82:85], v[94:97], v[54:57], v[82:85]\n"

#define CLOB "v54","v55","v56","v57","v58","v59","v60","v61","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89", \
             "v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","memory"

using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ unsigned partner_stream(int kind, int n) {  // exact integer arithmetic: x <- x * 1 + 1 etc.
  f32x2 x = {0.f, 1.f}, y = {3.f, 5.f}, z = {0.f, 2.f}, x2 = {7.f, 9.f};
  const f32x2 one = {1.f, 1.f};
  for (int k = 0; k < n; ++k) {
    if (kind == 1) {
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_fma_f32 %3, %3, %4, %4\n"
                   : "+v"(x), "+v"(y), "+v"(z), "+v"(x2) : "v"(one));
    } else {
      asm volatile("v_fma_f32 %0, %0, %2, %2\nv_fma_f32 %1, %1, %2, %2\n" : "+v"(x[0]), "+v"(x[1]) : "v"(1.0f));
      asm volatile("v_mul_f32 %0, %0, %2\nv_mul_f32 %1, %1, %2\n" : "+v"(y[0]), "+v"(y[1]) : "v"(1.0f));
      asm volatile("v_add_f32 %0, %0, %2\nv_add_f32 %1, %1, %2\n" : "+v"(z[0]), "+v"(z[1]) : "v"(1.0f));
      asm volatile("v_fma_f32 %0, %0, %2, %2\nv_fma_f32 %1, %1, %2, %2\n" : "+v"(x2[0]), "+v"(x2[1]) : "v"(1.0f));
    }
  }
  return (x[0] != (float)n) + (x[1] != (float)(n + 1)) + (y[0] != 3.f) + (y[1] != 5.f) + (z[0] != (float)n) +
         (z[1] != (float)(n + 2)) + (x2[0] != (float)(n + 7)) + (x2[1] != (float)(n + 9));
}

template <int VARIANT, int PARTNER = 0>
__global__ __launch_bounds__(512) void k(unsigned* bad, int iters) {
  __shared__ _Float16 w[8192];  // 16 KB of ALPHA = 2.0
  for (int j = threadIdx.x; j < 8192; j += blockDim.x) w[j] = (_Float16)2.0f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned ad = (unsigned)(unsigned long long)(__attribute__((address_space(3))) _Float16*)w + lane * 16;
  const unsigned bbits = 0x3C003C00u;  // BETA = 1.0 pairs
  unsigned nbad = 0;
  if (PARTNER && wave >= 4) {  // the SIMD partners run a VALU-only stream for about as long as the group waves take
    nbad = partner_stream(PARTNER, iters * 120);
    if (nbad) atomicAdd(bad + 1, nbad);
    return;
  }
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < (it * 5 + wave * 3 + blockIdx.x) % 11; ++j) asm volatile("v_mov_b32 v102, v102" ::: "v102");
    const bool delayed = ((VARIANT & 2) ? wave < 4 : true) && ((it + wave) % 3 != 1);
    float r[8];
#define BODY(D5, D4)                                                                                                        \
    asm volatile(                                                                                                           \
      "v_mov_b32 v54, %[b]\nv_mov_b32 v55, %[b]\nv_mov_b32 v56, %[b]\nv_mov_b32 v57, %[b]\n"                                \
      "v_mov_b32 v58, %[b]\nv_mov_b32 v59, %[b]\nv_mov_b32 v60, %[b]\nv_mov_b32 v61, %[b]\n"                                \
      "v_mov_b32 v78, 0\nv_mov_b32 v79, 0\nv_mov_b32 v80, 0\nv_mov_b32 v81, 0\n"                                            \
      "v_mov_b32 v82, 0\nv_mov_b32 v83, 0\nv_mov_b32 v84, 0\nv_mov_b32 v85, 0\ns_nop 7\n"                                   \
      GROUP(D5, D4) GROUP(D5, D4) GROUP(D5, D4) GROUP(D5, D4) GROUP(D5, D4) GROUP(D5, D4) GROUP(D5, D4) GROUP(D5, D4)       \
      "s_nop 15\ns_nop 15\n"                                                                                                \
      "v_mov_b32 %[r0], v78\nv_mov_b32 %[r1], v79\nv_mov_b32 %[r2], v80\nv_mov_b32 %[r3], v81\n"                            \
      "v_mov_b32 %[r4], v82\nv_mov_b32 %[r5], v83\nv_mov_b32 %[r6], v84\nv_mov_b32 %[r7], v85\ns_nop 3\n"                   \
      : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]), [r4] "=&v"(r[4]), [r5] "=&v"(r[5]),         \
        [r6] "=&v"(r[6]), [r7] "=&v"(r[7])                                                                                  \
      : [ad] "v"(ad), [b] "v"(bbits) : CLOB)
    if (delayed && (VARIANT & 1)) {
      if (VARIANT & 4) { BODY("", "s_nop 15\n"); } else { BODY("s_nop 15\n", ""); }
    } else {
      BODY("", "");
    }
    const float e = 8.0f * 3.0f * 32.0f * 2.0f;  // 8 groups x 3 MFMAs per accumulator x 32 x alpha x beta
    for (int q = 0; q < 8; ++q) nbad += r[q] != e;
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
  unsigned* d;
  if (hipMalloc(&d, 8) != hipSuccess) return 1;
  void (*ks[])(unsigned*, int) = {k<0>, k<1>, k<3>, k<5>, k<7>, k<0, 1>, k<1, 1>, k<5, 1>, k<0, 2>, k<1, 2>, k<5, 2>};
  const char* names[] = {"no delay", "s_nop 15 behind MFMA 5, every wave (2 of 3 iterations)", "same, waves 0-3 only",
                         "s_nop 15 behind MFMA 4, every wave", "s_nop 15 behind MFMA 4, waves 0-3 only",
                         "no delay | partners: packed-f32 stream", "delay behind MFMA 5 | partners: packed-f32 stream",
                         "delay behind MFMA 4 | partners: packed-f32 stream", "no delay | partners: plain VALU stream",
                         "delay behind MFMA 5 | partners: plain VALU stream", "delay behind MFMA 4 | partners: plain VALU stream"};
  for (int v = 0; v < 11; ++v) {
    unsigned tot = 0, totp = 0;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipMemset(d, 0, 8);
      hipLaunchKernelGGL(ks[v], dim3(1024), dim3(512), 0, 0, d, 300);
      unsigned h[2];
      if (hipMemcpy(h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
      tot += h[0];
      totp += h[1];
    }
    printf("%-60s mismatching accumulator values: %u, partner stream values: %u\n", names[v], tot, totp);
  }
  return 0;
}
