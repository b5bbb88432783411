#!/usr/bin/env python3
"""Generate tools/../build/probe_hazard.hip: which instruction pairs around v_mfma_f32_16x16x32_f16 need software wait
states on gfx950, measured -- the evidence for the packed-f32 repeatability defect.

Each test is ONE inline-asm block on hard-coded registers (the compiler pads nothing inside asm): producer, K filler
instructions, consumer, then a long drain.  Inputs are constants per lane (A = alpha, B = beta in every element, so every
output element is C + 32 alpha beta whatever the fragment layout), so the expected value is known exactly.  The kernel
runs 8 waves per workgroup (two per SIMD), each wave with its own jitter ahead of the block and two independent MFMAs
queued in front of the producer, so the partner wave's traffic lands at every phase.  Mismatches are counted per
(test, filler kind, K).

fillers: nop = s_nop 0 | valu = v_mov_b32 (independent) | pk = v_pk_mul_f32 (independent)

usage: gen_probe_hazard.py out.hip ;  hipcc --offload-arch=gfx950 -O2 out.hip -o probe_hazard ; ./probe_hazard
"""
import sys

KS = list(range(0, 11))
FILLS = {"nop": "s_nop 0", "valu": "v_mov_b32 v60, v60", "pk": "v_pk_mul_f32 v[62:63], v[62:63], v[62:63]"}
MF = "v_mfma_f32_16x16x32_f16"
# registers: A v[40:43], B v[44:47], C v[48:51], T v[52:55], R v[56:59], scratch v60..v63, bg accumulators v[64:71]
TESTS = {
    # name: (sequence with {F} for the fillers, which register block holds the result, multiples of 32 alpha beta added to C)
    "chain_same_dst": (f"{MF} v[48:51], v[40:43], v[44:47], v[48:51]\n{{F}}{MF} v[48:51], v[40:43], v[44:47], v[48:51]\n", 48, 2),
    "srcC_from_prev_dst": (f"{MF} v[52:55], v[40:43], v[44:47], v[48:51]\n{{F}}{MF} v[56:59], v[40:43], v[44:47], v[52:55]\n", 56, 2),
    "raw_mfma_to_valu": (f"{MF} v[52:55], v[40:43], v[44:47], v[48:51]\n{{F}}v_mov_b32 v56, v52\nv_mov_b32 v57, v53\nv_mov_b32 v58, v54\nv_mov_b32 v59, v55\n", 56, 1),
    "raw_mfma_to_pk": (f"{MF} v[52:55], v[40:43], v[44:47], v[48:51]\n{{F}}v_pk_mul_f32 v[56:57], v[52:53], v[72:73]\nv_pk_mul_f32 v[58:59], v[54:55], v[72:73]\n", 56, 1),
    "war_srcB_valu": (f"{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n{{F}}v_mov_b32 v47, 0\nv_mov_b32 v46, 0\n", 56, 1),
    "war_srcB_pk": (f"{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n{{F}}v_pk_mul_f32 v[46:47], v[74:75], v[74:75]\nv_pk_mul_f32 v[44:45], v[74:75], v[74:75]\n", 56, 1),
    "war_srcA_pk": (f"{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n{{F}}v_pk_mul_f32 v[42:43], v[74:75], v[74:75]\nv_pk_mul_f32 v[40:41], v[74:75], v[74:75]\n", 56, 1),
    "war_srcC_valu": (f"{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n{{F}}v_mov_b32 v48, 0\nv_mov_b32 v49, 0\nv_mov_b32 v50, 0\nv_mov_b32 v51, 0\n", 56, 1),
    "war_srcC_pk": (f"{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n{{F}}v_pk_mul_f32 v[48:49], v[74:75], v[74:75]\nv_pk_mul_f32 v[50:51], v[74:75], v[74:75]\n", 56, 1),
    # operands written just before the MFMA: B (upper half) was junk, the VALU / packed op puts the right bits there
    "raw_valu_to_srcB": (f"v_mov_b32 v46, 0\nv_mov_b32 v47, 0\ns_nop 7\nv_mov_b32 v46, v44\nv_mov_b32 v47, v44\n{{F}}{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n", 56, 1),
    "raw_pk_to_srcB": (f"v_mov_b32 v46, 0\nv_mov_b32 v47, 0\ns_nop 7\nv_pk_mov_b32 v[46:47], v[44:45], v[44:45]\n{{F}}{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n", 56, 1),
    "raw_pk_to_srcC": (f"v_pk_mul_f32 v[50:51], v[48:49], v[72:73]\n{{F}}{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n", 56, 1),
    # destination overlapping an input operand (the register allocator does this when the operand dies at the MFMA)
    "dst_eq_srcB": (f"{{F}}{MF} v[44:47], v[40:43], v[44:47], v[48:51]\n", 44, 1),
    "dst_eq_srcA": (f"{{F}}{MF} v[40:43], v[40:43], v[44:47], v[48:51]\n", 40, 1),
    # ... followed by a dependent MFMA that takes the result as C and ALSO overwrites its own B (the packed build's pattern)
    "dst_eq_srcB_chain": (f"v_mov_b32 v52, v44\nv_mov_b32 v53, v44\nv_mov_b32 v54, v44\nv_mov_b32 v55, v44\ns_nop 3\n"
                          f"{MF} v[44:47], v[40:43], v[44:47], v[48:51]\n{{F}}{MF} v[52:55], v[40:43], v[52:55], v[44:47]\n", 52, 2),
    # an LDS load overwriting an operand of the MFMA just issued (write-after-read by an asynchronous writer): the LDS word
    # holds zeros; the load lands a while later -- the MFMA must have read its operand by then
    "war_srcA_lds": (f"{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n{{F}}ds_read_b128 v[40:43], v77\ns_waitcnt lgkmcnt(0)\n", 56, 1),
    "war_srcC_lds": (f"{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n{{F}}ds_read_b128 v[48:51], v77\ns_waitcnt lgkmcnt(0)\n", 56, 1),
    # the ADDRESS register of an LDS read overwritten right behind it (three more reads queued ahead, as in the kernels'
    # fragment loads): must return the 5.0 stored at the original address, not the 9.0 at address + 8192
    "war_lds_addr_valu": (f"ds_read_b128 v[64:67], v77\nds_read_b128 v[68:71], v77 offset:16\nds_read_b128 v[52:55], v77 offset:32\n"
                          f"ds_read_b128 v[56:59], v77\n{{F}}v_add_u32_e32 v77, 0x2000, v77\ns_waitcnt lgkmcnt(0)\n", 56, "5.0f"),
    "war_lds_addr_valu_tr": (f"ds_read_b128 v[64:67], v77\nds_read_b128 v[68:71], v77 offset:16\nds_read_b64_tr_b16 v[52:53], v77 offset:32\n"
                             f"ds_read_b64_tr_b16 v[56:57], v77\nds_read_b64_tr_b16 v[58:59], v77\n{{F}}v_add_u32_e32 v77, 0x2000, v77\ns_waitcnt lgkmcnt(0)\n", 56, "5.0f"),
    "raw_cvt_to_srcB": (f"v_mov_b32 v47, 0\ns_nop 7\nv_cvt_pk_f16_f32 v47, v76, v76\n{{F}}{MF} v[56:59], v[40:43], v[44:47], v[48:51]\n", 56, 1),
}

out = []
w = out.append
w("// GENERATED by tools/debug/gen_probe_hazard.py -- hazard probe around v_mfma_f32_16x16x32_f16 on gfx950")
w("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>\n#include <string>")
w("struct Res { unsigned bad; unsigned first_got; unsigned first_exp; };")
w("constexpr int ITER = 400;")
w("""__device__ __forceinline__ unsigned f16pair(int v) {  // small integer -> packed f16 pair bits
  _Float16 h = (_Float16)(float)v; unsigned short b = __builtin_bit_cast(unsigned short, h); return (unsigned)b | ((unsigned)b << 16); }""")
names = []
for tname, (seq, rreg, mult) in TESTS.items():
    for fname, fins in FILLS.items():
        for K in KS:
            expr = mult if isinstance(mult, str) else f"cv + {mult}.0f * 32.0f * (float)(al * be)"
            kn = f"k_{tname}_{fname}_{K}"
            names.append((kn, tname, fname, K))
            body = seq.replace("{F}", (fins + "\n") * K)
            asm = (
                "v_mov_b32 v40, %[a]\nv_mov_b32 v41, %[a]\nv_mov_b32 v42, %[a]\nv_mov_b32 v43, %[a]\n"
                "v_mov_b32 v44, %[b]\nv_mov_b32 v45, %[b]\nv_mov_b32 v46, %[b]\nv_mov_b32 v47, %[b]\n"
                "v_mov_b32 v48, %[c]\nv_mov_b32 v49, %[c]\nv_mov_b32 v50, %[c]\nv_mov_b32 v51, %[c]\n"
                "v_mov_b32 v52, %[s]\nv_mov_b32 v53, %[s]\nv_mov_b32 v54, %[s]\nv_mov_b32 v55, %[s]\n"
                "v_mov_b32 v56, %[s]\nv_mov_b32 v57, %[s]\nv_mov_b32 v58, %[s]\nv_mov_b32 v59, %[s]\n"
                "v_mov_b32 v60, 0\nv_mov_b32 v62, 1.0\nv_mov_b32 v63, 1.0\nv_mov_b32 v72, 1.0\nv_mov_b32 v73, 1.0\n"
                "v_mov_b32 v74, 0\nv_mov_b32 v75, 0\nv_mov_b32 v76, %[bf]\nv_mov_b32 v77, %[la]\n"
                "s_nop 7\ns_nop 7\n"
                # two independent MFMAs queued ahead of the test sequence: the producer meets a busy matrix pipe
                f"{MF} v[64:67], v[40:43], v[44:47], 0\n{MF} v[68:71], v[40:43], v[44:47], 0\n"
                + body +
                "s_nop 15\ns_nop 15\ns_nop 15\n"
                f"v_mov_b32 %[r0], v{rreg}\nv_mov_b32 %[r1], v{rreg + 1}\nv_mov_b32 %[r2], v{rreg + 2}\nv_mov_b32 %[r3], v{rreg + 3}\n"
                "s_nop 3\n"
            )
            asm_c = "\n".join('      "' + l + '\\n"' for l in asm.strip().split("\n"))
            w(f"""__global__ __launch_bounds__(512) void {kn}(Res* res, int slot) {{
  const int wave = threadIdx.x >> 6;
  __shared__ float zeros[2 * 512 * 4];  // first 8 KB: 5.0, second 8 KB: 9.0 (LDS address tests read the first half)
  for (int j = threadIdx.x; j < 2 * 512 * 4; j += blockDim.x) zeros[j] = j < 512 * 4 ? 5.0f : 9.0f;
  __syncthreads();
  unsigned bad = 0, fg = 0, fe = 0;
  for (int it = 0; it < ITER; ++it) {{
    const int al = 1 + (it + wave) % 3, be = 1 + (it * 7 + blockIdx.x) % 3;
    const float cv = (float)((it * 13 + threadIdx.x * 3) % 64);
    const int jit = (it * 5 + wave * 3 + blockIdx.x) % 11;
    for (int j = 0; j < jit; ++j) asm volatile("v_mov_b32 v61, v61" ::: "v61");
    float r0, r1, r2, r3;
    asm volatile(
{asm_c}
      : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
      : [a] "v"(f16pair(al)), [b] "v"(f16pair(be)), [c] "v"(cv), [s] "v"(-7777.0f), [bf] "v"((float)be), [la] "v"((unsigned)(threadIdx.x * 16) + (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)zeros)
      : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59",
        "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","memory");
    const float exp = {expr};
    if (r0 != exp || r1 != exp || r2 != exp || r3 != exp) {{
      if (!bad) {{ fg = __builtin_bit_cast(unsigned, r0 != exp ? r0 : (r1 != exp ? r1 : (r2 != exp ? r2 : r3))); fe = __builtin_bit_cast(unsigned, exp); }}
      ++bad;
    }}
  }}
  if (bad) {{ atomicAdd(&res[slot].bad, bad); res[slot].first_got = fg; res[slot].first_exp = fe; }}
}}""")
w("struct Ent { void (*k)(Res*, int); const char* test; const char* fill; int K; };")
w("static const Ent ents[] = {")
for kn, tname, fname, K in names:
    w(f'  {{{kn}, "{tname}", "{fname}", {K}}},')
w("};")
w(f"""int main(int argc, char** argv) {{
  const int N = sizeof(ents) / sizeof(ents[0]);
  const int threads = argc > 1 ? atoi(argv[1]) : 512;   // 512 = two waves per SIMD, 256 = one
  Res* d; hipMalloc(&d, N * sizeof(Res)); hipMemset(d, 0, N * sizeof(Res));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(ents[i].k, dim3(256), dim3(threads), 0, 0, d, i);
  if (hipDeviceSynchronize() != hipSuccess) {{ printf("launch failed\\n"); return 1; }}
  std::vector<Res> h(N); hipMemcpy(h.data(), d, N * sizeof(Res), hipMemcpyDeviceToHost);
  printf("threads per workgroup %d; mismatching (lane, iteration) pairs of %d; columns: K = 0..{KS[-1]} filler instructions between producer and consumer\\n", threads, 256 * threads * ITER);
  std::string cur;
  for (int i = 0; i < N; ++i) {{
    std::string key = std::string(ents[i].test) + " / " + ents[i].fill;
    if (key != cur) {{ if (!cur.empty()) printf("\\n"); printf("%-32s", key.c_str()); cur = key; }}
    printf(" %9u", h[i].bad);
  }}
  printf("\\n");
  return 0;
}}""")
open(sys.argv[1], "w").write("\n".join(out) + "\n")
print(len(names), "kernels")
