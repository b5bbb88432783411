#!/bin/bash
# Link a (hand-edited) gfx950 assembly of the adjoint unit into a loadable build of the library, for hazard bisection:
#   hipcc --cuda-device-only -S phnn_grad.hip -> grad.s  (edit: insert s_nop, ...)  -> tools/asm_variant.sh <name> grad_mod.s
#   -> build/ab/lib_<name>.so   (load with PHNN_LIB_PATH; the other three translation units are the product objects)
# Replays hipcc's own pipeline (hipcc -###): assemble, lld -shared, clang-offload-bundler, host pass with
# -fcuda-include-gpubinary.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; asm="$2"
LLVM=/opt/rocm/lib/llvm/bin
src="$ROOT/phnn_mpc_amd/csrc"
out="$ROOT/build/ab"; mkdir -p "$out"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$asm" -o "$out/$name.dev.o"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$out/$name.hsaco" "$out/$name.dev.o"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
  -input=/dev/null -input="$out/$name.hsaco" -output="$out/$name.hipfb"
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-host-only -O3 -std=c++17 -fPIC -ffp-contract=off \
  -Xclang -fcuda-include-gpubinary -Xclang "$out/$name.hipfb" -c -o "$out/$name.host.o" "$src/phnn_grad.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$out/lib_$name.so" "$src/phnn_mpc.o" "$out/$name.host.o" "$src/phnn_wgrad.o" "$src/phnn_split.o"
rm -f "$out/$name.dev.o" "$out/$name.host.o" "$out/$name.hipfb"
echo "build/ab/lib_$name.so"
