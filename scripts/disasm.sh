#!/bin/bash
# Disassembles the gfx950 code object inside a built library: scripts/disasm.sh [library] > out.s
LIB=${1:-raytracing_amd/librtx_hip.so}
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co 2>/dev/null
$LLVM/llvm-objdump -d --no-show-raw-insn $T/dev.co
rm -rf $T
