#!/bin/bash
# kernel register / spill / occupancy report for the render kernels
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Iinclude -Iraytracing_amd/csrc \
  --cuda-device-only -c raytracing_amd/csrc/rtx_hip.hip -o /tmp/rtx_dev.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -A12 "Function Name: _Z8k_render" | grep -v "^--" | grep "Function Name\|VGPRs:\|Spill\|Occupancy\|ScratchSize\|SGPRs:\|LDS Size"
