#!/bin/bash
# Resource usage and the non-engine ISA of one k_tile instantiation (default: T=11, NT, 32-bit offsets, 2 tiles per workgroup)
#   tools/kernel_isa.sh [mangled-name-fragment]
frag=${1:-k_tileILi11ELb1ELb0ELi2E}
cd "$(dirname "$0")/../quantum_simulations_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -structurizecfg-skip-uniform-regions=true \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only -S -o /tmp/qsim_isa.s qsim_hip.hip 2>&1 | grep -A8 "Function Name: _Z6$frag" | grep -E "Function Name|SGPRs:|VGPRs:|Scratch|Occupancy"
awk "/^_Z6${frag}Ev8TileArgs:/,/s_endpgm/" /tmp/qsim_isa.s > /tmp/qsim_kernel.s
grep -n "global_load\|flat_load\|global_store\|flat_store\|v_writelane\|v_readlane\|scratch_" /tmp/qsim_kernel.s
