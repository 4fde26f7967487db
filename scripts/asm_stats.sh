#!/bin/bash
# Profiling aid (CPU): compile csrc/eb_env_step.hip to gfx950 assembly and print, per kernel of interest, the static
# instruction count, the number of s_barrier and the VGPR allocation.   usage: scripts/asm_stats.sh [out.s]
out=${1:-/tmp/eb_env_step.s}
here=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=12"
/opt/rocm/bin/hipcc $F --cuda-device-only -S "$here/env_build_amd/csrc/eb_env_step.hip" -o "$out" 2>&1 | grep -E "error" | head
[ -f "$out" ] || exit 1
for k in 15env_step_kernelILi0ELi64ELb0ELb0E 15env_step_kernelILi0ELi64ELb0ELb1E 15env_step_kernelILi0ELi64ELb1ELb0E 21env_reset_pool_kernelILi0ELi64E; do
  L=$(grep -n "^_ZN2eb${k}EEvNS_11EnvStepArgsE:" "$out" | cut -d: -f1)
  [ -z "$L" ] && continue
  awk -v L=$L 'NR>=L' "$out" | awk '/^\.Lfunc_end/{exit} {print}' > /tmp/_k.s
  echo "$k: $(awk '/^\t[a-z]/{n++} END{print n}' /tmp/_k.s) instructions, $(grep -c s_barrier /tmp/_k.s) barriers, $(awk -v L=$L 'NR>=L' "$out" | grep -m1 "NumVgprs:" | tr -d ';')"
done
