#!/bin/bash
# Same-box A/B: the library as it stood at the start of this round's second session (commit 2e18e99, sources under scripts/ab_src/, built here
# into /tmp) against the in-tree one — bench.py --env-step (65 536 x 16, 4 096 x 16, the flow source 65 536 x 60; plain and with auto reset)
# and the facade's no-resets loop, interleaved, two rounds.   usage: bash scripts/r5_ab_session.sh <tag>
TAG=${1:-r5ab}   # (the older sources: git archive 2e18e99 env_build_amd/csrc include/envbuild.h | tar -x -C scripts/ab_src — not tracked)
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=12"
S=scripts/ab_src/env_build_amd/csrc
for f in eb_capi eb_kernels eb_rollout eb_env_kernels eb_env_step eb_policy; do /opt/rocm/bin/hipcc $F -c $S/$f.hip -o /tmp/abA_$f.o 2>> $OUT/build.log & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abA_*.o -o /tmp/libA.so 2>> $OUT/build.log; ls -la /tmp/libA.so
{
for rep in 1 2; do
  for lib in /tmp/libA.so ""; do
    echo "== ${lib:-in-tree (final code)}"
    EB_AB_LIB=$lib python scripts/r5_ab_env.py 2>&1 | grep -v amdgpu.ids
  done
done
} | tee $OUT/ab_session.txt
