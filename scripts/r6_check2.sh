#!/bin/bash
# Round 6: (1) the GPU suite; (2) same-box A/B of the env step's preloaded hot arguments: the sources under scripts/ab_src/base (the tree
# before that change) built here into /tmp/libbase.so against the in-tree library; (3) more sizes of the rollout kernel's launch schedules.
TAG=${1:-r6chk2}; OUT=gpurun_out/$TAG; mkdir -p $OUT /tmp/abb
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14"
S=scripts/ab_src/base/env_build_amd/csrc
for f in eb_capi eb_kernels eb_rollout eb_env_kernels eb_env_step eb_env_step_t1 eb_env_step_t2 eb_policy; do /opt/rocm/bin/hipcc $F -c $S/$f.hip -o /tmp/abb/$f.o 2>> $OUT/build.log & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abb/*.o -o /tmp/libbase.so 2>> $OUT/build.log; ls -la /tmp/libbase.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/gpu_suite.txt
{
for rep in 1 2 3; do
  for lib in /tmp/libbase.so ""; do
    echo "== ${lib:-in-tree}"
    EB_AB_LIB=$lib python scripts/r5_ab_env.py 2>&1 | grep -v amdgpu.ids
  done
done
} 2>&1 | tee $OUT/ab_env.txt
{
for rep in 1 2; do
  for cfg in "131072 64 --f16" "16384 64 --f16" "65536 64" "32768 64" "65536 16" "131072 16" "65536 8" "65536 9"; do
    set -- $cfg
    for sched in "-1,-1" "0,0" "1,1" "0,1" "1,0"; do
      echo -n "rep $rep: "; python scripts/time_rollout.py --n-env $1 --n-veh $2 $3 --sched=$sched --iters 2000 2>&1 | grep "us/step  "
    done
  done
done
} 2>&1 | tee $OUT/sched.txt
