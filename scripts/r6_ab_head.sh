#!/bin/bash
# Same-box A/B of the working tree against the sources under scripts/ab_src/base (a copy of a commit's csrc/, not tracked): rollout kernel loops
TAG=${1:-r6abh}; OUT=gpurun_out/$TAG; mkdir -p $OUT /tmp/abh
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14"
S=scripts/ab_src/base/env_build_amd/csrc
for f in eb_capi eb_kernels eb_rollout eb_env_kernels eb_env_step eb_env_step_t1 eb_env_step_t2 eb_policy; do /opt/rocm/bin/hipcc $F -c $S/$f.hip -o /tmp/abh/$f.o 2>> $OUT/build.log & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abh/*.o -o /tmp/libbase.so 2>> $OUT/build.log; ls -la /tmp/libbase.so
{
python scripts/r6_hash_rollout.py 2>&1 | grep digest; python scripts/r6_hash_rollout.py --lib /tmp/libbase.so 2>&1 | grep digest
for rep in 1 2 3 4; do
  for x in base tree; do
    lib=""; [ $x = base ] && lib="--lib /tmp/libbase.so"
    for cfg in "32768 32" "65536 32" "131072 32" "65536 64 --f16" "65536 9"; do
      set -- $cfg
      echo -n "rep $rep x=$x: "; python scripts/time_rollout.py $lib --n-env $1 --n-veh $2 $3 --iters 3000 2>&1 | grep "us/step  "
    done
  done
done
} 2>&1 | tee $OUT/ab.txt
