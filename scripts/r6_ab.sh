#!/bin/bash
# Round-6 same-box A/B of the rollout kernel's experiment switches (eb_rollout.hip: EB_X).
# usage: bash scripts/r6_ab.sh <tag> "<X values>" <commands file (bash, sourced after the libraries /tmp/ab/libx<X>.so are built)>
TAG=${1:-r6ab}; XS=${2:-"2 4 8 12 14"}; CMDS=${3:-scripts/r6_ab_cmds1.sh}
OUT=gpurun_out/$TAG; mkdir -p $OUT /tmp/ab
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-pass-failed"
S=env_build_amd/csrc
for f in eb_capi eb_kernels eb_env_kernels eb_env_step eb_env_step_t1 eb_env_step_t2 eb_policy; do
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=12 -c $S/$f.hip -o /tmp/ab/$f.o 2>> $OUT/build.log &
done
for x in $XS; do
  pc=12; [ $((x & 4)) -ne 0 ] && pc=14
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=$pc -DEB_X=$x -c $S/eb_rollout.hip -o /tmp/ab/rollout_x$x.o 2>> $OUT/build.log &
done
wait
for x in $XS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ab/eb_*.o /tmp/ab/rollout_x$x.o -o /tmp/ab/libx$x.so 2>> $OUT/build.log
done
ls -la /tmp/ab/*.so
{
echo "== digests"
python scripts/r6_hash_rollout.py 2>&1 | grep digest
for x in $XS; do echo -n "x$x: "; python scripts/r6_hash_rollout.py --lib /tmp/ab/libx$x.so 2>&1 | grep -E "digest|Error|error" | head -3; done
source $CMDS
} 2>&1 | tee $OUT/ab.txt
