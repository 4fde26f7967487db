#!/bin/bash
# Round-6 same-box A/B of the rollout kernel's experiment switches (eb_rollout.hip: EB_X, EB_PF).
# usage: bash scripts/r6_ab.sh <tag> <variants file: one "name defines..." per line> <commands file (bash, sourced once /tmp/ab/lib<name>.so exist)>
TAG=${1:-r6ab}; VARS=${2:-scripts/r6_ab_vars3.txt}; CMDS=${3:-scripts/r6_ab_cmds3.sh}
OUT=gpurun_out/$TAG; mkdir -p $OUT /tmp/ab
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-pass-failed"
S=env_build_amd/csrc
for f in eb_capi eb_kernels eb_env_kernels eb_env_step eb_env_step_t1 eb_env_step_t2 eb_policy; do
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=12 -c $S/$f.hip -o /tmp/ab/$f.o 2>> $OUT/build.log &
done
NAMES=""
while read -r name defs; do
  [ -z "$name" ] && continue
  NAMES="$NAMES $name"
  pc=12; case "$defs" in *PRELOAD14*) pc=14;; esac
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=$pc $defs -c $S/eb_rollout.hip -o /tmp/ab/rollout_$name.o 2>> $OUT/build.log &
done < $VARS
wait
for name in $NAMES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ab/eb_*.o /tmp/ab/rollout_$name.o -o /tmp/ab/lib$name.so 2>> $OUT/build.log
done
ls -la /tmp/ab/*.so
{
echo "== digests"
python scripts/r6_hash_rollout.py 2>&1 | grep digest
for name in $NAMES; do echo -n "$name: "; python scripts/r6_hash_rollout.py --lib /tmp/ab/lib$name.so 2>&1 | grep -E "digest|Error|error" | head -3; done
source $CMDS
} 2>&1 | tee $OUT/ab.txt
