#!/bin/bash
# Round 6: GPU suite + rollout fuzz on the build with the new tile picker; priority by progress on the 1024-record tile (off by default) A/B
TAG=${1:-r6chk4}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/gpu_suite.txt
timeout 300 python scripts/fuzz_rollout.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/fuzz_rollout.txt
{
for rep in 1 2; do
  for cfg in "32768 16" "65536 16" "131072 16" "65536 8" "65536 9" "262144 8" "16384 32"; do
    set -- $cfg
    for sched in "-1,0" "-1,1"; do
      echo -n "rep $rep: "; python scripts/time_rollout.py --n-env $1 --n-veh $2 --sched=$sched --iters 3000 2>&1 | grep "us/step  "
    done
  done
done
} 2>&1 | tee $OUT/sched_tile1.txt
