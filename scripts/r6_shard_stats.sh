#!/bin/bash
# rocprofv3 kernel stats of the rollout kernel's other instantiations: the 8-GPU shard (32 768 x 32: rolling loads), a native shape
# (65 536 x 9, task straight: the 1024-record tile), 262 144 x 32 (the strong-scaling job on one GPU)
TAG=${1:-r6shard}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p32768 -o p -- python scripts/time_rollout.py --n-env 32768 --iters 2000 > $OUT/t32768.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p9 -o p -- python scripts/time_rollout.py --task straight --n-env 65536 --n-veh 9 --iters 2000 > $OUT/t9.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p262144 -o p -- python scripts/time_rollout.py --n-env 262144 --iters 500 > $OUT/t262144.txt 2>&1
for d in p32768 p9 p262144; do echo "== $d"; grep "rollout_fused" $OUT/$d/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200; done | tee $OUT/stats.txt
