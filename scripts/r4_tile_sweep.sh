#!/bin/bash
# round 4: (a) the rollout kernel's tile shapes at the per-rank shard of configs[3] on 8 GPUs (32 768 x 32) and at 16 384 / 65 536;
# (b) eb_debug_set_stage_paths 0 / 1 at the headline size (north_star's "LDS staging of the reference path per block" against the L2-resident
# cell grid the per-step kernel uses).   -> gpurun_out/<tag>_tile_sweep.txt, <tag>_ab_stage_paths.txt
tag=${1:-r4}
mkdir -p gpurun_out
{
for n in 16384 32768 65536; do for t in -1 0 1 2; do python scripts/time_rollout.py --n-env $n --n-veh 32 --tile $t --iters 400 2>/dev/null | tail -1; done; done
} | tee gpurun_out/${tag}_tile_sweep.txt
{
for rep in 1 2 3; do for sp in 0 1; do echo -n "stage_paths=$sp: "; python scripts/time_rollout.py --stage-paths $sp --n-env 65536 --n-veh 32 --iters 400 2>/dev/null | tail -1; done; done
for sp in 0 1; do echo -n "stage_paths=$sp: "; python scripts/time_rollout.py --stage-paths $sp --n-env 4096 --n-veh 16 --iters 400 2>/dev/null | tail -1; done
} | tee gpurun_out/${tag}_ab_stage_paths.txt
