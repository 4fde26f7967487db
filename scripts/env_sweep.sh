#!/bin/bash
# (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the host wait on this stack: left out)
# launch-path experiment: runtime environment settings against the per-launch time of the headline kernel
OUT=gpurun_out/${1:-r2envsweep}; mkdir -p $OUT
T="python scripts/time_rollout.py --iters 600"
{
for rep in 1 2; do
for v in "X=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1" "ROC_ACTIVE_WAIT_TIMEOUT=100" "HIP_FORCE_DEV_KERNARG=0" "ROC_USE_FGS_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HSA_NO_SCRATCH_RECLAIM=1" "AMD_SERIALIZE_KERNEL=0"; do
  echo -n "$v : "; timeout 60 env $v $T 2>&1 | tail -1; echo
done
done
} | tee $OUT/sweep.txt
