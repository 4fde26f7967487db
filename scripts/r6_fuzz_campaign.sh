#!/bin/bash
# Round 6: a longer randomised parity campaign on the final code — every fuzz sweep for S seconds with fresh seeds, quiet and under
# background load, then the flow-source soak.   usage: bash scripts/r6_fuzz_campaign.sh <tag> [seconds each]
TAG=${1:-r6fuzz}; S=${2:-240}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
{
echo "== quiet box, $S s each"
python scripts/fuzz_rollout.py --seconds $S --seed 601 2>&1 | tail -1
python scripts/fuzz_env_step.py --seconds $S --seed 602 2>&1 | tail -1
python scripts/fuzz_env.py --seconds $S --seed 603 2>&1 | tail -1
python scripts/fuzz_env_auto.py --seconds $S --seed 604 2>&1 | tail -1
python scripts/fuzz_env_auto.py --seconds $S --seed 605 --waves 4 2>&1 | tail -1
echo "== under background load (scripts/chaos_fuzz.sh $S)"
bash scripts/chaos_fuzz.sh $S 2>&1 | grep -v amdgpu.ids
echo "== soak, flow source, auto reset"
timeout 900 python scripts/soak_env.py --traffic flows --auto-reset --steps 20000 2>&1 | grep -v amdgpu.ids | tail -2
} 2>&1 | tee $OUT/campaign.txt
