#!/bin/bash
# Round-5: the coarse grid level of the closest-point search — its parity tests, then the façade loops (pool: no resets is the case it is for; flows).
TAG=${1:-r5og}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
timeout 1200 python -m pytest tests -m gpu -x -q -k "coarse or off_the_map or records_no_source or flow_rule or rollout or closest or tracking or G4" > $OUT/pytest_sel.log 2>&1; echo "pytest(sel) rc=$?"; tail -3 $OUT/pytest_sel.log
timeout 900 python scripts/time_env_step.py --sizes 4096,65536 --traffic pool --steps 2000 > $OUT/facade_pool.txt 2>&1; cat $OUT/facade_pool.txt
timeout 900 python scripts/time_env_step.py --sizes 65536 --traffic flows --steps 50 > $OUT/facade_flows.txt 2>&1; cat $OUT/facade_flows.txt
