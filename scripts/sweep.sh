#!/bin/bash
run() { echo -n "[$*] "; env "$@" python scripts/time_rollout.py --iters 1000 | tail -1; }
run EB_TILE_ENVS=32
run EB_TILE_ENVS=32 EB_NO_XCD_REMAP=1
run EB_TILE_ENVS=16
run EB_TILE_ENVS=8
for ab in 5 7; do run EB_ABLATE=$ab; done
