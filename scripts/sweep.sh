#!/bin/bash
run() { echo -n "[$*] "; env "$@" python scripts/time_rollout.py --iters 1000 | tail -1; }
run EB_ABLATE=8
run EB_ABLATE=13
run EB_ABLATE=16
run EB_ABLATE=18
