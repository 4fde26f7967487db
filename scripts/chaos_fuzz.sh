#!/bin/bash
# The randomised parity sweeps (scripts/fuzz_*.py) under the background load of scripts/chaos_gpu_tests.sh.  Usage: bash scripts/chaos_fuzz.sh [seconds each]
S=${1:-150}
python - <<'PY' &
import torch, time, random
x = [torch.randn(1 << k, device='cuda') for k in (12, 16, 20, 24)]
t0 = time.time()
while time.time() - t0 < 3600:
    a = random.choice(x)
    for _ in range(random.randint(1, 40)):
        a = torch.sin(a) * 1.0001 + 0.1
    if random.random() < 0.2:
        torch.cuda.synchronize()
PY
BG=$!
sleep 5
python scripts/fuzz_rollout.py --seconds $S --seed 41 2>&1 | tail -1
python scripts/fuzz_env_step.py --seconds $S --seed 42 2>&1 | tail -1
python scripts/fuzz_env.py --seconds $S --seed 43 2>&1 | tail -1
python scripts/fuzz_env_auto.py --seconds $S --seed 44 2>&1 | tail -1
kill $BG
