#!/bin/bash
# Robustness aid (GPU box): the GPU test suite while another process keeps the GPU busy with unrelated kernels of varying size —
# wave scheduling, cache contents and kernel start-up timing differ from a quiet box, which is what surfaces ordering hazards.
# Usage: bash scripts/chaos_gpu_tests.sh [rounds]
N=${1:-2}
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
for i in $(seq 1 $N); do python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
for i in $(seq 1 6); do python -m pytest tests/test_gpu_parity.py -m gpu -q -k "env_step or reset or masked" 2>&1 | tail -1; done
kill $BG
