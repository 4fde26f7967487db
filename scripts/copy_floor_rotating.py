#!/usr/bin/env python3
"""Profiling aid: the plain-copy floor of the HBM-resident measurement — device-to-device copies of an obs-sized buffer
(65 536 x 137 fp32 = 35.9 MB) rotating over `--sets` independent source / destination pairs (8: 575 MB footprint, every
copy reads cold lines), next to the same copy on one pair (Infinity-Cache resident)."""
import argparse, time, torch
ap = argparse.ArgumentParser(); ap.add_argument('--sets', type=int, default=8); ap.add_argument('--iters', type=int, default=400)
a = ap.parse_args()
dev = torch.device('cuda', 0)
n = 65536 * 137
for sets in (1, a.sets):
    src = [torch.rand(n, device=dev) for _ in range(sets)]
    dst = [torch.empty(n, device=dev) for _ in range(sets)]
    for i in range(20): dst[i % sets].copy_(src[i % sets])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(a.iters): dst[i % sets].copy_(src[i % sets])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
    print('%d buffer pair(s): %.2f us per 35.9 MB -> 35.9 MB copy = %.0f GB/s (read + write)' % (sets, dt * 1e6, 2 * n * 4 / dt / 1e9))
