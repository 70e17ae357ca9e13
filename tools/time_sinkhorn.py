"""Sinkhorn (100 iterations + read-out) per pair as a function of the number of pairs stacked per launch (m = n = 2500)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
g = torch.Generator(device='cuda').manual_seed(0)
for P in (1, 2, 3, 4, 6, 8, 16, 32):
    s = torch.randn((P * n, 32), device='cuda', generator=g) * 0.5; t = torch.randn((P * n, 32), device='cuda', generator=g) * 0.5
    seg = hip.Segments([n] * P)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hip.sinkhorn_batch(s, t, seg, seg, 3.0, 100)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'P={P:2d}: {dt * 1e3:7.2f} ms  {dt * 1e3 / P:6.3f} ms/pair   {100 * P * (n + 1) ** 2 * 4 / dt / 1e12:5.2f} TB/s (one read of the matrix per iteration)')
