"""Time the batched mutual matcher: 60 pairs of 5000 x 5000 x 32."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
rng = np.random.default_rng(0)
clouds = [torch.from_numpy(rng.standard_normal((5000, 32)).astype(np.float32)).cuda() for _ in range(16)]
tasks = []
for p in range(60):
    a, b = rng.choice(16, 2, replace=False)
    tasks.append((clouds[a], clouds[b], torch.from_numpy(rng.permutation(5000)).cuda(), torch.from_numpy(rng.permutation(5000)).cuda()))
for _ in range(2):
    hip.mutual_match_batch(tasks)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    hip.mutual_match_batch(tasks)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f'mutual_match_batch 60 pairs: {dt*1e3:.2f} ms  ({dt/120*1e6:.1f} us per direction)')
