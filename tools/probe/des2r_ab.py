"""The matcher's R_indicator kernel alone: 4 x 5000 correlations per pair, 52 stacked pairs -> ms per launch (ROREG_DES2R_SPLIT=0: the round-2 kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roreg_amd import hip
torch.manual_seed(0)
n_pts, M = 5000 * 52, 5000 * 52
own = torch.randn(n_pts, 32, 60, device='cuda'); other = torch.randn(n_pts, 32, 60, device='cuda')
nn = torch.randint(0, n_pts, (M,), device='cuda')
for transpose in (True, False):
    out = hip.group_corr(own, other, perm_rows=None, bcast_rows=nn, transpose=transpose)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = hip.group_corr(own, other, perm_rows=None, bcast_rows=nn, transpose=transpose)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f'ROREG_DES2R_SPLIT={os.environ.get("ROREG_DES2R_SPLIT", "1")} transpose={transpose}: {1e3 * dt:.2f} ms per {M} correlations = {1e9 * dt / M:.1f} ns each; checksum {float(out.double().sum()):.6f} {float(out.double().abs().max()):.6f}')
