"""Sinkhorn (100 iterations + read-out) per pair as a function of the number of pairs stacked per launch (m = n = 2500), both ways of
running the iterations: scores recomputed on the matrix cores in every pass (csrc/ot_flash.hip) and the materialised matrix re-read.
usage: python tools/time_sinkhorn.py [n | mxn] [P ...]   (ROREG_TS_RECOMPUTE_ONLY=1: skip the materialised form)"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
arg = sys.argv[1] if len(sys.argv) > 1 else '2500'
m, n = (int(x) for x in arg.split('x')) if 'x' in arg else (int(arg), int(arg))
only = os.environ.get('ROREG_TS_RECOMPUTE_ONLY') == '1'
Ps = [int(x) for x in sys.argv[2:]] or [1, 4, 8, 16, 30]
g = torch.Generator(device='cuda').manual_seed(0)
for P in Ps:
    s = torch.randn((P * m, 32), device='cuda', generator=g) * 0.5; t = torch.randn((P * n, 32), device='cuda', generator=g) * 0.5
    seg = hip.Segments([n] * P); seg_m = hip.Segments([m] * P)
    out = {}
    for rec in ((True,) if only else (True, False)):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = hip.sinkhorn_batch(s, t, seg_m, seg, 3.0, 100, recompute=(None if rec else False))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[rec] = (dt, res)
    if only:
        print(f'P={P:2d} {m}x{n}: recomputed {out[True][0] * 1e3:7.2f} ms = {out[True][0] * 1e3 / P:6.3f} ms/pair', flush=True)
        continue
    same = torch.equal(out[True][1][0], out[False][1][0])
    print(f'P={P:2d}: recomputed {out[True][0] * 1e3:7.2f} ms = {out[True][0] * 1e3 / P:6.3f} ms/pair   materialised {out[False][0] * 1e3:7.2f} ms = {out[False][0] * 1e3 / P:6.3f} ms/pair '
          f'({100 * P * (n + 1) * (m + 1) * 4 / out[False][0] / 1e12:5.2f} TB/s)   matches identical: {same}', flush=True)
