"""The matcher's top-k search alone: 52 stacked pairs of 5000 x 5000 (and ragged ones), k = 16 / 8 -> ms per search and a checksum of the lists
(ROREG_TOPK_PACKED=0: one insertion chain per candidate; ROREG_TOPK_VALU=1: the vector-pipe kernel)."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roreg_amd import hip
torch.manual_seed(0)
tag = f'PACKED={os.environ.get("ROREG_TOPK_PACKED", "1")} VALU={os.environ.get("ROREG_TOPK_VALU", "0")}'
for name, la, lb in (('52 x (5000 x 5000)', [5000] * 52, [5000] * 52), ('ragged', [4999, 1200, 5000, 37, 3000, 2561] * 6, [5000, 4000, 1203, 64, 2559, 4999] * 6)):
    A = torch.nn.functional.normalize(torch.randn(sum(la), 32, device='cuda'), dim=1)
    B = torch.nn.functional.normalize(torch.randn(sum(lb), 32, device='cuda'), dim=1)
    B[5:4000:7] = B[4:3999:7]                                    # equal scores: ties go to the lower index
    sa, sb = hip.Segments(la), hip.Segments(lb)
    for k in (16, 8):
        idx, val = hip.topk_dot(A, B, k, want_val=True, segA=sa, segB=sb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            hip.topk_dot(A, B, k, want_val=True, segA=sa, segB=sb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f'{tag} {name} k={k}: {1e3 * dt:.3f} ms; lists crc {zlib.crc32(idx.cpu().numpy().tobytes()):08x} values crc {zlib.crc32(val.cpu().numpy().tobytes()):08x}')
