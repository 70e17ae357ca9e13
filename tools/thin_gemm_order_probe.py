"""Does the 32 -> 256 layer (write-bound, 2-10 K16 steps per tile) run its matrix phase and its store phase in lock step across the chip?
The work list is ordered irrep by irrep, so all resident workgroups have tiles of the same length; this probe permutes the list inside
every per-XCD stream (results do not depend on the tile order) and times the launch.
Usage: python tools/thin_gemm_order_probe.py [B]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
torch.manual_seed(0)
for (C, O) in [(32, 256), (256, 512)]:
    L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
    X = torch.randn(hip.coef_size(C, B), device='cuda')
    Xp, xb = hip.pack_coefs_f16x2(X, C, B)
    planes = hip.XDMA and C * O == 256 * 512
    if planes:
        Xp = hip.words_to_planes(Xp, C, B)
    bn = (torch.rand(O, device='cuda') + 0.5, torch.randn(O, device='cuda') * 0.1)
    nb = hip.next_bound(bn, L.bias)
    ref = None
    for order in ('as built', 'shuffled tiles per stream', 'shuffled blocks of 8 per stream', 'irreps interleaved per stream'):
        hip._tile_cache.clear()
        hip.irrep_gemm(Xp, None, C, O, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb, x_planes=planes)       # builds the list
        (key, t), = hip._tile_cache.items()
        tl = t.cpu().numpy().reshape(-1, 8, 3)                     # [i][stream][3]
        rng = np.random.default_rng(1)
        if order != 'as built':
            for k in range(8):
                s = tl[:, k].copy()
                live = s[s[:, 0] >= 0]
                if order == 'shuffled tiles per stream':
                    live = live[rng.permutation(len(live))]
                elif order == 'shuffled blocks of 8 per stream':
                    nb8 = len(live) // 8
                    head = live[:nb8 * 8].reshape(nb8, 8, 3)[rng.permutation(nb8)].reshape(-1, 3)
                    live = np.concatenate([head, live[nb8 * 8:]])
                else:                                              # round robin over the irreps, blocks of 8 tiles, proportionally
                    runs = [live[live[:, 0] == r] for r in range(5)]
                    runs = [r.reshape(-1, 3) for r in runs if len(r)]
                    tot = sum(len(r) for r in runs); pos = [0] * len(runs); out = []
                    while len(out) < tot:
                        q = min(range(len(runs)), key=lambda a: (pos[a] / max(len(runs[a]), 1)) if pos[a] < len(runs[a]) else 2.0)
                        out.extend(runs[q][pos[q]:pos[q] + 8]); pos[q] += 8
                    live = np.array(out)
                s[:len(live)] = live; s[len(live):] = (-1, 0, 0)
                tl[:, k] = s
            hip._tile_cache[key] = torch.from_numpy(np.ascontiguousarray(tl.reshape(-1, 3))).cuda()
        for _ in range(3): out, bd = hip.irrep_gemm(Xp, None, C, O, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb, x_planes=planes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out, bd = hip.irrep_gemm(Xp, None, C, O, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb, x_planes=planes)
        e1.record(); torch.cuda.synchronize()
        same = True if ref is None else (torch.equal(out, ref[0]) and torch.equal(bd, ref[1]))
        if ref is None: ref = (out.clone(), bd.clone())
        print(f'{C}->{O} B={B} tile order {order}: {e0.elapsed_time(e1) / 10:.3f} ms   bitwise equal to the built order: {same}', flush=True)
    hip._tile_cache.clear()
