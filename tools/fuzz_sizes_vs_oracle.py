"""Size fuzz (GPU): GF / ET networks in all three matrix-core modes and the batched matcher against the numpy oracle at sizes around
every tile boundary of the kernels (32-keypoint transform tiles, 128/256-wide GEMM tiles, 128x128 matcher tiles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import ref_numpy as O
from roreg_amd import hip, synth
from roreg_amd.group import tables
from roreg_amd.network import name2network
from roreg_amd.network.gf_fourier import FourierGF
from roreg_amd.parses.parses_test import default_config

T = tables()
cfg = default_config()
bad = 0
rng = np.random.default_rng(0)

gf = name2network['GF_test'](cfg); gf_sd = synth.seeded_state_dict(gf, 101)
et = name2network['ET_test'](cfg); et_sd = synth.seeded_state_dict(et, 202)
gf_np = {k: v.numpy() for k, v in gf_sd.items()}; et_np = {k: v.numpy() for k, v in et_sd.items()}
for B in (1, 2, 31, 32, 33, 64, 65, 127, 129, 255, 257):
    x = rng.standard_normal((B, 32, 60)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    want = O.gf_forward(x, gf_np, T.Nei)['eqv']
    batch = {k: rng.standard_normal((B, 32, 60)).astype(np.float32) for k in ['before_eqv0', 'before_eqv1', 'after_eqv0', 'after_eqv1']}
    batch['pre_idx'] = rng.integers(0, 60, B)
    want_q = O.et_forward({k: v.copy() for k, v in batch.items()}, et_np, T.Nei, T.P)
    for mode in hip.GEMM_MODES:
        net = gf.PartI_net
        if net._fourier is None:
            object.__setattr__(net, '_fourier', FourierGF(net))
        net._fourier.gemm = mode; et.gemm = mode
        with torch.no_grad():
            got = gf(torch.from_numpy(x))['eqv'].cpu().numpy()
            q = et({k: torch.from_numpy(np.ascontiguousarray(v.copy())) for k, v in batch.items()})['quaternion_pre'].cpu().numpy()
        e1 = float(np.abs(got - want).max()); e2 = float(np.abs(q - want_q).max())
        ok = e1 < 2e-5 and e2 < 1e-4
        bad += not ok
        print(f'B={B:4d} {mode:7s}: GF max err {e1:.2e}  ET max err {e2:.2e}  {"ok" if ok else "FAIL"}', flush=True)

# batched mutual matcher: ragged sizes around the 128-tile boundaries, several pairs per launch
for trial in range(6):
    tasks, wants = [], []
    for p in range(5):
        n0, n1 = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        if trial == 0:
            n0, n1 = [(1, 1), (127, 129), (128, 128), (129, 127), (256, 1)][p]
        e0 = rng.standard_normal((n0, 32, 60)).astype(np.float32); e1 = rng.standard_normal((n1, 32, 60)).astype(np.float32)
        k = min(n0, n1) // 2
        e1[:k] = e0[rng.permutation(n0)[:k]] + 0.01 * rng.standard_normal((k, 32, 60)).astype(np.float32)
        s0 = rng.permutation(n0)[:max(1, int(n0 * 0.8))]; s1 = rng.permutation(n1)[:max(1, int(n1 * 0.8))]
        wants.append(O.mutual_match(e0, e1, s0, s1))
        i0 = hip.inv_descriptor(torch.from_numpy(e0).cuda()); i1 = hip.inv_descriptor(torch.from_numpy(e1).cuda())
        tasks.append((i0, i1, torch.from_numpy(s0.astype(np.int64)).cuda(), torch.from_numpy(s1.astype(np.int64)).cuda()))
    buf, cnt = hip.mutual_match_batch(tasks)
    cnt = cnt.cpu().numpy()
    ok = all(np.array_equal(buf[q, :int(cnt[q])].cpu().numpy(), wants[q]) for q in range(5))
    bad += not ok
    print(f'matcher trial {trial}: sizes {[(int(t[2].shape[0]), int(t[3].shape[0])) for t in tasks]} {"ok" if ok else "FAIL"}', flush=True)
sys.exit(1 if bad else 0)
