"""How fast do the Sinkhorn iterations of the bench's own pairs settle?  Captures the final descriptors hip.sinkhorn_batch receives for the first
group of a bench-like scene (--RD --RM --keynum 5000), replays the float32 iteration with torch on the device and prints max |du|, |dv| per iteration
(log2 units, like the kernels) plus the score range.  usage: python tools/probe/sinkhorn_convergence.py [keynum] [n_pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roreg_amd import hip, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
import bench

keynum = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = default_config(keynum=keynum, max_iter=1000, ET='yohoo', RD=True, RM=True)
gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
rd, rm, _ = bench.rd_rm_nets(cfg)
eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
feats, keys, poses = synth.make_scene_device(500, 8, 5000, 0.6)
pairs = [(str(a), str(b)) for a, b in synth.scene_pair_list(8, n_pairs, 900, locality=8.0)]
cap = {}
orig = hip.sinkhorn_batch
def spy(sf, tf, seg_s, seg_t, alpha, iters, **kw):
    cap.setdefault('args', (sf.clone(), tf.clone(), seg_s.host.copy(), seg_t.host.copy(), alpha, iters))
    return orig(sf, tf, seg_s, seg_t, alpha, iters, **kw)
hip.sinkhorn_batch = spy
import roreg_amd.network.rot_coh_match as R
R.hip.sinkhorn_batch = spy
hip.sinkhorn_iteration_stats()
res = eng.run_scene(feats, keys, pairs, pair_seeds=list(range(len(pairs))))
print('iteration stats (run, pairs):', hip.sinkhorn_iteration_stats())
sf, tf, hs, ht, alpha, iters = cap['args']
print('alpha', alpha, 'iters', iters, 'pairs in group', len(hs) - 1)
L2E = 1.4426950408889634
for q in range(min(3, len(hs) - 1)):
    s = sf[hs[q]:hs[q + 1]]; t = tf[ht[q]:ht[q + 1]]
    m, n = s.shape[0], t.shape[0]
    Z = torch.full((m + 1, n + 1), float(alpha), device='cuda'); Z[:m, :n] = s @ t.T
    print(f'pair {q}: m {m} n {n} |s| {float(s.norm(dim=1).mean()):.2f} |t| {float(t.norm(dim=1).mean()):.2f} scores min {float(Z[:m,:n].min()):.2f} max {float(Z[:m,:n].max()):.2f} '
          f'quantiles 50/99/99.99 % {[round(float(x), 2) for x in torch.quantile(Z[:m,:n].flatten()[::97], torch.tensor([0.5, 0.99, 0.9999], device="cuda"))]}')
    Z = Z * L2E
    norm = -np.log(m + n)
    lmu = torch.full((m + 1,), norm * L2E, device='cuda'); lmu[m] = (np.log(n) + norm) * L2E
    lnu = torch.full((n + 1,), norm * L2E, device='cuda'); lnu[n] = (np.log(m) + norm) * L2E
    u = -Z.max(1).values; v = torch.zeros(n + 1, device='cuda')
    tol = lambda x: torch.maximum(x.abs() * 2.0 ** -22, torch.tensor(2.0 ** -20, device='cuda'))
    for k in range(200):
        un = u + (lmu - torch.log2(torch.exp2(Z + u[:, None] + v[None, :]).sum(1)))
        vn = v + (lnu - torch.log2(torch.exp2(Z + un[:, None] + v[None, :]).sum(0)))
        du, dv = float((un - u).abs().max()), float((vn - v).abs().max())
        viol = bool(((un - u).abs() > tol(un)).any() or ((vn - v).abs() > tol(vn)).any())
        nviol = int(((un - u).abs() > tol(un)).sum() + ((vn - v).abs() > tol(vn)).sum())
        if k < 5 or k % 10 == 9 or not viol:
            print(f'   it {k:3d} du {du:.3e} dv {dv:.3e} max|u| {float(un.abs().max()):.1f} potentials still moving {nviol}')
        u, v = un, vn
        if not viol:
            print('   settled after', k + 1); break
