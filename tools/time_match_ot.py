"""Time the rotation-coherence matcher (Match_ot) on one pair, N = 5000 and 2500 keypoints; wall time vs kernel launches."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip, synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
net = name2network['RM_test'](default_config())
sd = dict(np.load('tests/golden/weights_RM.npz'))
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True); net.eval()
for N in (5000, 2500):
    ds = synth.make_scene(3, n_clouds=2, n_kpts=N, overlap=0.6)
    f0 = torch.from_numpy(ds.feats[0]).cuda(); f1 = torch.from_numpy(ds.feats[1]).cuda()
    f0 = f0 / f0.norm(dim=1, keepdim=True); f1 = f1 / f1.norm(dim=1, keepdim=True)
    batch = {'feats0': f1[None], 'feats1': f0[None], 'keys0': torch.from_numpy(ds.get_kps('1')[None].astype(np.float32)).cuda(),
             'keys1': torch.from_numpy(ds.get_kps('0')[None].astype(np.float32)).cuda()}
    with torch.no_grad():
        for it in range(4):
            torch.cuda.synchronize(); t = time.perf_counter(); out = net(batch); t1 = time.perf_counter(); torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f'Match_ot N={N}: {1e3 * (t2 - t):.2f} ms wall (host issue {1e3 * (t1 - t):.2f} ms), {int((out["matches0"] >= 0).sum())} matches')
