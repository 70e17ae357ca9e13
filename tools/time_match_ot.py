import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from roreg_amd import hip, synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
net = name2network['RM_test'](default_config())
sd = dict(np.load('/root/repo/tests/golden/weights_RM.npz'))
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True); net.eval()
ds = synth.make_scene(3, n_clouds=2, n_kpts=5000, overlap=0.6)
f0 = torch.from_numpy(ds.feats[0]).cuda(); f1 = torch.from_numpy(ds.feats[1]).cuda()
f0 = f0/ f0.norm(dim=1, keepdim=True); f1 = f1/f1.norm(dim=1,keepdim=True)
batch = {'feats0': f1[None], 'feats1': f0[None], 'keys0': torch.from_numpy(ds.get_kps('1')[None].astype(np.float32)).cuda(), 'keys1': torch.from_numpy(ds.get_kps('0')[None].astype(np.float32)).cuda()}
with torch.no_grad():
    for it in range(3):
        torch.cuda.synchronize(); t=time.perf_counter(); out = net(batch); torch.cuda.synchronize(); print('Match_ot N=5000: %.2f ms' % ((time.perf_counter()-t)*1e3), int((out['matches0']>=0).sum()))
