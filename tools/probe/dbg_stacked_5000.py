import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from conftest import load_golden
from roreg_amd import synth, hip
from roreg_amd.parses.parses_test import default_config
from roreg_amd.network import name2network
import test_hip_fullsize as T
net = name2network['RM_test'](default_config())
net.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()}, strict=True); net.eval()
z = load_golden('full_match_ot_5000')
f0, f1, k0, k1 = T._match_ot_inputs(z)
n = int(z['n']); cu = T.cu
seg = hip.Segments([n])
with torch.no_grad():
    (m0, s0), = net.match_stacked(cu(f1), cu(f0), cu(k1), cu(k0), seg, seg)
m0 = m0.cpu().numpy(); s0 = s0.cpu().numpy(); w = z['matches0'].astype(np.int64); ws = z['matching_scores0']
bad = np.nonzero(m0 != w)[0]
print('mismatches', len(bad), 'valid ref', (w>=0).sum(), 'valid mine', (m0>=0).sum())
for i in bad[:20]: print(i, 'mine', m0[i], s0[i], 'ref', w[i], ws[i])
print('max |ds| on agreeing', np.abs(s0 - ws)[m0 == w].max())
batch = {'feats0': torch.from_numpy(f1[None].copy()), 'feats1': torch.from_numpy(f0[None].copy()), 'keys0': torch.from_numpy(k1[None].copy()), 'keys1': torch.from_numpy(k0[None].copy())}
with torch.no_grad(): out = net(batch)
fs = out['matching_scores0'][0].cpu().numpy(); fm = out['matches0'][0].cpu().numpy()
for i in bad[:20]: print(i, 'forward', fm[i], fs[i])
