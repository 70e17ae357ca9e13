"""max |ours - reference| per Match_ot output on the two reference-generated cases (GPU)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from roreg_amd import synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
net = name2network['RM_test'](default_config())
net.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()}, strict=True); net.eval()
z = load_golden('match_ot')
with torch.no_grad():
    out = net({k: torch.from_numpy(z[k]) for k in ['feats0', 'feats1', 'keys0', 'keys1']})
for k in ('source_final', 'target_final', 'scores', 'matching_scores0', 'matching_scores1', 'scores_other'):
    d = np.abs(out[k].cpu().numpy() - z['out_' + k]); print('golden', k, float(d.max()), 'at |ref| up to', float(np.abs(z['out_' + k]).max()))
zf = load_golden('full_match_ot')
n = int(zf['n'])
ds = synth.make_scene(int(zf['scene_seed']), n_clouds=2, n_kpts=n, overlap=0.6, coord_noise=0.005, portable=True)
f0 = ds.feats[0]; f1 = ds.feats[1]
f0 = f0 / np.sqrt((f0 * f0).sum(1, keepdims=True)); f1 = f1 / np.sqrt((f1 * f1).sum(1, keepdims=True))
batch = {'feats0': torch.from_numpy(f1[None].copy()), 'feats1': torch.from_numpy(f0[None].copy()),
         'keys0': torch.from_numpy(ds.get_kps('1').astype(np.float32)[None].copy()), 'keys1': torch.from_numpy(ds.get_kps('0').astype(np.float32)[None].copy())}
with torch.no_grad():
    out = net(batch)
Z = out['scores'][0].cpu().numpy()
print('full scores sample', float(np.abs(Z[::40, ::40] - zf['scores_sample']).max()), 'lastrow', float(np.abs(Z[-1, ::10] - zf['scores_lastrow']).max()),
      'lastcol', float(np.abs(Z[::10, -1] - zf['scores_lastcol']).max()))
print('full source_final', float(np.abs(out['source_final'][0, :, ::25, 0].cpu().numpy() - zf['source_final_sample']).max()),
      'target_final', float(np.abs(out['target_final'][0, :, ::25, 0].cpu().numpy() - zf['target_final_sample']).max()))
print('full matching_scores0', float(np.abs(out['matching_scores0'][0].cpu().numpy() - zf['matching_scores0']).max()))
