"""Stage-wise matcher run of the config-4 fixture `tag` on the reference's own NMS samples: which match rows differ from the reference's?"""
import sys, os, shutil; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from roreg_amd import synth, hip
from roreg_amd.parses.parses_test import default_config
from roreg_amd.network import name2network
tag = sys.argv[1] if len(sys.argv) > 1 else 'full_pipeline_rd_rm_o60_s3'
z = load_golden(tag)
kn = int(z['keynum']) if 'keynum' in z.files else 2500
cfg = default_config(keynum=kn, RD=True, RM=True)
gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
rm = name2network['RM_test'](cfg); rm.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()}, strict=True); rm.eval()
ds = synth.make_scene(int(z['scene_seed']), n_clouds=2, n_kpts=5000, overlap=float(z['overlap']), coord_noise=0.005, name='synth/scene0', portable=True)
with torch.no_grad():
    eqv = [gf(torch.from_numpy(f).cuda())['eqv'] for f in ds.feats]
for pc, e in zip(ds.pc_ids, eqv):
    print('yoho sample err', float(np.abs(e.cpu().numpy()[::250] - z[f'yoho_sample_{pc}']).max()))
s = [z[f'nms_{pc}'].astype(np.int64) for pc in ds.pc_ids]
k = [torch.from_numpy(ds.get_kps(pc).astype(np.float32)).cuda() for pc in ds.pc_ids]
batch = {'feats0': eqv[1][torch.from_numpy(s[1]).cuda()][None], 'feats1': eqv[0][torch.from_numpy(s[0]).cuda()][None],
         'keys0': k[1][torch.from_numpy(s[1]).cuda()][None], 'keys1': k[0][torch.from_numpy(s[0]).cuda()][None]}
with torch.no_grad():
    out = rm(batch)
m0 = out['matches0'][0].cpu().numpy(); sc = out['matching_scores0'][0].cpu().numpy()
mine = {(int(s[0][j]), int(s[1][i])): float(sc[i]) for i, j in enumerate(m0) if j >= 0}
want = {(int(a), int(b)): float(v) for (a, b), v in zip(z['match_0_1'].astype(np.int64), z['mscore_0_1'])}
print('mine', len(mine), 'ref', len(want), 'common', len(set(mine) & set(want)))
for r in sorted(set(mine) - set(want)): print('  only mine', r, mine[r])
for r in sorted(set(want) - set(mine)): print('  only ref ', r, want[r])
d = [abs(mine[r] - want[r]) for r in set(mine) & set(want)]
print('max score diff on common rows', max(d))
