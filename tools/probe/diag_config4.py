"""Diagnostics for the low-overlap RD+RM configuration on synthetic scenes: match precision under the ground truth and pose error."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from roreg_amd import synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config

def weights(name, cfg, golden):
    net = name2network[name](cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden(golden).items()})
    return net.eval()

for RM in (True, False):
    cfg = default_config(keynum=2500, max_iter=1000, ET='yohoo', RD=True, RM=RM)
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    eng = RegistrationEngine(cfg, gf, et, rd_net=weights('RD_test', cfg, 'weights_RD'), rm_net=weights('RM_test', cfg, 'weights_RM') if RM else None)
    for seed in (31, 32, 33):
        ds = synth.make_scene(seed, n_clouds=3, n_kpts=5000, overlap=0.2, coord_noise=0.005)
        np.random.seed(3)
        res = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keynum=2500, keep_matches=True)
        for r in res:
            gt = ds.get_transform(r.id0, r.id1)
            m = r.matches.cpu().numpy()
            k0 = ds.get_kps(r.id0)[m[:, 0]]; k1 = ds.get_kps(r.id1)[m[:, 1]] @ gt[:, :3].T + gt[:, 3]
            prec = (np.linalg.norm(k0 - k1, axis=1) < 0.1).mean()
            R = r.trans[:3, :3] @ gt[:, :3].T
            rre = np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))); rte = np.linalg.norm(r.trans[:3, 3] - gt[:, 3])
            print(f'RM={RM} seed {seed} pair {r.id0}-{r.id1}: matches {r.n_match}, precision {prec:.3f}, rre {rre:.3f} rte {rte:.4f} recall {r.recalltime}')
