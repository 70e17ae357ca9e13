"""Robustness (GPU): non-finite inputs must never fault the device -- they may produce unmatched points / NaN transforms, not crashes.
Each case runs in its own process so that a device fault is reported, not fatal to the sweep."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ['engine_mutual_yohoo', 'engine_mutual_yohoc', 'engine_rd_rm_yohoo', 'match_many', 'knn_nms']

def run(case):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np, torch
    from conftest import load_golden
    from roreg_amd import synth, hip
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.parses.parses_test import default_config
    def weights(name, cfg, golden):
        net = name2network[name](cfg); net.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden(golden).items()}); return net.eval()
    ds = synth.make_scene(5, n_clouds=3, n_kpts=300, overlap=0.6)
    feats = [f.copy() for f in ds.feats]; keys = [ds.get_kps(i).copy() for i in ds.pc_ids]
    feats[0][7] = np.nan; feats[1][::50, 3, 5] = np.inf; feats[2][11] = -np.inf            # poisoned keypoints
    keys[1][5] = np.nan
    if case.startswith('engine'):
        RD = RM = 'rd_rm' in case
        ET = 'yohoc' if case.endswith('yohoc') else 'yohoo'
        cfg = default_config(keynum=200 if RM else 300, max_iter=1000, ET=ET, RD=RD, RM=RM)
        gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
        et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
        eng = RegistrationEngine(cfg, gf, et, rd_net=weights('RD_test', cfg, 'weights_RD') if RD else None, rm_net=weights('RM_test', cfg, 'weights_RM') if RM else None)
        np.random.seed(1)
        res = eng.run_scene(feats, keys, ds.pair_ids, keynum=cfg.keynum)
        torch.cuda.synchronize()
        print(case, 'ok:', [(r.n_match, bool(np.isfinite(r.trans).all())) for r in res])
    elif case == 'match_many':
        cfg = default_config()
        rm = weights('RM_test', cfg, 'weights_RM')
        pairs = [(torch.from_numpy(feats[0]).cuda(), torch.from_numpy(feats[1]).cuda(), torch.from_numpy(keys[0]).float().cuda(), torch.from_numpy(keys[1]).float().cuda()),
                 (torch.from_numpy(feats[2]).cuda(), torch.from_numpy(ds.feats[0]).cuda(), torch.from_numpy(keys[2]).float().cuda(), torch.from_numpy(keys[0]).float().cuda())]
        with torch.no_grad():
            out = rm.match_many(pairs)
            one = rm({'feats0': pairs[0][0][None], 'feats1': pairs[0][1][None], 'keys0': pairs[0][2][None], 'keys1': pairs[0][3][None]})
        torch.cuda.synchronize()
        print(case, 'ok:', [int((m >= 0).sum()) for m, _ in out], int((one['matches0'] >= 0).sum()))
    elif case == 'knn_nms':
        from roreg_amd.test.matcher import NMS_sample
        s = np.random.default_rng(0).random(300).astype(np.float32); s[3] = np.nan
        idx = NMS_sample(200, 5).sample(keys[1], s)
        print(case, 'ok:', len(idx))

if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1]); sys.exit(0)
    bad = 0
    for c in CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=600)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and 'amdgpu.ids' not in l and 'Extension modules' not in l]
        print(f'{c}: rc={r.returncode}', '|', tail[-1][:200] if tail else '')
        bad += r.returncode != 0
    sys.exit(1 if bad else 0)
