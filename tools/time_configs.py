"""Throughput of the other BASELINE.json configurations on the same synthetic chunk (secondary figures for DESIGN.md):
   A: mutual + yohoo (the bench config)   A': mutual + yohoc (rotation-bin RANSAC)   B: RD detector + mutual + yohoo   C: RD + RM (rotation-coherence matcher, keynum 2500) + yohoo."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
sys.path.insert(0, 'tests')
from conftest import load_golden

n_clouds, n_pairs = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 24)
scene = synth.make_scene(1000, n_clouds=n_clouds, n_kpts=5000, overlap=0.6, coord_noise=0.005)
order = np.random.default_rng(4242).permutation(len(scene.pair_ids))
pair_ids = [scene.pair_ids[i] for i in sorted(order[:n_pairs])]
feats = [torch.from_numpy(f).cuda() for f in scene.feats]; keys = [torch.from_numpy(k).cuda() for k in scene._kps]
for name, kw in (('mutual+yohoo', dict(keynum=5000, ET='yohoo')), ('mutual+yohoc', dict(keynum=5000, ET='yohoc')), ('RD+mutual+yohoo', dict(keynum=5000, ET='yohoo', RD=True)),
                 ('RD+RM+yohoo', dict(keynum=2500, ET='yohoo', RD=True, RM=True)),
                 ('RD+RM+yohoo@1000', dict(keynum=1000, ET='yohoo', RD=True, RM=True))):      # the reference README's command line
    if '--only' in sys.argv and sys.argv[sys.argv.index('--only') + 1] != name:
        continue
    if '--keynum' in sys.argv:
        kw = dict(kw, keynum=int(sys.argv[sys.argv.index('--keynum') + 1]))
    cfg = default_config(max_iter=1000, **kw)
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    rd = rm = None
    if cfg.RD:
        rd = name2network['RD_test'](cfg); rd.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RD').items()})
    if cfg.RM:
        rm = name2network['RM_test'](cfg); rm.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()})
    eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
    try:
        times = []
        for rep in range(7):                                      # two warm-up passes, then the median of five
            eng.phase_ms = {} if rep == 6 and '--phases' in sys.argv else None
            np.random.seed(7); torch.cuda.synchronize(); t = time.perf_counter()
            res = eng.run_scene(feats, keys, pair_ids, keynum=cfg.keynum)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t)
        dt = float(np.median(times[2:]))
        if eng.phase_ms:
            print('   phases (ms, synchronised):', {k: round(v, 1) for k, v in eng.phase_ms.items()})
        ok = np.mean([np.isfinite(r.trans).all() for r in res])
        print(f'{name}: {n_pairs / dt:.1f} pairs/s ({dt * 1e3:.1f} ms for {n_clouds} clouds / {n_pairs} pairs; finite results {ok:.2f})')
    except Exception as e:
        print(f'{name}: FAILED {type(e).__name__}: {e}')
