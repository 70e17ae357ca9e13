"""Wall time of the engine's stages (with synchronisation between stages) for both GEMM modes."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo')
gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
eng = RegistrationEngine(cfg, gf, et)
scene = synth.make_scene(1000, n_clouds=16, n_kpts=5000, overlap=0.6, coord_noise=0.005)
feats = [torch.from_numpy(f).cuda() for f in scene.feats]; keys = [torch.from_numpy(k).cuda() for k in scene._kps]
order = np.random.default_rng(4242).permutation(len(scene.pair_ids)); pair_ids = [scene.pair_ids[i] for i in sorted(order[:60])]
for rep in range(3):
    eng.phase_ms = {} if rep == 2 else None
    np.random.seed(7)
    eng.run_scene(feats, keys, pair_ids)
print('phases (synchronised, ms):', {k: round(v, 2) for k, v in eng.phase_ms.items()})
eng.phase_ms = None
for mode in ['f32', 'split']:
    eng.set_gemm_mode(mode)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        clouds = eng.extract_many(feats, keys)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        np.random.seed(7)
        res = eng.run_scene(feats, keys, pair_ids)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{mode}: extract {1e3*(t1-t0):.1f} ms, whole scene {1e3*(t2-t1):.1f} ms')
