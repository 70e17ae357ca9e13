import sys, time, types, zlib
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from roreg_amd import hip, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
args = types.SimpleNamespace(workload='3dmatch-full', kpts=5000, pair_lists='banded')
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo')
gf = name2network['GF_test'](cfg); gf.load_state_dict(synth.seeded_state_dict(gf, 101)); gf = gf.cuda().eval()
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202)); et = et.cuda().eval()
eng = RegistrationEngine(cfg, gf, et)
scenes, plan, totals = bench.build_workload(args, 0, 1)
seeds = {s: [(7 + zlib.crc32(f'{s}:{a}:{b}'.encode())) % (2 ** 32) for a, b in scenes[s][3]] for s in scenes}
def step():
    for (s, a, b) in plan:
        feats, keys, _, pl = scenes[s]
        eng.run_scene(feats, keys, pl[a:b], pair_seeds=seeds[s][a:b])
step(); torch.cuda.synchronize()
eng.phase_ms = {}
t = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print('step with synchronised phase marks: %.0f ms' % (1e3 * dt), {k: round(v, 1) for k, v in eng.phase_ms.items()})
eng.phase_ms = None
# host-only cost of the hypothesis draws of the kitchen scene
import cProfile, pstats
s = plan[0][0]; feats, keys, _, pl = scenes[s]
pr = cProfile.Profile(); pr.enable(); eng.run_scene(feats, keys, pl, pair_seeds=seeds[s]); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative'); st.print_stats(28)
