"""Where does the HOST spend its time while the --RD --RM pipeline runs?  cProfile over three pipelined passes of a kitchen-shaped scene
(60 clouds, 449 pairs, keynum 5000); the GPU idles whenever the host is the slower side between a download and the next upload."""
import cProfile, io, os, pstats, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roreg_amd import hip, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
import bench

RD = '--mutual' not in sys.argv
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo', RD=RD, RM=RD)
gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
rd = rm = None
if RD:
    rd, rm, _ = bench.rd_rm_nets(cfg)
eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
feats, keys, poses = synth.make_scene_device(500, 60, 5000, 0.6)
pairs = [(str(a), str(b)) for a, b in synth.scene_pair_list(60, 449, 900, locality=8.0)]
seeds = [(7 + zlib.crc32(f'k:{a}:{b}'.encode())) % (2 ** 32) for a, b in pairs]
job = (feats, keys, pairs, dict(pair_seeds=seeds))
eng.run_scenes([job] * 2)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
eng.run_scenes([job] * 3)
torch.cuda.synchronize()
pr.disable()
print(f'3 passes: {time.perf_counter() - t0:.3f} s')
for key in ('tottime', 'cumtime'):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28); print(s.getvalue()[:6000])
