"""Where does the launching thread sit while run_scenes pipelines scenes?  A second thread samples its stack every 0.5 ms over three pipelined
passes of a kitchen-shaped scene (60 clouds, 449 pairs, keynum 5000); stays >= 3 ms on one line are listed, summed by line.
usage: python tools/probe/pipeline_sample.py [--mutual]"""
import collections, os, sys, threading, time, zlib
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
eng.run_scenes([job] * 3)
torch.cuda.synchronize()
samples, stop = [], threading.Event()
main_id = threading.main_thread().ident
def sampler():
    while not stop.is_set():
        fr = sys._current_frames().get(main_id)
        stack = []
        while fr is not None and len(stack) < 5:
            stack.append(f'{os.path.basename(fr.f_code.co_filename)}:{fr.f_lineno}({fr.f_code.co_name})'); fr = fr.f_back
        samples.append((time.perf_counter(), ' < '.join(stack)))
        time.sleep(0.0005)
threading.Thread(target=sampler, daemon=True).start()
t0 = time.perf_counter()
eng.run_scenes([job] * 4)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop.set()
print(f'{"rd+rm" if RD else "mutual"}: 4 pipelined passes {dt:.3f} s = {4 * len(pairs) / dt:.1f} pairs/s (with the sampler running)')
runs, cur = [], None
for t, w in samples:
    if cur is not None and cur[2] == w: cur[1] = t
    else:
        cur = [t, t, w]; runs.append(cur)
agg = collections.defaultdict(lambda: [0, 0.0])
for a, b, w in runs:
    if b - a >= 0.003:
        agg[w][0] += 1; agg[w][1] += b - a
print('stays >= 3 ms on one line, summed (count, total ms, innermost frames):')
for w, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f'   {c:4d} {1e3 * t:8.1f}  {w}')
