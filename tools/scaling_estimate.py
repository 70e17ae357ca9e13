"""What the strong-scaling curve of bench.py will look like, measured on ONE GPU: for N in 1, 2, 4, 8 the plan of distributed.shard_scenes
is built exactly as bench.py does, every rank's share is run (one after the other, same engine) and timed; estimated efficiency =
T(1) / (N * max_r T(r)).  No collective is involved (bench.py adds one all_gather of ~0.3 MB per step).
Usage: python tools/scaling_estimate.py [steps] [uniform|banded]"""
import sys, time, types, zlib
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from roreg_amd import hip, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
args = types.SimpleNamespace(workload='3dmatch-full', kpts=5000, pair_lists=sys.argv[2] if len(sys.argv) > 2 else 'uniform')
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo')
gf = name2network['GF_test'](cfg); gf.load_state_dict(synth.seeded_state_dict(gf, 101))
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202))
eng = RegistrationEngine(cfg, gf.cuda().eval(), et.cuda().eval())
t1 = None
for world in (1, 2, 4, 8):
    times, pairs, clouds = [], [], []
    for rank in range(world):
        scenes, plan, totals = bench.build_workload(args, rank, world)
        seeds = {s: [(7 + zlib.crc32(f'{s}:{a}:{b}'.encode())) % (2 ** 32) for a, b in scenes[s][3]] for s in scenes}
        def step():
            for (s, a, b) in plan:
                feats, keys, _, pl = scenes[s]
                eng.run_scene(feats, keys, pl[a:b], pair_seeds=seeds[s][a:b])
        step(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t) / steps)
        pairs.append(sum(b - a for _, a, b in plan))
        clouds.append(sum(len({int(i) for pr in scenes[s][3][a:b] for i in pr}) for s, a, b in plan))
        del scenes
        torch.cuda.empty_cache()
    if world == 1:
        t1 = times[0]
    print(f'N={world}: per-rank step time {["%.0f" % (1e3 * x) for x in times]} ms, pairs {pairs}, clouds extracted {clouds} (sum {sum(clouds)});  '
          f'throughput {1623 / max(times):.0f} pairs/s,  efficiency {t1 / (world * max(times)):.3f}', flush=True)
