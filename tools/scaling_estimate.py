"""What the strong-scaling curve of bench.py will look like, measured on ONE GPU: for N in 1, 2, 4, 8 the plan of distributed.shard_scenes is
built exactly as bench.py does, every rank's share is run through distributed.run_plan (one rank after the other, same engine) and timed;
estimated efficiency = T(1) / (N * max_r T(r)).  With the extractor-output exchange (default) a rank extracts only the clouds it owns; the
clouds it would RECEIVE are handed to it by a stand-in exchange object from copies extracted outside the timed region (the transfer itself
-- 38.4 MB per cloud over xGMI beside the rank's whole scenes -- and the result-table all_gather of ~0.3 MB are not modelled); it still
pays for rebuilding the received clouds' derivatives.
Usage: python tools/scaling_estimate.py [steps] [uniform|banded] [--no-exchange] [--worlds=1,2,4,8]"""
import sys, time, types, zlib
sys.path.insert(0, '.')
import torch
import bench
from roreg_amd import distributed as D, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config

argv = [a for a in sys.argv[1:] if not a.startswith('--')]
steps = int(argv[0]) if argv else 2
exchange = '--no-exchange' not in sys.argv
args = types.SimpleNamespace(workload='3dmatch-full', kpts=5000, pair_lists=argv[1] if len(argv) > 1 else 'banded')
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo')
gf = name2network['GF_test'](cfg); gf.load_state_dict(synth.seeded_state_dict(gf, 101))
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202))
eng = RegistrationEngine(cfg, gf.cuda().eval(), et.cuda().eval())


class HandedOver:
    """Stand-in for distributed.EqvExchange on one GPU: the clouds this rank would receive were extracted beforehand."""
    def __init__(self, eqv): self.eqv = eqv
    def start(self, transfers, get_eqv, alloc): pass
    def wait(self): return dict(self.eqv)


t1 = None
worlds = [int(x) for x in next((a.split('=')[1] for a in sys.argv if a.startswith('--worlds=')), '1,2,4,8').split(',')]
for world in worlds:
    times, pairs, clouds = [], [], []
    for rank in range(world):
        scenes, plan, totals = bench.build_workload(args, rank, world, exchange=exchange)
        transfers = totals['transfers']
        seeds = {s: [(7 + zlib.crc32(f'{s}:{a}:{b}'.encode())) % (2 ** 32) for a, b in scenes[s][3]] for s in scenes}
        inputs = lambda s: (scenes[s][0], scenes[s][1], scenes[s][3], seeds[s])
        recv = [(s, i) for s, i, src, dst in transfers if dst == rank]
        handed = {}
        for s in sorted({s for s, _ in recv}):
            ids = sorted(i for sc, i in recv if sc == s)
            for i, c in zip(ids, eng.extract_many([scenes[s][0][i] for i in ids], [scenes[s][1][i] for i in ids])):
                handed[(s, i)] = c.eqv.clone()
        step = lambda: D.run_plan(eng, plan, inputs, transfers, rank, exchange=HandedOver(handed) if transfers else None)
        step(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t) / steps)
        pairs.append(sum(b - a for _, a, b in plan))
        clouds.append(totals['extractions_per_rank'][rank] if 'extractions_per_rank' in totals else sum(len({int(i) for pr in scenes[s][3][a:b] for i in pr}) for s, a, b in plan))
        del scenes, handed
        torch.cuda.empty_cache()
    if world == 1:
        t1 = times[0]
    t1 = t1 if t1 is not None else float('nan')
    print(f'N={world} ({"exchange" if exchange else "replicated extraction"}): per-rank step time {["%.0f" % (1e3 * x) for x in times]} ms, pairs {pairs}, clouds extracted {clouds} '
          f'(sum {sum(clouds)});  throughput {1623 / max(times):.0f} pairs/s,  efficiency {t1 / (world * max(times)):.3f}', flush=True)
