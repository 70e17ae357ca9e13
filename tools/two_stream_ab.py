"""Do whole scenes overlap usefully on two HIP streams?    The bench's workload (synthetic 3DMatch scenes, mutual matcher +
one-shot estimator) is run (a) the product's way: all scenes software-pipelined on ONE stream (engine.run_scenes), and (b) on TWO streams:
two host threads, each with an engine and a stream of its own, each driving every other scene -- the HBM-bound transforms, kernel tails and
host gaps of one scene under the matrix-core-bound GEMMs of the other.  Alternating runs on one box; pairs/s of both ways.
    python tools/two_stream_ab.py [n_scenes=4] [reps=3]"""
import sys, threading, time, zlib
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo')
gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
jobs = []
for i, s in enumerate(synth.THREEDMATCH_SCENES[:n_scenes]):
    feats, keys, poses = synth.make_scene_device(500 + i, synth.THREEDMATCH_CLOUDS[i], 5000, 0.6)
    pairs = [(str(a), str(b)) for a, b in synth.scene_pair_list(synth.THREEDMATCH_CLOUDS[i], synth.THREEDMATCH_PAIRS[i], 900 + i, locality=8.0)]
    seeds = [(7 + zlib.crc32(f'{s}:{a}:{b}'.encode())) % (2 ** 32) for a, b in pairs]
    jobs.append((feats, keys, pairs, dict(pair_seeds=seeds)))
n_pairs = sum(len(j[2]) for j in jobs)
engines = [RegistrationEngine(cfg, gf, et) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(2)]


def one_stream():
    return engines[0].run_scenes(jobs)


def two_streams():
    out = [None, None]; err = []

    def work(q):
        try:
            with torch.cuda.stream(streams[q]):
                out[q] = engines[1 + q].run_scenes(jobs[q::2])
                streams[q].synchronize()
        except Exception as e:                                      # noqa: BLE001
            err.append(e)
    ts = [threading.Thread(target=work, args=(q,)) for q in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
    if err:
        raise err[0]
    res = [None] * len(jobs)
    res[0::2] = out[0]; res[1::2] = out[1]
    return res


ref = one_stream(); two = two_streams()                              # warm-up + equality of the results
same = all(np.array_equal(a.trans, b.trans, equal_nan=True) and a.recalltime == b.recalltime for ra, rb in zip(ref, two) for a, b in zip(ra, rb))
print(f'{n_scenes} scenes, {n_pairs} pairs; results of the two ways identical: {same}')
for r in range(reps):
    for name, fn in (('one stream ', one_stream), ('two streams', two_streams)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'   rep {r}: {name}: {n_pairs / dt:8.1f} pairs/s ({dt * 1e3:.0f} ms)', flush=True)
