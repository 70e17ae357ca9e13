import sys, time, types, zlib, threading, queue, hashlib
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from roreg_amd import hip, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
args = types.SimpleNamespace(workload='3dmatch-full', kpts=5000, pair_lists='banded')
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo')
gf = name2network['GF_test'](cfg); gf.load_state_dict(synth.seeded_state_dict(gf, 101)); gf = gf.cuda().eval()
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202)); et = et.cuda().eval()
engs = [RegistrationEngine(cfg, gf, et) for _ in range(nthreads)]
scenes, plan, totals = bench.build_workload(args, 0, 1)
seeds = {s: [(7 + zlib.crc32(f'{s}:{a}:{b}'.encode())) % (2 ** 32) for a, b in scenes[s][3]] for s in scenes}
results = {}
def run(k, p):
    s, a, b = p
    feats, keys, _, pl = scenes[s]
    results[p] = engs[k].run_scene(feats, keys, pl[a:b], pair_seeds=seeds[s][a:b])
def step():
    if nthreads == 1:
        for p in plan: run(0, p)
        return
    q = queue.Queue()
    for p in plan: q.put(p)
    def worker(k):
        while True:
            try: p = q.get_nowait()
            except queue.Empty: return
            run(k, p)
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
    for t in ts: t.start()
    for t in ts: t.join()
step(); torch.cuda.synchronize()
hip.PROFILE = []
t = time.perf_counter()
n = 3
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
prof = hip.PROFILE; hip.PROFILE = None
big = [e0.elapsed_time(e1) for (tag, e0, e1) in prof if tag[0] == 'irrep_gemm_f16x2' and tag[2] * tag[3] == 256 * 512]
h = hashlib.sha1()
for p in plan:
    for r in results[p]: h.update(np.asarray(r.trans).tobytes()); h.update(str((r.id0, r.id1, r.n_match, r.recalltime)).encode())
print(f'{nthreads} host thread(s), one stream: {1e3 * dt:.0f} ms per step = {1623 / dt:.0f} pairs/s;  BIG GEMM by events: {len(big)} launches, average {np.mean(big):.3f} ms;  results {h.hexdigest()[:12]}')
