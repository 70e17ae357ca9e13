"""Host-side stand-in for RegistrationEngine, plugged into bench.py through ROREG_BENCH_ENGINE=tests._bench_stub:make by
tests/test_host_logic.py::test_bench_launcher_* -- it lets a CPU-only box drive bench.py's launcher, shard plan, extractor-output
exchange (gloo), result-table collective and max-over-ranks timing.  No kernel runs and bench.py marks its line as such.

A pair's 'registration' is a function of the pair's generator stream and of a checksum of its two clouds' 'extractor outputs', so a wrong
or missing transfer changes the result table."""
import numpy as np
import torch


class Cloud:
    def __init__(self, before, eqv, keys):
        self.before, self.eqv, self.keys = before, eqv, keys


class StubEngine:
    extractions = 0

    def __init__(self, cfg):
        self.cfg = cfg
        self.phase_ms = None

    def _eqv(self, before):
        StubEngine.extractions += 1
        return (before.float() * 2.0 + 1.0).contiguous()

    def extract_many(self, feats_list, keys_list, max_rows=None):
        return [Cloud(f, self._eqv(f), k) for f, k in zip(feats_list, keys_list)]

    def alloc_eqv(self, before):
        return torch.empty(tuple(before.shape), dtype=torch.float32)

    def cloud_from_eqv(self, before, eqv, keys):
        return Cloud(before, eqv, keys)

    def run_scene(self, feats, keys, pair_ids, keynum=None, max_iter=None, keep_matches=False, pair_seeds=None, ready=None, **kw):
        from roreg_amd.engine import PairResult
        have = {} if ready is None else ready
        used = sorted({int(i) for p in pair_ids for i in p})
        todo = [i for i in used if i not in have]
        fresh = dict(zip(todo, self.extract_many([feats[i] for i in todo], [keys[i] for i in todo])))
        if ready is not None:
            ready.update(fresh)
        clouds = {i: (have[i] if i in have else fresh[i]) for i in used}
        out = []
        for q, (a, b) in enumerate(pair_ids):
            rng = np.random.RandomState(pair_seeds[q] if pair_seeds is not None else 0)
            T = np.eye(4); T[:3] = rng.rand(3, 4) + float(clouds[int(a)].eqv.double().sum()) * 1e-6 + float(clouds[int(b)].eqv[0].double().sum()) * 1e-3
            m = torch.from_numpy(np.stack([np.arange(5), np.arange(5)], 1).astype(np.int64))
            out.append(PairResult(a, b, 5, T, int(rng.randint(0, 1000)), matches=m if keep_matches else None, scores=None))
        return out


def make(cfg):
    return StubEngine(cfg)
