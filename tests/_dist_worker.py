"""Worker of the multi-rank tests: runs roreg_amd.run_distributed.evaluate as rank `rank` of `world` (gloo; with mode 'real' every rank
uses GPU 0) on a deterministic synthetic two-scene dataset and, on rank 0, stores every pair's result.

    python tests/_dist_worker.py ROOT WORKDIR RANK WORLD PORT MODE OUTFILE     (MODE: stub | real)
"""
import os
import sys

import numpy as np
import torch


class StubEngine:
    """Host-only stand-in for RegistrationEngine.run_scene: a pair's 'registration' is a function of the pair's own generator stream
    (pair_seeds) and of its clouds' features, like the real engine's."""

    def run_scene(self, feats, keys, pair_ids, keynum=None, max_iter=None, keep_matches=False, pair_seeds=None, **kw):
        from roreg_amd.engine import PairResult
        out = []
        for q, (a, b) in enumerate(pair_ids):
            if pair_seeds is not None:
                np.random.seed(pair_seeds[q])
            n = feats[int(a)].shape[0]
            s = np.arange(n); np.random.shuffle(s)
            T = np.eye(4); T[:3] = np.random.rand(3, 4) + float(feats[int(a)].sum()) * 1e-3 + float(feats[int(b)][0].sum())
            m = torch.from_numpy(np.stack([s[:5], s[::-1][:5]], 1).astype(np.int64))
            out.append(PairResult(a, b, 5, T, int(s[0]), matches=m, scores=None))
        return out


def build(workdir, real):
    from roreg_amd import synth
    from roreg_amd.parses.parses_test import default_config
    cfg = default_config(output_cache_fn=f'{workdir}/cache', model_fn=f'{workdir}/ckpt', base_dir=workdir, SO3_related_files=None,
                         keynum=96 if real else 16, ET='yohoo', testset='synth')
    n = 128 if real else 16
    ds0 = synth.make_scene(5, n_clouds=10, n_kpts=n, overlap=0.6, name='synth/scene0')      # 45 pairs: cut across the ranks at world 2
    ds1 = synth.make_scene(77, n_clouds=3, n_kpts=n, overlap=0.6, name='synth/scene1')
    for d in (ds0, ds1):
        d.write_inputs(cfg.output_cache_fn)
        d.gt_dir = f'{workdir}/nonexistent/{d.name}/gt.log'
    return cfg, {'wholesetname': 'synth', 'scene0': ds0, 'scene1': ds1}


def main():
    root, workdir, rank, world, port, mode, outfile = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], sys.argv[7]
    sys.path.insert(0, root)
    from roreg_amd import run_distributed as RD, distributed as D, synth
    if world > 1:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = port
        dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg, datasets = build(f'{workdir}/rank{rank}', mode == 'real')
    if mode == 'real':
        from roreg_amd.engine import RegistrationEngine
        from roreg_amd.network import name2network
        torch.cuda.set_device(0)
        gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
        et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
        engine = RegistrationEngine(cfg, gf, et)
    else:
        engine = StubEngine()
    plan = D.shard_scenes({s: len(datasets[s].pair_ids) for s in ('scene0', 'scene1')}, world, {s: len(datasets[s].pc_ids) for s in ('scene0', 'scene1')},
                          pair_lists={s: datasets[s].pair_ids for s in ('scene0', 'scene1')})
    res = RD.evaluate(cfg, datasets, engine, rank=rank, world=world, seed=2024)
    if rank == 0:
        out = {'split': np.int64(sum(1 for r in plan for p in r if p[0] == 'scene0') > 1)}
        for s in ('scene0', 'scene1'):
            for a, b in datasets[s].pair_ids:
                z = np.load(f'{cfg.output_cache_fn}/{datasets[s].name}/match_{cfg.keynum}/{cfg.ET}/{cfg.max_iter}iters/{a}-{b}.npz')
                out[f'{s}_{a}_{b}_trans'] = z['trans']; out[f'{s}_{a}_{b}_recall'] = np.int64(z['recalltime'])
        out['metrics'] = np.array([res['fmr'], res['ir'], res['rr']])
        np.savez(outfile, **out)
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
    print('ok', rank)


if __name__ == '__main__':
    main()
