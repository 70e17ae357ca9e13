"""Multi-GPU evaluation driver: one process per GPU, scan pairs sharded by scene, ONE gather of the result table.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m roreg_amd.run_distributed \
           --testset 3dmatch --ET yohoo --keynum 5000 [--RD] [--RM] [--seed 0]

Every rank builds the same shard plan (roreg_amd.distributed.shard_scenes), extracts only the clouds its pair ranges touch,
registers its pairs with the device-resident engine, computes the per-pair inlier ratio locally, and contributes fixed-width
float64 rows to one all_gather (backend nccl = RCCL over xGMI).  Rank 0 then writes the reference's result files
({ET}/{iters}iters/*.npz, pre.log) and computes FMR / IR / RR(pointdsc) / RR(predator) like test/evaluator.py:103-145.
With --seed every pair draws from its own generator stream (seed + crc32(scene, id0, id1)), and every block scale of the kernels is per
keypoint / per correspondence, so a pair's result is a function of the pair alone: the result table does not depend on the number of
ranks nor on how the shard plan cuts the scenes (tested at world size 1 vs 2)."""
import os
import zlib

import numpy as np
import torch

from . import distributed as D
from .utils import RR_cal
from .utils.r_eval import compute_R_diff
from .utils.utils import make_non_exists_dir, transform_points, load_checkpoint
from .test.estimator import pre_log_entry


def build_engine(cfg):
    from .engine import RegistrationEngine
    from .network import name2network

    def load(kind, sub, strict=True):
        net = name2network[kind](cfg)
        fn = f'{cfg.model_fn}/{sub}/model_best.pth'
        if not os.path.exists(fn):
            raise ValueError("No model exists")
        net.load_state_dict(load_checkpoint(fn)['network_state_dict'], strict=strict)
        return net.eval()
    gf = load('GF_test', 'GF')
    et = load('ET_test', 'ET', strict=False) if cfg.ET == 'yohoo' else None     # yohoc never evaluates the ET network (estimator.py:266-272)
    rd = load('RD_test', 'RD') if cfg.RD else None
    rm = load('RM_test', 'RM') if cfg.RM else None
    return RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)


def _feature_dir(cfg, dataset):
    name = f'3d{dataset.name[4:]}' if dataset.name[0:4] == '3dLo' else dataset.name
    return f'{cfg.output_cache_fn}/{name}/{cfg.backbone}_Input_Group_feature'


def inlier_ratio(cfg, result, keys0, keys1, gt):
    """Inlier ratio of a pair's (top-scored) correspondences under the ground truth, test/evaluator.py:50-81 (host numpy, float64)."""
    corr = result.matches.cpu().numpy() if torch.is_tensor(result.matches) else np.asarray(result.matches)
    if corr.shape[0] == 0:
        return 0.0
    if cfg.RM and result.scores is not None:
        num = max(result.scores.shape[0] * cfg.match_n, 10) if cfg.match_n < 0.999 else cfg.match_n
        corr = corr[np.argsort(result.scores)[-int(num):]]
    k0 = keys0[corr[:, 0]]; k1 = transform_points(keys1[corr[:, 1]], gt)
    return float(np.mean(np.sqrt(np.sum(np.square(k0 - k1), axis=-1)) < cfg.tau_2))


def scene_metrics(cfg, rows, gt_of):
    """rows: [{'id0','id1','trans','ir'}] of ONE scene in pair-list order; gt_of(id0, id1) -> [3,4] or [4,4] ground truth.
    -> (FMR, IR, RR(pointdsc), mean RRE, mean RTE of the successes) exactly as test/evaluator.py:50-101,111-129."""
    ir_s, ok_s, re_s, te_s = [], [], [], []
    for row in rows:
        T = row['trans']; gt = gt_of(row['id0'], row['id1'])
        ir_s.append(row['ir'])
        if np.isfinite(T).all():
            rd = compute_R_diff(T[0:3, 0:3], gt[0:3, 0:3]); td = np.sqrt(np.sum(np.square(T[0:3, -1] - gt[0:3, -1])))
            good = bool(rd < 15 and td < 0.3)
        else:
            good = False
        ok_s.append(1 if good else 0)
        if good:
            re_s.append(rd); te_s.append(td)
    return (float(np.mean([1 if i > cfg.tau_1 else 0 for i in ir_s])), float(np.mean(ir_s)), float(np.mean(ok_s)),
            float(np.mean(re_s)) if re_s else float('nan'), float(np.mean(te_s)) if te_s else float('nan'))


def evaluate(cfg, datasets, engine, rank=0, world=1, seed=None, exchange=True):
    scenes = [s for s in datasets if s not in ('wholesetname', 'valscenes')]
    pair_counts = {s: len(datasets[s].pair_ids) for s in scenes}
    cloud_counts = {s: len(datasets[s].pc_ids) for s in scenes}
    pair_lists = {s: datasets[s].pair_ids for s in scenes}
    exchange = bool(exchange and world > 1 and hasattr(engine, 'cloud_from_eqv'))
    plan = D.shard_scenes(pair_counts, world, cloud_counts, pair_lists=pair_lists, exchange=exchange)
    transfers = D.exchange_plan(plan, pair_lists)[1] if exchange else []
    inputs = {}

    class LazyFeats:
        """{cloud id: [N,32,60] float32} of one scene, memory-mapped on access and never held: a cloud's file is read when the engine uploads it
        (once per extraction), under the kernels of the scenes already in flight; the host keeps no copy (the page cache does)."""

        def __init__(self, fdir, used):
            self.fdir, self.used = fdir, set(used)

        def __getitem__(self, i):
            if int(i) not in self.used:
                raise KeyError(i)
            return np.load(f'{self.fdir}/{int(i)}.npy', mmap_mode='r')

        def __contains__(self, i):
            return int(i) in self.used

    def scene_inputs(scene):
        """(feats, keys, pair_ids, seeds) of a scene: keypoints (small) are read once, input features stay on disk until they are uploaded"""
        if scene not in inputs:
            ds = datasets[scene]
            used = sorted({int(i) for sc, a, b in plan[rank] if sc == scene for p in ds.pair_ids[a:b] for i in p} |
                          {i for sc, i, src, dst in transfers if sc == scene and rank in (src, dst)})
            seeds = None if seed is None else [(int(seed) + zlib.crc32(f'{scene}:{p0}:{p1}'.encode())) % (2 ** 32) for p0, p1 in ds.pair_ids]
            inputs[scene] = (LazyFeats(_feature_dir(cfg, ds), used), {i: ds.get_kps(str(i)) for i in used}, ds.pair_ids, seeds)
        return inputs[scene]

    rows = []
    for scene, a, b, res in D.run_plan(engine, plan[rank], scene_inputs, transfers, rank, seeded=seed is not None,
                                       keynum=cfg.keynum, max_iter=cfg.max_iter, keep_matches=True):
        ds = datasets[scene]
        keys = scene_inputs(scene)[1]
        for r in res:                                    # inlier ratio of the (top-scored) correspondences, evaluator.py:50-81
            r.ir = inlier_ratio(cfg, r, keys[int(r.id0)], keys[int(r.id1)], ds.get_transform(r.id0, r.id1))
        rows.append(D.pack_rows(scenes.index(scene), res))
    with D.watchdog(D.collective_timeout(600.0), 'gather of the result table'):     # (waits for the slowest rank's whole share)
        table = D.gather_table(np.concatenate(rows, 0) if rows else np.zeros((0, D.ROW)))
    if rank != 0:
        return None
    by_scene = {s: {} for s in scenes}
    for row in D.unpack_rows(table):
        by_scene[scenes[row['scene']]][(row['id0'], row['id1'])] = row
    fmrs, irs, rrs, rres, rtes = [], [], [], [], []
    for s in scenes:
        ds = datasets[s]
        save_dir = f'{cfg.output_cache_fn}/{ds.name}/match_{cfg.keynum}/{cfg.ET}/{cfg.max_iter}iters'
        make_non_exists_dir(save_dir)
        with open(f'{save_dir}/pre.log', 'w') as w:
            for (a, b) in ds.pair_ids:
                row = by_scene[s][(a, b)]
                np.savez(f'{save_dir}/{a}-{b}.npz', trans=row['trans'], recalltime=row['recalltime'])
                w.write(pre_log_entry(a, b, len(ds.pc_ids), row['trans']))
        f, i, r, re, te = scene_metrics(cfg, [by_scene[s][p] for p in ds.pair_ids], ds.get_transform)
        fmrs.append(f); irs.append(i); rrs.append(r); rres.append(re); rtes.append(te)
    out = {'fmr': float(np.mean(fmrs)), 'ir': float(np.mean(irs)), 'rr': float(np.mean(rrs)), 'rre': float(np.mean(rres)),
           'rte': float(np.mean(rtes)), 'pairs': int(table.shape[0])}
    if datasets['wholesetname'] == 'demo' or not all(os.path.exists(datasets[s].gt_dir[:datasets[s].gt_dir.rfind('.')] + '.info') for s in scenes):
        out['rr_predator'] = 1.0 if datasets['wholesetname'] == 'demo' else float('nan')
    else:
        out['rr_predator'] = float(RR_cal.benchmark(cfg, datasets, cfg.keynum, cfg.max_iter, yoho_sign=cfg.ET)[0])
    msg = f"{datasets['wholesetname']}-{cfg.GF}-{'yoho_det' if cfg.RD else 'nodet'}-{'yoho_mat' if cfg.RM else 'matmul'}-{cfg.ET}-{cfg.keynum}keys-{cfg.max_iter}iters\n"
    msg += f"feature matching recall          : {out['fmr']:.5f}\n" \
           f"inlier ratio                     : {out['ir']:.5f}\n" \
           f"registration recall(predator)    : {out['rr_predator']:.5f}\n" \
           f"rotation error(pointdsc)         : {out['rre']:.5f}\n" \
           f"translation error(pointdsc)      : {out['rte']:.5f}\n" \
           f"registration recall(pointdsc)    : {out['rr']:.5f}"
    make_non_exists_dir(cfg.base_dir)
    with open(f'{cfg.base_dir}/results.log', 'a') as f:
        f.write(msg + '\n')
    print(msg)
    return out


def main():
    from .parses.parses_test import build_parser
    from .dataops.dataset import get_dataset_name
    parser = build_parser()
    parser.add_argument('--seed', type=int, default=None, help='one generator stream per pair (results independent of the number of ranks)')
    cfg, _ = parser.parse_known_args()
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1)); local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1 or D.forced():
        D.init_collectives('nccl', rank, world, local)
    datasets = get_dataset_name(cfg.testset, cfg.origin_data_dir)
    evaluate(cfg, datasets, build_engine(cfg), rank, world, cfg.seed)
    if world > 1 or D.forced():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
