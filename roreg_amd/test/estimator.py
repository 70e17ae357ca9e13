"""Stage 4: transformation estimation (mirror of test/estimator.py:14-454).

  extractor_dr_index   -- coarse rotation index per correspondence (Des2R 60x60 cross-correlation argmax)
  extractor_localtrans -- ET network + assembly of one local rigid transform per correspondence
  yohoo_ransac / yohoo -- one-shot RANSAC over those transforms + 2 weighted-Kabsch refinements
  yohoc_ransac / yohoc -- rotation-bin-restricted 3-point RANSAC
  refiner, R_pre_log

Files: match_{k}/DR_index/{a}-{b}.npy [M] int64, match_{k}/Trans_pre/{a}-{b}.npy [M,3,4] f64,
match_{k}/{yohoo|yohoc}/{it}iters/{a}-{b}.npz {trans [4,4] f64, recalltime}, .../pre.log."""
import os

import numpy as np
import torch
from tqdm import tqdm

from .. import hip
from ..group import tables
from ..network import name2network
from ..utils.r_eval import compute_R_diff
from ..utils.utils import make_non_exists_dir, load_checkpoint
from . import _cache
from .extractor import scene_feature_name


def pre_log_entry(id0, id1, n_clouds, trans):
    """One record of pre.log (the format utils/RR_cal.py:66-97 reads back): a header line `id0 <tab> id1 <tab> #clouds of the
    scene`, then the four rows of the 4x4 estimate, tab-separated, every number in Python's shortest round-trip repr of the float64; the last
    row is the constant `0.0 0.0 0.0 1.0`."""
    rows = [[repr(float(v)) for v in trans[r][:4]] for r in range(3)] + [['0.0', '0.0', '0.0', '1.0']]
    return '\t'.join([str(int(id0)), str(int(id1)), str(int(n_clouds))]) + '\n' + ''.join('\t'.join(r) + '\n' for r in rows)


def R_pre_log(dataset, save_dir):
    """pre.log of a scene from the per-pair result files {save_dir}/{id0}-{id1}.npz, in pair-list order (test/estimator.py:14-26)."""
    with open(f'{save_dir}/pre.log', 'w') as out:
        for id0, id1 in dataset.pair_ids:
            estimate = np.load(f'{save_dir}/{id0}-{id1}.npz', allow_pickle=True)['trans']
            out.write(pre_log_entry(id0, id1, len(dataset.pc_ids), estimate))


def _dev64(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float64)).cuda()


def _is_f32(scores):
    """True for the rotation-coherence matcher's float32 score arrays (test/matcher.py:210): the reference then sums and normalises them
    in float32 (numpy's pairwise reduction), and the kernels follow (hip.ransac_score / hip.refine `w_f32`)."""
    return getattr(scores, 'dtype', None) == np.float32


def _select_top(scores, match_n):
    """estimator.py:415-421: the top-`match_n` share (or count) of correspondences by score."""
    num = max(scores.shape[0] * match_n, 10) if match_n < 0.999 else match_n
    return np.argsort(scores)[-int(num):]


def _kabsch_host(stats):
    try:
        return hip.kabsch_from_stats(stats)
    except np.linalg.LinAlgError:                   # zero inliers: NaN statistics
        T = np.full((4, 4), np.nan); T[3] = [0, 0, 0, 1]
        return T


def refine_twice(k0, k1, w, ird, T_in=None, Trans=None, hyp_rows=None, best=None, w_f32=False):
    """The two refinements of estimator.py:438-439.  The M-sized reductions (inlier test, weighted centroids,
    3x3 cross-covariance) run on the device; the closing 3x3 SVD is the reference's own LAPACK call on the host,
    so that rank-deficient cases (<= 2 inliers) give the reference's value too."""
    _, st = hip.refine(k0, k1, w, ird * 2.0, T_in=T_in, Trans=Trans, hyp_rows=hyp_rows, best=best, want_stats=True, w_f32=w_f32)
    T1 = _kabsch_host(st)
    _, st = hip.refine(k0, k1, w, ird, T_in=_dev64(T1), want_stats=True, w_f32=w_f32)
    return _kabsch_host(st)


class refiner:
    def Refine_trans(self, key_m0, key_m1, T, scores, inlinerdist=None):
        """Weighted Kabsch on the inliers of T at `inlinerdist` (estimator.py:53-72) -> [4,4] float64."""
        T = np.asarray(T, np.float64)
        T4 = np.zeros((4, 4)); T4[:T.shape[0], :] = T
        _, st = hip.refine(_dev64(key_m0), _dev64(key_m1), _dev64(scores), inlinerdist, T_in=_dev64(T4), want_stats=True, w_f32=_is_f32(scores))
        return _kabsch_host(st)


class extractor_dr_index:
    def __init__(self, cfg):
        self.cfg = cfg
        tables(self.cfg.SO3_related_files)           # validates the files when the directory exists

    def Batch_Des2R_torch(self, des1_eqv, des2_eqv):  # beforerot afterrot
        return hip.des2r(des1_eqv.to('cuda', torch.float32).contiguous(), des2_eqv.to('cuda', torch.float32).contiguous())

    def Des2R_torch(self, des1_eqv, des2_eqv):
        return self.Batch_Des2R_torch(des1_eqv[None], des2_eqv[None])[0]

    def Rindex(self, dataset, keynum):
        match_dir = f'{self.cfg.output_cache_fn}/{dataset.name}/match_{keynum}'
        Save_dir = f'{match_dir}/DR_index'
        make_non_exists_dir(Save_dir)
        datasetname = scene_feature_name(dataset)
        Feature_dir = f'{self.cfg.output_cache_fn}/{datasetname}/YOHO_Output_Group_feature'
        print(f'extract the drindex of the matches on {dataset.name}')
        for pair in tqdm(dataset.pair_ids):
            id0, id1 = pair
            match_pps = torch.from_numpy(np.load(f'{match_dir}/{id0}-{id1}.npy').astype(np.int64)).cuda()
            ft = _cache.feat_dtype(self.cfg)
            feats0 = _cache.load_device(f'{Feature_dir}/{id0}.npy', ft)
            feats1 = _cache.load_device(f'{Feature_dir}/{id1}.npy', ft)
            # irrep-domain bound + exact re-check of near ties: the literal arg-max with ~10x fewer operations (coefficients cached per cloud)
            pre_idxs = hip.des2r(feats1, feats0, rows1=match_pps[:, 1].contiguous(), rows0=match_pps[:, 0].contiguous(),
                                 coefs1=_cache.load_coefs(f'{Feature_dir}/{id1}.npy', ft), coefs0=_cache.load_coefs(f'{Feature_dir}/{id0}.npy', ft))
            np.save(f'{Save_dir}/{id0}-{id1}.npy', pre_idxs.cpu().numpy())


def three_point_transforms(kps0, kps1):
    """Threepps2Tran (estimator.py:139-147) for a stack of triples [H,3,3] -> [H,3,4]: the same numpy / LAPACK calls per triple, so
    the result is bitwise the per-triple one.  That matters: the cross-covariance of three centred points has rank <= 2, so the
    sign of `VT.T @ U.T`'s third direction (rotation or reflection) is whatever LAPACK's null-space vectors give -- the reference's
    hypotheses ARE LAPACK's, and only the same call reproduces them."""
    c0 = np.mean(kps0, 1, keepdims=True)
    c1 = np.mean(kps1, 1, keepdims=True)
    m = (kps1 - c1).transpose(0, 2, 1) @ (kps0 - c0)
    U, S, VT = np.linalg.svd(m)
    R = VT.transpose(0, 2, 1) @ U.transpose(0, 2, 1)
    offset = c0 - (c1 @ R.transpose(0, 2, 1))
    return np.concatenate([R, offset.transpose(0, 2, 1)], 2)


def dr_bins(Index):
    """DR_statictic (estimator.py:119-137) without the per-correspondence Python loop -> (counts [60], members, prob or None).
    members[starts[r] + p] is the p-th correspondence (increasing order) whose coarse rotation is r."""
    Index = np.asarray(Index, np.int64)
    counts = np.bincount(Index, minlength=60)
    num = counts / 100.0
    prob = np.where(counts < 2, 0.0, num * (num - 0.01) * (num - 0.02))
    members = np.argsort(Index, kind='stable')
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    if np.sum(prob) == 0:
        return counts, members, starts, np.zeros(60)
    return counts, members, starts, prob / np.sum(prob)


def yohoc_draws(Index, max_iter, rng=None):
    """The draws of the YOHO-C loop (estimator.py:220-230) -> rows [H,3] of the correspondence list, or None when no rotation bin holds
    two correspondences (:214-216).  The loop's control flow depends only on the global generator and the bin statistics, never on
    the overlaps, so every hypothesis is drawn first (hip.yohoc_draw: the reference's generator calls, replayed) and all of them are
    scored in one device launch.  rng=None: the process-global generator (the reference's); a RandomState: that stream, the global one
    untouched."""
    counts, members, starts, prob = dr_bins(Index)
    if np.sum(prob) < 1e-5:
        return None
    bins, picks = hip.yohoc_draw(prob, counts, max_iter, rng=rng)
    return members[starts[bins][:, None] + picks]


def yohoc_hypotheses(Index, Keys_m0, Keys_m1, max_iter):
    """-> hypotheses [H,3,4] f64 of one pair, or None (see yohoc_draws)."""
    idxs = yohoc_draws(Index, max_iter)
    return None if idxs is None else three_point_transforms(Keys_m0[idxs], Keys_m1[idxs])


class yohoc_ransac:
    def __init__(self, cfg):
        self.cfg = cfg
        self.inliner_dist = cfg.ransac_ird
        self.refiner = refiner()

    def DR_statictic(self, DR_indexs):
        """(correspondences per coarse rotation as {rotation: [rows]}, sampling probability per rotation) -- estimator.py:119-137;
        (None, zeros) when no rotation collects two correspondences."""
        counts, members, starts, prob = dr_bins(np.asarray(DR_indexs).reshape(-1))
        if not prob.any():
            return None, prob
        return {r: members[starts[r]:starts[r] + counts[r]].tolist() for r in range(60)}, prob

    def Threepps2Tran(self, kps0_init, kps1_init):
        return three_point_transforms(np.asarray(kps0_init)[None], np.asarray(kps1_init)[None])[0]          # 3*4

    def overlap_cal(self, key_m0, key_m1, T, scores):
        ov, _, _ = hip.ransac_score(_dev64(key_m0), _dev64(key_m1), _dev64(scores), _dev64(np.asarray(T)[None, :3, :]), self.inliner_dist,
                                    w_f32=_is_f32(scores))
        return float(ov[0].item())

    def transdiff(self, gt, pre):
        Rdiff = compute_R_diff(gt[0:3:, 0:3], pre[0:3:, 0:3])
        tdiff = np.sqrt(np.sum(np.square(gt[0:3, 3] - pre[0:3, 3])))
        return Rdiff, tdiff

    def ransac_once(self, dataset, keynum, max_iter, pair):
        match_dir = f'{self.cfg.output_cache_fn}/{dataset.name}/match_{keynum}'
        Index_dir = f'{match_dir}/DR_index'
        Save_dir = f'{match_dir}/yohoc/{max_iter}iters'
        id0, id1 = pair
        Keys0 = dataset.get_kps(id0)
        Keys1 = dataset.get_kps(id1)
        scores = np.load(f'{match_dir}/scores/{id0}-{id1}.npy')
        pps = np.load(f'{match_dir}/{id0}-{id1}.npy')
        Keys_m0_init = Keys0[pps[:, 0]]
        Keys_m1_init = Keys1[pps[:, 1]]
        sample_index = np.arange(pps.shape[0])
        if self.cfg.RM:
            sample_index = _select_top(scores, self.cfg.match_n)
        Index = np.load(f'{Index_dir}/{id0}-{id1}.npy')[sample_index]
        hyps = yohoc_hypotheses(Index, Keys_m0_init[sample_index], Keys_m1_init[sample_index], max_iter)
        if hyps is None:
            np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=np.random.rand(4, 4), center=np.ones([6, 3]), recalltime=50000)
            return 0
        k0 = _dev64(Keys_m0_init); k1 = _dev64(Keys_m1_init); w = _dev64(scores)
        Trans = _dev64(hyps)
        _, best, _ = hip.ransac_score(k0, k1, w, Trans, self.inliner_dist, w_f32=_is_f32(scores))
        T2 = refine_twice(k0, k1, w, self.inliner_dist, Trans=Trans, best=best, w_f32=_is_f32(scores))
        recall_time = int(best.item()) + 1                      # iter_ransac is 1-based when recorded
        np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=T2, recalltime=recall_time)

    def ransac(self, dataset, keynum, max_iter=1000):
        match_dir = f'{self.cfg.output_cache_fn}/{dataset.name}/match_{keynum}'
        Save_dir = f'{match_dir}/yohoc/{max_iter}iters'
        make_non_exists_dir(Save_dir)
        print(f'Ransac with YOHO-C on {dataset.name}:')
        # the reference forks one process per pair (Pool(len(pair_ids)), estimator.py:258-262); hypothesis scoring is a
        # single kernel launch per pair here, so the pairs simply run in order on the device
        for pair in tqdm(dataset.pair_ids):
            self.ransac_once(dataset, keynum, max_iter, pair)
        R_pre_log(dataset, Save_dir)
        print('Done')


class yohoc:
    def __init__(self, cfg):
        self.rind_extractor = extractor_dr_index(cfg)
        self.ransacer = yohoc_ransac(cfg)

    def run(self, dataset, keynum, max_iter):
        self.rind_extractor.Rindex(dataset, keynum)
        self.ransacer.ransac(dataset, keynum, max_iter)


class extractor_localtrans():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network['ET_test'](self.cfg)
        self.best_model_fn = f'{self.cfg.model_fn}/ET/model_best.pth'
        self.Rgroup = tables(self.cfg.SO3_related_files).R.astype(np.float32)
        self.test_batch_size = self.cfg.bs_ET

    def _load_model(self):
        if os.path.exists(self.best_model_fn):
            checkpoint = load_checkpoint(self.best_model_fn)
            self.network.load_state_dict(checkpoint['network_state_dict'], strict=False)
        else:
            raise ValueError("No model exists")

    def Rt_pre(self, dataset, keynum):
        self._load_model()
        self.network.eval()
        match_dir = f'{self.cfg.output_cache_fn}/{dataset.name}/match_{keynum}'
        DRindex_dir = f'{match_dir}/DR_index'
        Save_dir = f'{match_dir}/Trans_pre'
        make_non_exists_dir(Save_dir)
        datasetname = scene_feature_name(dataset)
        FCGF_dir = f'{self.cfg.output_cache_fn}/{datasetname}/{self.cfg.backbone}_Input_Group_feature'
        YOMO_dir = f'{self.cfg.output_cache_fn}/{datasetname}/YOHO_Output_Group_feature'
        print(f'Extracting the local transformation on each correspondence of {dataset.name}')
        for pair in tqdm(dataset.pair_ids):
            id0, id1 = pair
            pps = torch.from_numpy(np.load(f'{match_dir}/{id0}-{id1}.npy').astype(np.int64)).cuda()
            rows0 = pps[:, 0].contiguous(); rows1 = pps[:, 1].contiguous()
            ft = _cache.feat_dtype(self.cfg)
            f0_in = _cache.load_device(f'{FCGF_dir}/{id0}.npy', ft); f1_in = _cache.load_device(f'{FCGF_dir}/{id1}.npy', ft)
            f0_out = _cache.load_device(f'{YOMO_dir}/{id0}.npy', ft); f1_out = _cache.load_device(f'{YOMO_dir}/{id1}.npy', ft)
            Index_pre = torch.from_numpy(np.load(f'{DRindex_dir}/{id0}-{id1}.npy').astype(np.int64)).cuda()
            keys0 = _dev64(dataset.get_kps(id0)); keys1 = _dev64(dataset.get_kps(id1))
            outs = []
            for start in range(0, pps.shape[0], self.test_batch_size):
                sl = slice(start, start + self.test_batch_size)
                # feats1 is the "before rotation" side: it is the one permuted by the anchor (estimator.py:293-306)
                x = hip.et_gather(f0_in, f1_in, f0_out, f1_out, Index_pre[sl].contiguous(), rows0=rows0[sl].contiguous(), rows1=rows1[sl].contiguous())
                with torch.no_grad():
                    q = self.network.trunk_and_head(x)
                outs.append(hip.quat_to_trans(q, Index_pre[sl].contiguous(), keys0, keys1, rows0=rows0[sl].contiguous(), rows1=rows1[sl].contiguous()))
            Trans = torch.cat(outs, 0).cpu().numpy() if outs else np.zeros((0, 3, 4))
            np.save(f'{Save_dir}/{id0}-{id1}.npy', Trans)


class yohoo_ransac:
    def __init__(self, cfg):
        self.cfg = cfg
        self.inliner_dist = cfg.ransac_ird
        T = tables(self.cfg.SO3_related_files)
        self.Nei_in_SO3 = T.P.astype(np.float64)
        self.Rgroup = T.R
        self.refiner = refiner()

    def overlap_cal(self, key_m0, key_m1, T, scores):
        if len(key_m0) == 0:
            return float('nan')                                  # 0 / 0 in the reference (estimator.py:396-403)
        ov, _, _ = hip.ransac_score(_dev64(key_m0), _dev64(key_m1), _dev64(scores), _dev64(np.asarray(T)[None, :3, :]), self.inliner_dist,
                                    w_f32=_is_f32(scores))
        return float(ov[0].item())

    def transdiff(self, gt, pre):
        Rdiff = compute_R_diff(gt[0:3:, 0:3], pre[0:3:, 0:3])
        tdiff = np.sqrt(np.sum(np.square(gt[0:3, 3] - pre[0:3, 3])))
        return Rdiff, tdiff

    def ransac(self, dataset, keynum, max_iter=1000):
        match_dir = f'{self.cfg.output_cache_fn}/{dataset.name}/match_{keynum}'
        Trans_dir = f'{match_dir}/Trans_pre'
        Save_dir = f'{match_dir}/yohoo/{max_iter}iters'
        make_non_exists_dir(Save_dir)
        print(f'Ransac with YOHO-O on {dataset.name}:')
        for pair in tqdm(dataset.pair_ids):
            id0, id1 = pair
            Keys0 = dataset.get_kps(id0)
            Keys1 = dataset.get_kps(id1)
            scores = np.load(f'{match_dir}/scores/{id0}-{id1}.npy')
            pps = np.load(f'{match_dir}/{id0}-{id1}.npy')
            if pps.shape[0] == 0:
                # an empty match list (the reference's matcher crashes before writing one, matcher.py:98-107): the engine's result for
                # such a pair -- no hypothesis, no inlier, NaN transform, recalltime 0 -- so the two supported paths agree
                T = np.full((4, 4), np.nan); T[3] = [0.0, 0.0, 0.0, 1.0]
                np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=T, recalltime=0)
                continue
            k0 = _dev64(Keys0[pps[:, 0]]); k1 = _dev64(Keys1[pps[:, 1]])
            Trans = np.load(f'{Trans_dir}/{id0}-{id1}.npy')
            rows = np.arange(Trans.shape[0])
            if self.cfg.RM:
                rows = _select_top(scores, self.cfg.match_n)           # hypotheses only from the best-scored matches
            index = np.arange(rows.shape[0])
            np.random.shuffle(index)                                    # estimator.py:423-425
            hyp = torch.from_numpy(np.ascontiguousarray(rows[index[0:max_iter]], np.int64)).cuda()
            w = _dev64(scores)
            TransD = _dev64(Trans)
            _, best, _ = hip.ransac_score(k0, k1, w, TransD, self.inliner_dist, hyp_rows=hyp, w_f32=_is_f32(scores))
            T2 = refine_twice(k0, k1, w, self.inliner_dist, Trans=TransD, hyp_rows=hyp, best=best, w_f32=_is_f32(scores))
            recall_time = max(int(best.item()), 0)
            np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=T2, recalltime=recall_time)
        R_pre_log(dataset, Save_dir)


class yohoo:
    def __init__(self, cfg):
        self.cfg = cfg
        self.rind_extractor = extractor_dr_index(cfg)
        self.localT_extractor = extractor_localtrans(cfg)
        self.ransacer = yohoo_ransac(cfg)

    def run(self, dataset, keynum, max_iter):
        self.rind_extractor.Rindex(dataset, keynum)
        self.localT_extractor.Rt_pre(dataset, keynum)
        self.ransacer.ransac(dataset, keynum, max_iter)
