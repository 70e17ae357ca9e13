"""Stage 4: transformation estimation (mirror of test/estimator.py:14-454).

  extractor_dr_index   -- coarse rotation index per correspondence (Des2R 60x60 cross-correlation argmax)
  extractor_localtrans -- ET network + assembly of one local rigid transform per correspondence
  yohoo_ransac / yohoo -- one-shot RANSAC over those transforms + 2 weighted-Kabsch refinements
  yohoc_ransac / yohoc -- rotation-bin-restricted 3-point RANSAC
  refiner, R_pre_log

Files: match_{k}/DR_index/{a}-{b}.npy [M] int64, match_{k}/Trans_pre/{a}-{b}.npy [M,3,4] f64,
match_{k}/{yohoo|yohoc}/{it}iters/{a}-{b}.npz {trans [4,4] f64, recalltime}, .../pre.log."""
import numpy as np
import torch
from tqdm import tqdm

from .. import hip
from ..group import tables
from ..network import name2network
from ..utils.r_eval import compute_R_diff
from . import _cache
from .extractor import restore_weights
from ._files import SceneFiles


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


def pose_errors(gt, pre):
    """(rotation error in degrees, translation error = Euclidean distance of the translation columns) between two rigid transforms given
    as 3x4 / 4x4 matrices -- the pair the estimators' `transdiff` methods return (test/estimator.py:155-158, 384-387)."""
    gt, pre = np.asarray(gt), np.asarray(pre)
    shift = gt[:3, 3] - pre[:3, 3]
    return compute_R_diff(gt[:3, :3], pre[:3, :3]), np.sqrt(np.sum(np.square(shift)))


def _dev64(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float64)).cuda()


def _dev_i64(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.int64)).cuda()


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
        files = SceneFiles(self.cfg, dataset, keynum)
        files.make('DR_index')
        print(f'extract the drindex of the matches on {dataset.name}')
        ft = _cache.feat_dtype(self.cfg)
        for id0, id1 in tqdm(dataset.pair_ids):
            rows = _dev_i64(files.load_matches(id0, id1))
            path0, path1 = files.feature(id0), files.feature(id1)
            # irrep-domain bound + exact re-check of near ties: the literal arg-max with ~10x fewer operations (coefficients cached per cloud)
            anchors = hip.des2r(_cache.load_device(path1, ft), _cache.load_device(path0, ft), rows1=rows[:, 1].contiguous(), rows0=rows[:, 0].contiguous(),
                                coefs1=_cache.load_coefs(path1, ft), coefs0=_cache.load_coefs(path0, ft))
            np.save(files.dr_index(id0, id1), anchors.cpu().numpy())


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
        return pose_errors(gt, pre)

    def ransac_once(self, dataset, keynum, max_iter, pair):
        files = SceneFiles(self.cfg, dataset, keynum)
        id0, id1 = pair
        out = files.result('yohoc', max_iter, id0, id1)
        scores = files.load_scores(id0, id1)
        pps = files.load_matches(id0, id1)
        pts0 = dataset.get_kps(id0)[pps[:, 0]]                     # the matched keypoints, all M of them (scored against)
        pts1 = dataset.get_kps(id1)[pps[:, 1]]
        drawn_from = _select_top(scores, self.cfg.match_n) if self.cfg.RM else np.arange(pps.shape[0])
        anchors = np.load(files.dr_index(id0, id1))[drawn_from]
        hyps = yohoc_hypotheses(anchors, pts0[drawn_from], pts1[drawn_from], max_iter)
        if hyps is None:                                           # no rotation bin with two correspondences (estimator.py:214-218)
            np.savez(out, trans=np.random.rand(4, 4), center=np.ones([6, 3]), recalltime=50000)
            return 0
        k0, k1, w, Trans = _dev64(pts0), _dev64(pts1), _dev64(scores), _dev64(hyps)
        _, best, _ = hip.ransac_score(k0, k1, w, Trans, self.inliner_dist, w_f32=_is_f32(scores))
        T2 = refine_twice(k0, k1, w, self.inliner_dist, Trans=Trans, best=best, w_f32=_is_f32(scores))
        np.savez(out, trans=T2, recalltime=int(best.item()) + 1)   # the try count is 1-based when recorded (:241)

    def ransac(self, dataset, keynum, max_iter=1000):
        files = SceneFiles(self.cfg, dataset, keynum)
        files.make(files.result_dir('yohoc', max_iter))
        print(f'Ransac with YOHO-C on {dataset.name}:')
        # the reference forks one process per pair (Pool(len(pair_ids)), estimator.py:258-262); hypothesis scoring is a
        # single kernel launch per pair here, so the pairs simply run in order on the device
        for pair in tqdm(dataset.pair_ids):
            self.ransac_once(dataset, keynum, max_iter, pair)
        R_pre_log(dataset, files.result_dir('yohoc', max_iter))
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
        restore_weights(self.network, self.best_model_fn, strict=False)      # (the reference loads this checkpoint non-strictly, estimator.py:289)

    def batch_create(self, feats0_fcgf, feats1_fcgf, feats0_yomo, feats1_yomo, index_pre, start, end):
        """The ET network's input dictionary for correspondences [start, end) of gathered host arrays (test/estimator.py:293-306; note the
        exchange: cloud 1 is the `*_eqv0` side, the one the anchor permutes).  Rt_pre() itself gathers on the device (roreg_et_gather)
        and never builds this dictionary; the method is here for callers that feed ET_test.forward() the reference's way."""
        cut = lambda a: torch.from_numpy(np.ascontiguousarray(a[start:end], dtype=np.float32))
        return {'before_eqv0': cut(feats1_fcgf), 'before_eqv1': cut(feats0_fcgf), 'after_eqv0': cut(feats1_yomo), 'after_eqv1': cut(feats0_yomo),
                'pre_idx': torch.from_numpy(np.ascontiguousarray(index_pre[start:end]).astype(np.int64))}

    def Rt_pre(self, dataset, keynum):
        self._load_model()
        self.network.eval()
        files = SceneFiles(self.cfg, dataset, keynum)
        files.make('Trans_pre')
        print(f'Extracting the local transformation on each correspondence of {dataset.name}')
        ft = _cache.feat_dtype(self.cfg)
        step = self.test_batch_size
        for id0, id1 in tqdm(dataset.pair_ids):
            pps = _dev_i64(files.load_matches(id0, id1))
            rows0 = pps[:, 0].contiguous(); rows1 = pps[:, 1].contiguous()
            f0_in = _cache.load_device(files.input_feature(id0), ft); f1_in = _cache.load_device(files.input_feature(id1), ft)
            f0_out = _cache.load_device(files.feature(id0), ft); f1_out = _cache.load_device(files.feature(id1), ft)
            anchors = _dev_i64(np.load(files.dr_index(id0, id1)))
            keys0 = _dev64(dataset.get_kps(id0)); keys1 = _dev64(dataset.get_kps(id1))
            outs = []
            for start in range(0, pps.shape[0], step):
                a, r0, r1 = (t[start:start + step].contiguous() for t in (anchors, rows0, rows1))
                # cloud 1 is the "before rotation" side: it is the one permuted by the anchor (estimator.py:293-306)
                x = hip.et_gather(f0_in, f1_in, f0_out, f1_out, a, rows0=r0, rows1=r1)
                with torch.no_grad():
                    q = self.network.trunk_and_head(x)
                outs.append(hip.quat_to_trans(q, a, keys0, keys1, rows0=r0, rows1=r1))
            np.save(files.trans_pre(id0, id1), torch.cat(outs, 0).cpu().numpy() if outs else np.zeros((0, 3, 4)))


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
        return pose_errors(gt, pre)

    def ransac(self, dataset, keynum, max_iter=1000):
        files = SceneFiles(self.cfg, dataset, keynum)
        files.make(files.result_dir('yohoo', max_iter))
        print(f'Ransac with YOHO-O on {dataset.name}:')
        for id0, id1 in tqdm(dataset.pair_ids):
            out = files.result('yohoo', max_iter, id0, id1)
            scores = files.load_scores(id0, id1)
            pps = files.load_matches(id0, id1)
            if pps.shape[0] == 0:
                # an empty match list (the reference's matcher crashes before writing one, matcher.py:98-107): the engine's result for
                # such a pair -- no hypothesis, no inlier, NaN transform, recalltime 0 -- so the two supported paths agree
                T = np.full((4, 4), np.nan); T[3] = [0.0, 0.0, 0.0, 1.0]
                np.savez(out, trans=T, recalltime=0)
                continue
            k0 = _dev64(dataset.get_kps(id0)[pps[:, 0]]); k1 = _dev64(dataset.get_kps(id1)[pps[:, 1]])
            Trans = np.load(files.trans_pre(id0, id1))
            # hypotheses: the local transforms of all matches, or of the best-scored share of them with --RM (:415-421), in the order of
            # one shuffle of the process-global generator, the first max_iter of them (:423-425)
            rows = _select_top(scores, self.cfg.match_n) if self.cfg.RM else np.arange(Trans.shape[0])
            order = np.arange(rows.shape[0])
            np.random.shuffle(order)
            hyp = _dev_i64(rows[order[0:max_iter]])
            w, TransD = _dev64(scores), _dev64(Trans)
            _, best, _ = hip.ransac_score(k0, k1, w, TransD, self.inliner_dist, hyp_rows=hyp, w_f32=_is_f32(scores))
            T2 = refine_twice(k0, k1, w, self.inliner_dist, Trans=TransD, hyp_rows=hyp, best=best, w_f32=_is_f32(scores))
            np.savez(out, trans=T2, recalltime=max(int(best.item()), 0))
        R_pre_log(dataset, files.result_dir('yohoo', max_iter))


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
