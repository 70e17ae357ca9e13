"""Scene orchestration and benchmark metrics behind the reference's `yoho_evaluator` interface (test/evaluator.py:13-145).

`process_scene` chains extractor -> [detector] -> matcher -> estimator on the file-coupled stage classes; `run` then reports
feature-matching recall, inlier ratio, the PointDSC-style registration recall (RRE < 15 deg and RTE < 0.3 m, errors averaged over the
successes) and the Predator / Redwood recall of utils.RR_cal, and appends the reference's text block to {base_dir}/results.log."""
import numpy as np

from ..utils import RR_cal
from ..utils.r_eval import compute_R_diff
from ..utils.utils import transform_points
from . import name2extractor, name2detector, name2matcher, name2estimator

RRE_MAX_DEG, RTE_MAX_M = 15, 0.3            # success thresholds of the pointdsc-style recall (evaluator.py:81)


def _scenes(datasets):
    """The dataset objects of a benchmark dict (its string entries are names, evaluator.py:93-95)."""
    return [d for d in datasets.values() if not isinstance(d, str)]


class yoho_evaluator:
    def __init__(self, cfg):
        self.cfg = cfg
        self.GF, self.ET = cfg.GF, cfg.ET
        self.keynum, self.max_iter = cfg.keynum, cfg.max_iter
        # the flags become the stage names that label the results.log entry (evaluator.py:24-37)
        self.RD = 'yoho_det' if cfg.RD else 'nodet'
        self.RM = 'yoho_mat' if cfg.RM else 'matmul'
        self.extractor = name2extractor[self.GF](cfg)
        self.detector = name2detector[self.RD](cfg) if cfg.RD else None
        self.matcher = name2matcher[self.RM](cfg)
        self.estimator = name2estimator[self.ET](cfg)

    # ---- stages ---------------------------------------------------------------------------------------------
    def process_scene(self, dataset):
        stages = [(self.extractor, ()), (self.detector, ()), (self.matcher, (self.keynum,)), (self.estimator, (self.keynum, self.max_iter))]
        for stage, args in stages:
            if stage is not None:
                stage.run(dataset, *args)

    # ---- metrics --------------------------------------------------------------------------------------------
    def _match_dir(self, dataset):
        return f'{self.cfg.output_cache_fn}/{dataset.name}/match_{self.keynum}'

    def fmr_ir_scene(self, dataset):
        """(feature matching recall, mean inlier ratio) of a scene; with the rotation-coherence matcher only the best-scored
        share `match_n` of the correspondences counts (evaluator.py:55-59)."""
        ratios = []
        for a, b in dataset.pair_ids:
            pairs = np.load(f'{self._match_dir(dataset)}/{a}-{b}.npy')
            if self.cfg.RM:
                w = np.load(f'{self._match_dir(dataset)}/scores/{a}-{b}.npy')
                keep = max(w.shape[0] * self.cfg.match_n, 10) if self.cfg.match_n < 0.999 else self.cfg.match_n
                pairs = pairs[np.argsort(w)[-int(keep):]]
            p0 = dataset.get_kps(a)[pairs[:, 0]]
            p1 = transform_points(dataset.get_kps(b)[pairs[:, 1]], dataset.get_transform(a, b))
            ratios.append(np.mean(np.sqrt(np.sum(np.square(p0 - p1), axis=-1)) < self.cfg.tau_2))
        hits = [1 if r > self.cfg.tau_1 else 0 for r in ratios]
        return np.mean(np.array(hits)), np.mean(np.array(ratios))

    def rr_scene(self, dataset):
        """(registration recall, mean RRE, mean RTE) with the errors averaged over the successful pairs only."""
        ok, e_rot, e_tra = [], [], []
        for a, b in dataset.pair_ids:
            T = np.load(f'{self._match_dir(dataset)}/{self.ET}/{self.max_iter}iters/{a}-{b}.npz')['trans']
            G = dataset.get_transform(a, b)
            dr = compute_R_diff(T[0:3, 0:3], G[0:3, 0:3])
            dt = np.sqrt(np.sum(np.square(T[0:3, -1] - G[0:3, -1])))
            good = (dr < RRE_MAX_DEG) and (dt < RTE_MAX_M)
            ok.append(1 if good else 0)
            if good:
                e_rot.append(dr); e_tra.append(dt)
        return np.mean(np.array(ok)), np.mean(np.array(e_rot)), np.mean(np.array(e_tra))

    def run(self, datasets=None):
        if datasets is None:
            from ..dataops.dataset import get_dataset_name
            datasets = get_dataset_name(self.cfg.testset, self.cfg.origin_data_dir)
        scenes = _scenes(datasets)
        for d in scenes:
            self.process_scene(d)
        per_scene = np.array([self.fmr_ir_scene(d) + self.rr_scene(d) for d in scenes])      # [scene, (fmr, ir, rr, rre, rte)]
        fmr, ir, rr, rre, rte = (np.mean(per_scene[:, c]) for c in range(5))
        whole = datasets['wholesetname']
        rr_predator = 1.0 if whole == 'demo' else RR_cal.benchmark(self.cfg, datasets, self.keynum, self.max_iter, yoho_sign=self.ET)[0]
        rows = [('feature matching recall', fmr), ('inlier ratio', ir), ('registration recall(predator)', rr_predator),
                ('rotation error(pointdsc)', rre), ('translation error(pointdsc)', rte), ('registration recall(pointdsc)', rr)]
        msg = f'{whole}-{self.GF}-{self.RD}-{self.RM}-{self.ET}-{self.keynum}keys-{self.max_iter}iters\n'
        msg += '\n'.join(f'{label:<33}: {value:.5f}' for label, value in rows)
        with open(f'{self.cfg.base_dir}/results.log', 'a') as f:
            f.write(msg + '\n')
        print(msg)
        return {'fmr': fmr, 'ir': ir, 'rr_predator': rr_predator, 'rre': rre, 'rte': rte, 'rr': rr}
