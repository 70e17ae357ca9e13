"""Orchestration + metrics (mirror of test/evaluator.py:13-145): runs the four stages per scene, then FMR / IR /
RR(pointdsc) / RR(predator) and appends {base_dir}/results.log in the reference's text format."""
import numpy as np

from ..utils import RR_cal
from ..utils.r_eval import compute_R_diff
from ..utils.utils import transform_points
from . import name2extractor, name2detector, name2matcher, name2estimator


class yoho_evaluator:
    def __init__(self, cfg):
        self.cfg = cfg
        self.GF = self.cfg.GF
        self.RD = self.cfg.RD
        self.RM = self.cfg.RM
        self.ET = self.cfg.ET
        self.keynum = self.cfg.keynum
        self.max_iter = self.cfg.max_iter
        self.extractor = name2extractor[self.GF](self.cfg)
        self.detector = None
        self.matcher = name2matcher['matmul'](self.cfg)
        self.estimator = name2estimator[self.ET](self.cfg)
        if self.RD:
            self.RD = 'yoho_det'
            self.detector = name2detector['yoho_det'](self.cfg)
        else:
            self.RD = 'nodet'
        if self.RM:
            self.RM = 'yoho_mat'
            self.matcher = name2matcher['yoho_mat'](self.cfg)
        else:
            self.RM = 'matmul'

    def process_scene(self, dataset):
        self.extractor.run(dataset)
        if self.detector is not None:
            self.detector.run(dataset)
        self.matcher.run(dataset, self.keynum)
        self.estimator.run(dataset, self.keynum, self.max_iter)

    def fmr_ir_scene(self, dataset):
        fmrs, irs = [], []
        for pair in dataset.pair_ids:
            id0, id1 = pair
            corr = np.load(f'{self.cfg.output_cache_fn}/{dataset.name}/match_{self.keynum}/{id0}-{id1}.npy')
            corr_s = np.load(f'{self.cfg.output_cache_fn}/{dataset.name}/match_{self.keynum}/scores/{id0}-{id1}.npy')
            if self.cfg.RM:
                num = max(corr_s.shape[0] * self.cfg.match_n, 10) if self.cfg.match_n < 0.999 else self.cfg.match_n
                corr = corr[np.argsort(corr_s)[-int(num):]]
            keysm0 = dataset.get_kps(id0)[corr[:, 0]]
            keysm1 = dataset.get_kps(id1)[corr[:, 1]]
            gt = dataset.get_transform(id0, id1)
            keysm1 = transform_points(keysm1, gt)
            dist = np.sqrt(np.sum(np.square(keysm0 - keysm1), axis=-1))
            ir = np.mean(dist < self.cfg.tau_2)
            irs.append(ir)
            fmrs.append(1 if ir > self.cfg.tau_1 else 0)
        return np.mean(np.array(fmrs)), np.mean(np.array(irs))

    def rr_scene(self, dataset):
        rrs, rre, rte = [], [], []
        for pair in dataset.pair_ids:
            id0, id1 = pair
            gt = dataset.get_transform(id0, id1)
            trans = np.load(f'{self.cfg.output_cache_fn}/{dataset.name}/match_{self.keynum}/{self.ET}/{self.max_iter}iters/{id0}-{id1}.npz')['trans']
            Rpre, tpre = trans[0:3, 0:3], trans[0:3, -1]
            Rgt, tgt = gt[0:3, 0:3], gt[0:3, -1]
            Rdiff = compute_R_diff(Rpre, Rgt)
            tdiff = np.sqrt(np.sum(np.square(tpre - tgt)))
            if (Rdiff < 15) and (tdiff < 0.3):
                rrs.append(1); rre.append(Rdiff); rte.append(tdiff)      # pointdsc: errors over successes only
            else:
                rrs.append(0)
        return np.mean(np.array(rrs)), np.mean(np.array(rre)), np.mean(np.array(rte))

    def run(self, datasets=None):
        if datasets is None:
            from ..dataops.dataset import get_dataset_name
            datasets = get_dataset_name(self.cfg.testset, self.cfg.origin_data_dir)
        for name, dataset in datasets.items():
            if type(dataset) is str:
                continue
            self.process_scene(dataset)
        fmrs, irs = [], []
        for name, dataset in datasets.items():
            if type(dataset) is str:
                continue
            fmr, ir = self.fmr_ir_scene(dataset)
            fmrs.append(fmr); irs.append(ir)
        fmr = np.mean(np.array(fmrs)); ir = np.mean(np.array(irs))
        rr_dsc, rre_dsc, rte_dsc = [], [], []
        for name, dataset in datasets.items():
            if type(dataset) is str:
                continue
            rr, rre, rte = self.rr_scene(dataset)
            rr_dsc.append(rr); rre_dsc.append(rre); rte_dsc.append(rte)
        rr_dsc = np.mean(np.array(rr_dsc)); rre_dsc = np.mean(np.array(rre_dsc)); rte_dsc = np.mean(np.array(rte_dsc))
        if datasets['wholesetname'] == 'demo':
            rr_predator = 1.0
        else:
            rr_predator, _, _ = RR_cal.benchmark(self.cfg, datasets, self.keynum, self.max_iter, yoho_sign=self.ET)
        datasetname = datasets['wholesetname']
        msg = f'{datasetname}-{self.GF}-{self.RD}-{self.RM}-{self.ET}-{self.keynum}keys-{self.max_iter}iters\n'
        msg += f'feature matching recall          : {fmr:.5f}\n' \
               f'inlier ratio                     : {ir:.5f}\n' \
               f'registration recall(predator)    : {rr_predator:.5f}\n' \
               f'rotation error(pointdsc)         : {rre_dsc:.5f}\n' \
               f'translation error(pointdsc)      : {rte_dsc:.5f}\n' \
               f'registration recall(pointdsc)    : {rr_dsc:.5f}'
        with open(f'{self.cfg.base_dir}/results.log', 'a') as f:
            f.write(msg + '\n')
        print(msg)
        return {'fmr': fmr, 'ir': ir, 'rr_predator': rr_predator, 'rre': rre_dsc, 'rte': rte_dsc, 'rr': rr_dsc}
