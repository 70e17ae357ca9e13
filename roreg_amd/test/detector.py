"""Stage 2 -- rotation-guided detector behind the reference's `yoho_det` interface (test/detector.py:10-47).

    in : {cache}/{scene}/YOHO_Output_Group_feature/{pc}.npy   float32 [N,32,60]
    out: {cache}/{scene}/det_score/{pc}.npy                   float32 [N]: the RANK of the keypoint's saliency divided by N

The rank transform (detector.py:45-46) is what the NMS sampling of the matcher compares; raw scores are never stored."""
import os

import numpy as np
import torch
from tqdm import tqdm

from ..group import tables
from ..network import name2network
from ..utils import utils
from .extractor import scene_feature_name, restore_weights


def rank_scores(raw):
    """scores[argsort(scores)] = arange(N)/N, in place on a copy (detector.py:45-46; numpy's default argsort, as there)."""
    out = np.array(raw, copy=True)
    out[np.argsort(out)] = np.arange(out.shape[0]) / out.shape[0]
    return out


class yoho_det():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network['RD_test'](cfg)
        self.best_model_fn = f'{cfg.model_fn}/RD/model_best.pth'
        self.Rgroup = tables(cfg.SO3_related_files).R.astype(np.float32)
        self._load_model()

    def _load_model(self):
        restore_weights(self.network, self.best_model_fn, strict=True)

    def run(self, dataset):
        self.network.eval()
        scene = scene_feature_name(dataset)
        feat_dir = f'{self.cfg.output_cache_fn}/{scene}/YOHO_Output_Group_feature'
        out_dir = f'{self.cfg.output_cache_fn}/{scene}/det_score'
        utils.make_non_exists_dir(out_dir)
        print(f'Evaluating the saliency of points using rotaion guided detector on {dataset.name}')
        for pc in tqdm(range(len(dataset.pc_ids))):
            if os.path.exists(f'{out_dir}/{pc}.npy'):
                continue
            feats = torch.from_numpy(np.load(f'{feat_dir}/{pc}.npy').astype(np.float32))
            with torch.no_grad():
                raw = self.network({'feats': feats})['scores'].cpu().numpy()
            np.save(f'{out_dir}/{pc}.npy', rank_scores(raw))
