"""Stage 2: rotation-guided detector (mirror of test/detector.py:10-47).
Reads YOHO_Output_Group_feature/{pc}.npy, writes det_score/{pc}.npy [N] f32 holding rank/N."""
import os

import numpy as np
import torch
from tqdm import tqdm

from ..group import tables
from ..network import name2network
from ..utils import utils
from .extractor import scene_feature_name


class yoho_det():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network['RD_test'](cfg)
        self.best_model_fn = f'{self.cfg.model_fn}/RD/model_best.pth'
        self.Rgroup = tables(self.cfg.SO3_related_files).R.astype(np.float32)
        self._load_model()

    def _load_model(self):
        if os.path.exists(self.best_model_fn):
            checkpoint = utils.load_checkpoint(self.best_model_fn)
            self.network.load_state_dict(checkpoint['network_state_dict'], strict=True)
        else:
            raise ValueError("No model exists")

    def run(self, dataset):
        self.network.eval()
        datasetname = scene_feature_name(dataset)
        savedir = f'{self.cfg.output_cache_fn}/{datasetname}/det_score'
        utils.make_non_exists_dir(savedir)
        print(f'Evaluating the saliency of points using rotaion guided detector on {dataset.name}')
        for pc_id in tqdm(range(len(dataset.pc_ids))):
            if os.path.exists(f'{savedir}/{pc_id}.npy'):
                continue
            feats = np.load(f'{self.cfg.output_cache_fn}/{datasetname}/YOHO_Output_Group_feature/{pc_id}.npy')
            batch = {'feats': torch.from_numpy(feats.astype(np.float32))}
            with torch.no_grad():
                scores = self.network(batch)['scores'].cpu().numpy()
            # normalization for NMS comparison only (detector.py:45-46)
            argscores = np.argsort(scores)
            scores[argscores] = np.arange(scores.shape[0]) / scores.shape[0]
            np.save(f'{savedir}/{pc_id}.npy', scores)
