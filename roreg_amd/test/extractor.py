"""Stage 1: group-feature extraction (mirror of test/extractor.py:13-60).
Reads  {cache}/{scene}/{backbone}_Input_Group_feature/{pc}.npy  [N,32,60] f32
writes {cache}/{scene}/YOHO_Output_Group_feature/{pc}.npy        [N,32,60] f32 ('eqv' only)."""
import os

import numpy as np
import torch
from tqdm import tqdm

from ..network import name2network
from ..utils.utils import make_non_exists_dir, load_checkpoint


def scene_feature_name(dataset):
    """'3dLomatch/x' shares the feature directory of '3dmatch/x' (extractor.py:38-41)."""
    return f'3d{dataset.name[4:]}' if dataset.name[0:4] == '3dLo' else dataset.name


class yoho_des():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network['GF_test'](self.cfg)
        self.model_fn = f'{self.cfg.model_fn}/GF/model.pth'
        self.best_model_fn = f'{self.cfg.model_fn}/GF/model_best.pth'
        self.test_batch_size = self.cfg.bs_GF

    def _load_model(self):
        if os.path.exists(self.best_model_fn):
            checkpoint = load_checkpoint(self.best_model_fn)
            self.network.load_state_dict(checkpoint['network_state_dict'])
        else:
            raise ValueError("No model exists")

    def run(self, dataset):
        self._load_model()
        self.network.eval()
        datasetname = scene_feature_name(dataset)
        FCGF_input_dir = f'{self.cfg.output_cache_fn}/{datasetname}/{self.cfg.backbone}_Input_Group_feature'
        YOHO_output_dir = f'{self.cfg.output_cache_fn}/{datasetname}/YOHO_Output_Group_feature'
        make_non_exists_dir(YOHO_output_dir)
        print(f'Extracting the PartI descriptors on {dataset.name}')
        for pc_id in tqdm(dataset.pc_ids):
            if os.path.exists(f'{YOHO_output_dir}/{pc_id}.npy'):
                continue
            Input_feature = np.load(f'{FCGF_input_dir}/{pc_id}.npy')                 # N*32*60
            x = torch.from_numpy(Input_feature.astype(np.float32)).cuda()
            outs = []
            # the batch size only bounds the activation footprint; results do not depend on it
            for start in range(0, x.shape[0], self.test_batch_size):
                with torch.no_grad():
                    outs.append(self.network(x[start:start + self.test_batch_size])['eqv'])
            np.save(f'{YOHO_output_dir}/{pc_id}.npy', torch.cat(outs, 0).cpu().numpy())
