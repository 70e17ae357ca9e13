"""Stage 1 -- group-feature extraction behind the reference's `yoho_des` interface (test/extractor.py:13-60).

    in : {cache}/{scene}/{backbone}_Input_Group_feature/{pc}.npy   float32 [N,32,60]
    out: {cache}/{scene}/YOHO_Output_Group_feature/{pc}.npy         float32 [N,32,60]   (the network's 'eqv'; 'inv' is not stored)

Clouds whose output file exists are skipped, like in the reference."""
import os

import numpy as np
import torch
from tqdm import tqdm

from ..network import name2network
from ..utils.utils import make_non_exists_dir, load_checkpoint


def scene_feature_name(dataset):
    """Feature directory of a scene: the low-overlap split '3dLomatch/x' reuses the clouds of '3dmatch/x' (extractor.py:38-41)."""
    name = dataset.name
    return '3d' + name[4:] if name.startswith('3dLo') else name


def restore_weights(network, path, strict=True):
    """Load 'network_state_dict' from a reference checkpoint; a missing file is the reference's ValueError (extractor.py:26-31)."""
    if not os.path.exists(path):
        raise ValueError("No model exists")
    network.load_state_dict(load_checkpoint(path)['network_state_dict'], strict=strict)
    return network


class yoho_des():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network['GF_test'](cfg)
        self.model_fn = f'{cfg.model_fn}/GF/model.pth'
        self.best_model_fn = f'{cfg.model_fn}/GF/model_best.pth'
        self.test_batch_size = cfg.bs_GF

    def _load_model(self):
        restore_weights(self.network, self.best_model_fn)

    def run(self, dataset):
        self._load_model()
        self.network.eval()
        scene = scene_feature_name(dataset)
        src = f'{self.cfg.output_cache_fn}/{scene}/{self.cfg.backbone}_Input_Group_feature'
        dst = f'{self.cfg.output_cache_fn}/{scene}/YOHO_Output_Group_feature'
        make_non_exists_dir(dst)
        print(f'Extracting the PartI descriptors on {dataset.name}')
        step = self.test_batch_size                      # bounds the activation footprint only: the result does not depend on it
        from ._cache import feat_dtype
        ft = feat_dtype(self.cfg)
        for pc in tqdm(dataset.pc_ids):
            if os.path.exists(f'{dst}/{pc}.npy'):
                continue
            # --dtype bf16: the features live on the device in bfloat16 (input rounded once here, output rounded by the network's last
            # kernel); the .npy contract stays float32 (the stored values are then exactly representable in bfloat16)
            x = torch.from_numpy(np.load(f'{src}/{pc}.npy').astype(np.float32)).cuda().to(ft)
            with torch.no_grad():
                chunks = [self.network.PartI_net(x[i:i + step], want_inv=False, out_dtype=ft)['eqv'] for i in range(0, x.shape[0], step)]
            np.save(f'{dst}/{pc}.npy', torch.cat(chunks, 0).float().cpu().numpy())
