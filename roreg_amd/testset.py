"""Row N3 of the scope table: the step immediately upstream of the path -- assembling the [N,32,60] group feature of a cloud's
keypoints from a point-wise backbone evaluated on the 60 icosahedrally rotated copies of the cloud (testset.py:124-181).

The sparse-conv FCGF backbone (MinkowskiEngine) is out of scope; it is a plug-in here:

    backbone(xyz [n,3] float32 ndarray) -> (xyz_down [m,3] float32, feats [m,F] float32)     (host arrays or device tensors)

For every group element g the cloud and its keypoints are rotated by R_g, the backbone runs on the rotated cloud, and each
rotated keypoint takes the feature of its nearest down-sampled point (knn_module.KNN(1), testset.py:170-172) -- that lookup is
roreg_nn_search with F=3.  Output layout = the hot path's input contract: float32 [N, F, 60], group index fastest."""
import numpy as np
import torch

from . import hip
from .group import tables
from .utils.utils import make_non_exists_dir


def assemble_group_features(backbone, points, keypoints, group_dir=None):
    """points [n,3], keypoints [N,3] (float64 like dataset.get_pc / get_kps) -> float32 [N,F,60] (host ndarray)."""
    T = tables(group_dir)
    pts = np.asarray(points, np.float64)
    kps = np.asarray(keypoints, np.float64)
    # the 60 rotated keypoint sets in ONE upload (host float64 products rounded to float32 exactly as testset.py:80-81 does; 3.6 MB at 5000
    # keypoints); the group feature is assembled in place on the device and downloaded once
    kps_all = hip.upload(np.stack([(kps @ T.R[g].T).astype(np.float32) for g in range(60)]))            # [60,N,3]
    out = None
    for g in range(60):
        xyz_g = (pts @ T.R[g].T).astype(np.float32)                      # testset.py:43
        xyz_down, feats = backbone(xyz_g)                                 # the plug-in decides where its outputs live; device tensors are used as they are
        xyz_down = torch.as_tensor(xyz_down, dtype=torch.float32).to('cuda', non_blocking=True).contiguous()
        feats = torch.as_tensor(feats, dtype=torch.float32).to('cuda', non_blocking=True)
        if out is None:
            out = torch.empty((kps.shape[0], feats.shape[1], 60), dtype=torch.float32, device='cuda')
        nn = hip.nn_search(kps_all[g], xyz_down)                          # nearest down-sampled point of every keypoint (KNN(1), testset.py:170-172)
        out[:, :, g] = feats[nn]                                          # [N,F] into group column g
    return out.cpu().numpy()                                              # [N,F,60]


def write_group_features(backbone, dataset, output_cache_fn, backbone_name='FCGF', group_dir=None):
    """Writes {output_cache_fn}/{dataset.name}/{backbone}_Input_Group_feature/{pc}.npy for every cloud of a scene."""
    out = f'{output_cache_fn}/{dataset.name}/{backbone_name}_Input_Group_feature'
    make_non_exists_dir(out)
    for pc_id in dataset.pc_ids:
        f = assemble_group_features(backbone, dataset.get_pc(pc_id), dataset.get_kps(pc_id), group_dir)
        np.save(f'{out}/{pc_id}.npy', f)
