"""Synthetic stand-ins for the data the path consumes (there is no dataset or GF/ET checkpoint in the
build container or on the GPU box).

* seeded_state_dict(): deterministic, recipe-generated weights for any module that exposes the
  reference's state_dict key names -- the same recipe is applied to the reference's modules by
  tools/gen_golden.py and to this package's mirrors by the tests, so multi-MB weight tensors never
  have to be committed as fixtures.
* make_scene(): a scene of partially overlapping 5000-keypoint clouds whose FCGF-like group features
  [N,32,60] are rotation-consistent (cloud pose = icosahedral group element + translation), written
  in the on-disk layout of test/extractor.py:42-48 or kept in memory.

Everything uses numpy's PCG64 Generator, which is stream-stable for a fixed numpy version (the GPU
box runs this same image).
"""
import os
import zlib
import numpy as np

from .group import tables, G


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def _rng_for(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def seeded_tensor(seed, name, shape):
    """float32 array for state_dict entry `name` of shape `shape` (recipe keyed on the name suffix)."""
    rng = _rng_for(seed, name)
    shape = tuple(int(s) for s in shape)
    leaf = name.split('.')[-1]
    # ET head: keep the residual rotation small (q ~ (1, eps)) so that, as with a trained network, the local
    # transforms of correct correspondences agree with each other and one-shot RANSAC has real inliers
    if name == 'PartII_To_R_FC.6.weight':
        return rng.normal(0.0, 0.02 / np.sqrt(shape[1]), shape).astype(np.float32)
    if name == 'PartII_To_R_FC.6.bias':
        return np.array([1.0, 0.0, 0.0, 0.0], np.float32) + rng.normal(0.0, 0.005, shape).astype(np.float32)
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, np.int64)
    if leaf == 'running_mean':
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    if leaf == 'running_var':
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == 'bin_score':
        return np.asarray(1.0, np.float32).reshape(shape)
    if leaf == 'weight' and len(shape) == 1:          # BatchNorm gamma
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == 'bias':
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    if leaf == 'weight':                              # conv [O,C,kh,kw]
        fan_in = int(np.prod(shape[1:]))
        return rng.normal(0.0, 1.0 / np.sqrt(fan_in), shape).astype(np.float32)
    raise KeyError(f'no recipe for state_dict entry {name} {shape}')


def seeded_state_dict(module, seed):
    """Overwrite every entry of module.state_dict() by recipe and load it back.  Returns the dict."""
    import torch
    sd = module.state_dict()
    new = {}
    for name, t in sd.items():
        new[name] = torch.from_numpy(seeded_tensor(seed, name, t.shape)).to(t.dtype).reshape(t.shape)
    module.load_state_dict(new, strict=True)
    return new


# ------------------------------------------------------------------------------------------------
# scenes
# ------------------------------------------------------------------------------------------------
class SynthScene:
    """Duck-typed stand-in for dataops/dataset.py:41-129 `ThrDMatchPartDataset`:
    .name .pair_ids .pc_ids .get_kps(id) .get_transform(id0,id1) .gt_dir"""

    def __init__(self, name, kps, feats, poses, pair_ids, gt_dir=None):
        self.name = name
        self.pc_ids = [str(i) for i in range(len(kps))]
        self.pair_ids = [(str(a), str(b)) for a, b in pair_ids]
        self._kps = kps                # list of [N,3] float64
        self.feats = feats             # list of [N,32,60] float32 (FCGF-like group features)
        self.poses = poses             # list of (g_c, t_c):  x_c = R_g^T (x_w - t_c)
        self.gt_dir = gt_dir

    def get_kps(self, pc_id):
        return self._kps[int(pc_id)]

    def get_transform(self, id0, id1):
        """[3,4] float32 T with x_0 = R x_1 + t  (dataops/dataset.py:60-75, get_transform)."""
        R = tables().R
        g0, t0 = self.poses[int(id0)]
        g1, t1 = self.poses[int(id1)]
        Rr = R[g0].T @ R[g1]
        tr = R[g0].T @ (t1 - t0)
        return np.concatenate([Rr, tr[:, None]], 1).astype(np.float32)

    def write_inputs(self, output_cache_fn, backbone='FCGF'):
        d = f'{output_cache_fn}/{self.name}/{backbone}_Input_Group_feature'
        os.makedirs(d, exist_ok=True)
        for i, f in zip(self.pc_ids, self.feats):
            np.save(f'{d}/{i}.npy', f)


def make_scene(seed, n_clouds=2, n_kpts=256, overlap=0.6, feat_noise=0.05, coord_noise=0.0,
               pair_ids=None, name='synth/scene0', extent=3.0):
    """Clouds share a common set of round(overlap*N) world points; the rest are private.
    Cloud c sees world point x_w at x_c = R_{g_c}^T (x_w - t_c) and its group feature as
    f_w[:, :, P[g_c]] + noise, so that descriptors are consistent with the pose (the same
    construction SURVEY.md section 8d config 1 used against the reference)."""
    rng = np.random.default_rng(int(seed))
    T = tables()
    n_sh = int(round(overlap * n_kpts))
    n_pr = n_kpts - n_sh

    def world(n):
        x = rng.uniform(0.0, extent, (n, 3))
        f = rng.standard_normal((n, 32, G)).astype(np.float32)
        f /= np.sqrt((f * f).sum(1, keepdims=True))
        return x, f

    xs, fs = world(n_sh)
    kps, feats, poses = [], [], []
    for c in range(n_clouds):
        xp, fp = world(n_pr)
        xw = np.concatenate([xs, xp], 0)
        fw = np.concatenate([fs, fp], 0)
        g = 0 if c == 0 else int(rng.integers(0, G))
        t = np.zeros(3) if c == 0 else rng.uniform(-0.5, 0.5, 3)
        xc = (xw - t) @ T.R[g]                       # rows: R_g^T (x - t)
        if coord_noise > 0:
            xc = xc + rng.normal(0, coord_noise, xc.shape)
        fc = fw[:, :, T.P[g]] + feat_noise * rng.standard_normal(fw.shape).astype(np.float32)
        perm = rng.permutation(n_kpts)
        kps.append(np.ascontiguousarray(xc[perm]))
        feats.append(np.ascontiguousarray(fc[perm].astype(np.float32)))
        poses.append((g, t))
    if pair_ids is None:
        pair_ids = [(a, b) for a in range(n_clouds) for b in range(a + 1, n_clouds)]
    return SynthScene(name, kps, feats, poses, pair_ids)
