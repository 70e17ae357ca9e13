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
               pair_ids=None, name='synth/scene0', extent=3.0, portable=False):
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
        # rows: R_g^T (x - t); portable: fixed evaluation order instead of a BLAS matmul (full-size golden cases, see below)
        xc = _rows_times_matrix(xw - t, T.R[g]) if portable else (xw - t) @ T.R[g]
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


def config1_pair():
    """BASELINE configs[0] / SURVEY 8(d) "config 1", the plumbing case, as one fixed recipe: np.random.seed(0) and torch.manual_seed(0) streams
    (private generator objects: the process-global ones are untouched); N = 256; feats0 = randn(N,32,60) float32; keys0 uniform in [0,3]^3
    float64; cloud 1 = cloud 0 seen from the pose (group element 7, t = (0.3, -0.2, 0.5)) with its rows permuted and
    feats1 = feats0[:, :, P[7]] + 0.05 randn.  -> SynthScene with the pair ('0', '1'); its ground truth is T(0 <- 1) = [R_7 | t]."""
    import torch
    rs = np.random.RandomState(0)
    gen = torch.Generator().manual_seed(0)
    T = tables()
    N, g, t = 256, 7, np.array([0.3, -0.2, 0.5])
    feats0 = torch.randn(N, 32, G, generator=gen).numpy()
    noise = torch.randn(N, 32, G, generator=gen).numpy()
    keys0 = rs.uniform(0.0, 3.0, (N, 3))
    perm = rs.permutation(N)
    keys1 = np.ascontiguousarray(((keys0 - t) @ T.R[g])[perm])
    feats1 = np.ascontiguousarray((feats0[:, :, T.P[g]] + np.float32(0.05) * noise)[perm].astype(np.float32))
    ds = SynthScene('synth/config1', [keys0, keys1], [feats0, feats1], [(0, np.zeros(3)), (g, t)], [(0, 1)])
    ds.perm = perm
    return ds


# ------------------------------------------------------------------------------------------------
# full-size parity cases (tests/golden/full_*.npz store only the reference's small outputs; the inputs are rebuilt here from the seed)
# ------------------------------------------------------------------------------------------------
# These generators use only stream-stable Generator draws and correctly rounded elementwise numpy arithmetic (+ - * / sqrt written out
# term by term: no BLAS matmul, no einsum, no sin/cos, whose last bit may depend on the host CPU), so the GPU box rebuilds bit-identical
# inputs to the ones the reference was run on in the build container.
def _rows_times_matrix(x, A):
    """x [n,3] @ A [3,3] (or A [n,3,3]) with a fixed evaluation order."""
    A = np.asarray(A)
    col = (lambda j: A[:, j]) if A.ndim == 2 else (lambda j: A[:, :, j])
    out = np.empty_like(x)
    for j in range(3):
        a = col(j)
        a0, a1, a2 = (a[0], a[1], a[2]) if A.ndim == 2 else (a[:, 0], a[:, 1], a[:, 2])
        out[:, j] = (x[:, 0] * a0 + x[:, 1] * a1) + x[:, 2] * a2
    return out


def make_neartie_scene(seed, n_clouds=2, n_kpts=5000, copies=4, name='synth/neartie', extent=3.0):
    """Like make_scene(), but the world has only n_kpts/copies distinct descriptors: every cloud holds `copies` perturbed copies of each
    (perturbation scales 0, 1e-6, 1e-4, 1e-2 in turn), so nearest-neighbour searches and 60x60 correlations meet exact duplicates and
    thousands of near ties -- the cases in which summation order and tie-breaking decide an index."""
    rng = np.random.default_rng(int(seed))
    T = tables()
    n_base = n_kpts // copies
    xb = rng.uniform(0.0, extent, (n_base, 3))
    fb = rng.standard_normal((n_base, 32, G)).astype(np.float32)
    fb /= np.sqrt((fb * fb).sum(1, keepdims=True))
    scales = np.array([0.0, 1e-6, 1e-4, 1e-2], np.float32)
    kps, feats, poses = [], [], []
    for c in range(n_clouds):
        src = np.arange(n_kpts) % n_base
        eps = scales[(np.arange(n_kpts) // n_base + c) % len(scales)]
        xw = xb[src] + 0.02 * rng.standard_normal((n_kpts, 3))
        fw = fb[src] + eps[:, None, None] * rng.standard_normal((n_kpts, 32, G)).astype(np.float32)
        g = 0 if c == 0 else int(rng.integers(0, G))
        t = np.zeros(3) if c == 0 else rng.uniform(-0.5, 0.5, 3)
        xc = _rows_times_matrix(xw - t, T.R[g])       # rows: R_g^T (x - t)
        fc = np.ascontiguousarray(fw[:, :, T.P[g]])
        perm = rng.permutation(n_kpts)
        kps.append(np.ascontiguousarray(xc[perm]))
        feats.append(np.ascontiguousarray(fc[perm].astype(np.float32)))
        poses.append((g, t))
    pair_ids = [(a, b) for a in range(n_clouds) for b in range(a + 1, n_clouds)]
    return SynthScene(name, kps, feats, poses, pair_ids)


def make_ransac_case(seed, M=5000, H=1000, outlier=0.6, f32_scores=False, anchor=17):
    """Correspondences + per-correspondence local transforms for the one-shot RANSAC stage at full size:
    -> (k0 [M,3], k1 [M,3], scores [M], Trans [M,3,4], hyp int64 [H])."""
    rng = np.random.default_rng(int(seed))
    T = tables()
    Rgt = T.R[anchor]; tgt = np.array([0.2, -0.4, 0.1])
    k1 = rng.uniform(0, 3, (M, 3))
    k0 = _rows_times_matrix(k1, Rgt.T) + tgt + 0.01 * rng.standard_normal((M, 3))
    bad = rng.random(M) < outlier
    k0[bad] = rng.uniform(0, 3, (int(bad.sum()), 3))
    scores = rng.uniform(0.1, 1.0, M).astype(np.float32) if f32_scores else np.ones(M)
    g = np.where(rng.random(M) < 0.3, anchor, rng.integers(0, G, M))
    # small random rotation from the unit quaternion (1, v) / |(1, v)|  (sqrt and rational arithmetic only)
    v = 0.025 * rng.standard_normal((M, 3))
    nrm = np.sqrt(1.0 + v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])
    w, x, y, z = 1.0 / nrm, v[:, 0] / nrm, v[:, 1] / nrm, v[:, 2] / nrm
    dR = np.empty((M, 3, 3))
    dR[:, 0, 0] = 1 - 2 * (y * y + z * z); dR[:, 0, 1] = 2 * (x * y - z * w); dR[:, 0, 2] = 2 * (x * z + y * w)
    dR[:, 1, 0] = 2 * (x * y + z * w); dR[:, 1, 1] = 1 - 2 * (x * x + z * z); dR[:, 1, 2] = 2 * (y * z - x * w)
    dR[:, 2, 0] = 2 * (x * z - y * w); dR[:, 2, 1] = 2 * (y * z + x * w); dR[:, 2, 2] = 1 - 2 * (x * x + y * y)
    Rg = T.R[g]
    R = np.empty((M, 3, 3))
    for i in range(3):
        for j in range(3):
            R[:, i, j] = (dR[:, i, 0] * Rg[:, 0, j] + dR[:, i, 1] * Rg[:, 1, j]) + dR[:, i, 2] * Rg[:, 2, j]
    Trans = np.zeros((M, 3, 4))
    Trans[:, :, :3] = R
    for i in range(3):
        Trans[:, i, 3] = k0[:, i] - ((k1[:, 0] * R[:, i, 0] + k1[:, 1] * R[:, i, 1]) + k1[:, 2] * R[:, i, 2])
    hyp = rng.permutation(M)[:H].astype(np.int64)
    return k0, k1, scores, Trans, hyp


def make_ransac_tie_case(seed, M=3000, H=600, anchor=17):
    """One-shot RANSAC inputs whose best hypotheses tie at float32 precision (the rotation-coherence matcher's score type): every score
    is float32(0.75 + j 2^-16), j in 0..7, and every hypothesis is the true rotation with a millimetre-level translation jitter, so the
    hypotheses' inlier sets differ by a handful of borderline correspondences and their score sums (~1500) by a few 2^-16 -- below the
    float32 spacing there (1.2e-4).  Which hypothesis the reference's strict `>` scan keeps is then decided by the rounding of numpy's
    PAIRWISE float32 sum and of its float32 division by M (test/estimator.py:381,433); a float64 accumulation picks another one.
    Portable arithmetic only.  -> (k0 [M,3], k1 [M,3], scores float32 [M], Trans [H,3,4], hyp int64 [H] = arange)."""
    rng = np.random.default_rng(int(seed))
    T = tables()
    R = T.R[anchor]; tgt = np.array([0.2, -0.4, 0.1])
    k1 = rng.uniform(0, 3, (M, 3))
    k0 = _rows_times_matrix(k1, R.T) + tgt + 0.045 * rng.standard_normal((M, 3))
    scores = (0.75 + rng.integers(0, 8, M) * 2.0 ** -16).astype(np.float32)
    Trans = np.zeros((H, 3, 4))
    Trans[:, :, :3] = R
    Trans[:, :, 3] = tgt + 1.5e-3 * rng.standard_normal((H, 3))
    return k0, k1, scores, Trans, np.arange(H, dtype=np.int64)


# ------------------------------------------------------------------------------------------------
# benchmark-sized scenes generated on the device (bench.py: 433 clouds = 16.6 GB of group features would take minutes on the host)
# ------------------------------------------------------------------------------------------------
# 3DMatch test split: clouds per scene (dataops/dataset.py:152 `stationnums`) and a 1623-pair split over the scenes (synthetic stand-in for
# the benchmark's gt.log pair lists: kitchen is the largest scene, like in the dataset)
THREEDMATCH_SCENES = ['kitchen', 'sun3d-home_at-home_at_scan1_2013_jan_1', 'sun3d-home_md-home_md_scan9_2012_sep_30', 'sun3d-hotel_uc-scan3',
                      'sun3d-hotel_umd-maryland_hotel1', 'sun3d-hotel_umd-maryland_hotel3', 'sun3d-mit_76_studyroom-76-1studyroom2',
                      'sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika']
THREEDMATCH_CLOUDS = [60, 60, 60, 55, 57, 37, 66, 38]
THREEDMATCH_PAIRS = [449, 217, 159, 207, 104, 54, 295, 138]          # sum = 1623


def scene_pair_list(n_clouds, n_pairs, seed, locality=None):
    """A fixed pseudo-random subset of the cloud pairs that touches every cloud (like a scene's gt.log lists its overlapping pairs):
    the chain (0,1), (1,2), ... first, then random further pairs; sorted.  locality=None: the further pairs are uniform over all pairs
    (every contiguous range of the list touches most clouds: the worst case for cutting a scene across ranks); locality=tau: pair (i, j)
    is drawn with probability ~ exp(-|i - j| / tau), as in a scan sequence whose overlapping fragments are mostly temporally close."""
    rng = np.random.default_rng(int(seed))
    allp = [(a, b) for a in range(n_clouds) for b in range(a + 1, n_clouds)]
    chain = [(a, a + 1) for a in range(n_clouds - 1)]
    rest = [p for p in allp if p[1] != p[0] + 1]
    k = max(0, n_pairs - len(chain))
    if locality is None:
        order = rng.permutation(len(rest))
        extra = [rest[i] for i in order[:k]]
    else:
        w = np.array([np.exp(-(b - a) / float(locality)) for a, b in rest]); w /= w.sum()
        extra = [rest[i] for i in rng.choice(len(rest), size=min(k, len(rest)), replace=False, p=w)]
    return sorted((chain + extra)[:n_pairs])


def make_scene_device(seed, n_clouds, n_kpts=5000, overlap=0.6, feat_noise=0.05, coord_noise=0.005, extent=3.0, device='cuda'):
    """make_scene() on the device with torch's generator: -> (feats [n_clouds] of [N,32,60] f32, keys [n_clouds] of [N,3] f64, poses).
    Same construction (shared world points, pose = group element + translation, features permuted by P[g]); not bit-identical to the
    numpy version, which is what the parity tests use."""
    import torch
    T = tables()
    gen = torch.Generator(device=device); gen.manual_seed(int(seed))
    rng = np.random.default_rng(int(seed))
    P = torch.from_numpy(np.ascontiguousarray(T.P, np.int64)).to(device)
    R = torch.from_numpy(np.ascontiguousarray(T.R, np.float64)).to(device)
    n_sh = int(round(overlap * n_kpts)); n_pr = n_kpts - n_sh

    def world(n):
        x = torch.rand((n, 3), generator=gen, device=device, dtype=torch.float64) * extent
        f = torch.randn((n, 32, G), generator=gen, device=device, dtype=torch.float32)
        f /= torch.sqrt((f * f).sum(1, keepdim=True))
        return x, f
    xs, fs = world(n_sh)
    feats, keys, poses = [], [], []
    pool = torch.empty((n_clouds * n_kpts, 32, G), dtype=torch.float32, device=device)      # one allocation: the clouds are views of it
    for c in range(n_clouds):
        xp, fp = world(n_pr)
        xw = torch.cat([xs, xp]); fw = torch.cat([fs, fp])
        g = 0 if c == 0 else int(rng.integers(0, G))
        t = np.zeros(3) if c == 0 else rng.uniform(-0.5, 0.5, 3)
        xc = (xw - torch.from_numpy(t).to(device)) @ R[g]
        if coord_noise > 0:
            xc = xc + coord_noise * torch.randn(xc.shape, generator=gen, device=device, dtype=torch.float64)
        fc = fw[:, :, P[g]] + feat_noise * torch.randn(fw.shape, generator=gen, device=device, dtype=torch.float32)
        perm = torch.randperm(n_kpts, generator=gen, device=device)
        view = pool[c * n_kpts:(c + 1) * n_kpts]
        view.copy_(fc[perm])
        keys.append(xc[perm].contiguous()); feats.append(view); poses.append((g, t))
    return feats, keys, poses


def scene_poses(seed, n_clouds):
    """The poses [(group element, translation)] make_scene_device(seed, n_clouds, ...) gives its clouds -- they come from the numpy generator
    alone, so a rank that does not hold a scene can still rebuild its ground truth (bench.py's accuracy figures on rank 0)."""
    rng = np.random.default_rng(int(seed))
    return [(0, np.zeros(3)) if c == 0 else (int(rng.integers(0, G)), rng.uniform(-0.5, 0.5, 3)) for c in range(n_clouds)]


def pose_transform(poses, id0, id1):
    """[3,4] float64 ground truth x_0 = R x_1 + t of two clouds of a make_scene_device() scene."""
    R = tables().R
    g0, t0 = poses[int(id0)]; g1, t1 = poses[int(id1)]
    return np.concatenate([R[g0].T @ R[g1], (R[g0].T @ (t1 - t0))[:, None]], 1)
