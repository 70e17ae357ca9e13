"""Generate tests/golden/*.npz by importing and running the reference (/root/reference) on CPU.

Runs in the BUILD CONTAINER ONLY: /root/reference does not exist on the GPU box and no reference code
travels -- only the input/output vectors written here (plus the shipped RD/RM weight tensors, which
are data).  The import recipe is SURVEY.md Appendix B: numpy-2 aliases, stub modules for
open3d/nibabel/tensorboardX, identity .cuda(), torch.load(map_location='cpu').

Usage:  python tools/gen_golden.py            (rewrites every fixture; deterministic)
"""
import os
import sys
import types
import shutil
import tempfile

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)

# ---- shims (Appendix B) ---------------------------------------------------------------------------
np.int = int; np.float = float        # (np.bool still exists in numpy 2; overriding it breaks numpy.ma)
for _n in ['open3d', 'nibabel', 'nibabel.quaternions', 'tensorboardX']:
    sys.modules[_n] = types.ModuleType(_n)
sys.modules['nibabel'].quaternions = sys.modules['nibabel.quaternions']
sys.modules['tensorboardX'].SummaryWriter = object
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
_tl = torch.load


def _load(f, *a, **k):
    k.setdefault('map_location', 'cpu'); k.setdefault('weights_only', False)
    return _tl(f, *a, **k)


torch.load = _load
from roreg_amd import synth                      # noqa: E402  (before the reference shadows `test`/`utils`)
from roreg_amd.group import tables               # noqa: E402

sys.path.insert(0, REF)
os.chdir(REF)
from types import SimpleNamespace as NS          # noqa: E402
from network import name2network                 # noqa: E402
from test import name2extractor, name2detector, name2matcher, name2estimator   # noqa: E402
import test.estimator as ref_est                 # noqa: E402
import test.matcher as ref_mat                   # noqa: E402
from test.evaluator import yoho_evaluator        # noqa: E402
from utils.knn_search import knn_module          # noqa: E402
import utils.r_eval as ref_reval                 # noqa: E402
import utils.utils as ref_utils                  # noqa: E402

# numpy-2 shim: r_eval.py:42 uses np.array(copy=False), which now raises when a dtype conversion is needed
_qfm = ref_reval.quaternion_from_matrix
ref_reval.quaternion_from_matrix = lambda m, isprecise=False: _qfm(np.asarray(m, dtype=np.float64), isprecise)

GF_SEED, ET_SEED = 101, 202
torch.set_num_threads(8)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrs)
    sz = os.path.getsize(os.path.join(OUT, name + '.npz'))
    print(f'  wrote {name}.npz  {sz/1024:.0f} KiB  keys={list(arrs)[:8]}{"..." if len(arrs) > 8 else ""}')


def make_cfg(root, RD=False, RM=False, ET='yohoo', keynum=128, match_n=0.5):
    mf = f'{root}/ckpt'
    if not os.path.exists(mf):
        os.makedirs(mf + '/GF'); os.makedirs(mf + '/ET')
        for d in ['RD', 'RM']:
            os.symlink(f'{REF}/checkpoints/FCGF/{d}', f'{mf}/{d}')
    cfg = NS(SO3_related_files=f'{REF}/utils/group_related', model_fn=mf, RM=RM, RD=RD, ET=ET, GF='yoho_des',
             match_n=match_n, ransac_ird=0.1, output_cache_fn=f'{root}/cache', backbone='FCGF', bs_GF=50, bs_ET=40,
             keynum=keynum, max_iter=1000, tau_1=0.05, tau_2=0.1, tau_3=0.2, base_dir=root, testset='synth',
             origin_data_dir=root)
    if not os.path.exists(f'{mf}/GF/model_best.pth'):
        for k, d, s in [('GF_test', 'GF', GF_SEED), ('ET_test', 'ET', ET_SEED)]:
            net = name2network[k](cfg)
            synth.seeded_state_dict(net, s)
            torch.save({'best_para': 0, 'network_state_dict': net.state_dict()}, f'{mf}/{d}/model_best.pth')
    return cfg


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


# ====================================================================================================
def gen_weights():
    """Shipped RD / RM weights (data; network_state_dict only, optimizer state dropped)."""
    for d in ['RD', 'RM']:
        ck = torch.load(f'{REF}/checkpoints/FCGF/{d}/model_best.pth')
        save(f'weights_{d}', **sd_to_np(ck['network_state_dict']))


def gen_gf(cfg):
    net = name2network['GF_test'](cfg); synth.seeded_state_dict(net, GF_SEED); net.eval()
    rng = np.random.default_rng(7)
    x = rng.standard_normal((16, 32, 60)).astype(np.float32)
    x /= np.sqrt((x * x).sum(1, keepdims=True))
    with torch.no_grad():
        out = net(torch.from_numpy(x.copy()))
        # per-layer taps for the group-conv kernel tests
        P1 = net.PartI_net
        t0 = P1.Conv_in(P1.data_process(torch.from_numpy(x.copy())))[:, :, :, 0]
        t1 = P1.SO3_Conv_layers[0](t0)
    save('gf_forward', x=x, eqv=out['eqv'].numpy(), inv=out['inv'].numpy(), conv_in=t0.numpy(), res=t1.numpy(),
         seed=np.int64(GF_SEED))


def gen_rd(cfg):
    net = name2network['RD_test'](cfg)
    ck = torch.load(f'{REF}/checkpoints/FCGF/RD/model_best.pth')
    net.load_state_dict(ck['network_state_dict'], strict=True); net.eval()
    rng = np.random.default_rng(8)
    x = rng.standard_normal((96, 32, 60)).astype(np.float32)
    x /= np.sqrt((x * x).sum(1, keepdims=True))
    with torch.no_grad():
        f = net.eqv_encoder[0](torch.from_numpy(x.copy()))
        s = net({'feats': torch.from_numpy(x.copy())})['scores'].numpy()
    save('rd_forward', x=x, enc=f.numpy(), scores=s)


def gen_knn():
    rng = np.random.default_rng(9)
    out = {}
    for tag, (n, m, f) in {'a': (700, 613, 32), 'b': (1201, 1000, 32), 'c': (37, 5, 32)}.items():
        A = rng.standard_normal((n, f)).astype(np.float32); A /= np.linalg.norm(A, axis=1, keepdims=True)
        B = rng.standard_normal((m, f)).astype(np.float32); B /= np.linalg.norm(B, axis=1, keepdims=True)
        d, idx = knn_module.KNN(1)(torch.from_numpy(A.T[None].copy()), torch.from_numpy(B.T[None].copy()))
        out.update({f'{tag}_target': A, f'{tag}_source': B, f'{tag}_d': d.numpy(), f'{tag}_idx': idx.numpy()})
    # duplicated targets -> first-index tie-break
    A = rng.standard_normal((64, 32)).astype(np.float32); A = np.concatenate([A, A], 0)
    B = A[:64] + 0.01 * rng.standard_normal((64, 32)).astype(np.float32)
    d, idx = knn_module.KNN(1)(torch.from_numpy(A.T[None].copy()), torch.from_numpy(B.T[None].copy()))
    out.update({'tie_target': A, 'tie_source': B, 'tie_d': d.numpy(), 'tie_idx': idx.numpy()})
    # k=5 on 3-D coordinates (NMS use)
    K = rng.uniform(0, 3, (777, 3)).astype(np.float32)
    d, idx = knn_module.KNN(5)(torch.from_numpy(K.T[None].copy()), torch.from_numpy(K.T[None].copy()))
    out.update({'k5_keys': K, 'k5_idx': idx.numpy()})
    save('knn', **out)


def gen_knn_api():
    """The other public methods of modified_knn_matcher (utils/knn_search.py:17-136) and batch_create (test/estimator.py:293-306)."""
    rng = np.random.default_rng(19)
    out = {}
    M = knn_module.KNN(5)
    A = rng.standard_normal((301, 32)).astype(np.float32); A /= np.linalg.norm(A, axis=1, keepdims=True)
    B = rng.standard_normal((450, 32)).astype(np.float32); B /= np.linalg.norm(B, axis=1, keepdims=True)
    B[100:164] = A[:64]                                           # exact hits (distance 0 / sqrt(1e-7)) ...
    B[200:264] = A[:64]                                           # ... twice: first-index ties
    B[300:332] = A[:32] + np.float32(3e-5) * rng.standard_normal((32, 32)).astype(np.float32)   # roots that may round together
    tA, tB = torch.from_numpy(A), torch.from_numpy(B)
    out.update(A=A, B=B)
    for dt in ('L2', 'SquareL2'):
        out[f'pdist_{dt}'] = M.pdist(tA[:40], tB, dist_type=dt).numpy()
        d, i = M.find_nn_gpu(tA, tB, nn_max_n=128, dist_type=dt)
        out[f'nn_d_{dt}'] = d.numpy(); out[f'nn_i_{dt}'] = i.numpy()
        d, i = M.find_knn_gpu(tA, tB, nn_max_n=128, dist_type=dt)
        out[f'knn_d_{dt}'] = d.numpy(); out[f'knn_i_{dt}'] = i.numpy()
        d, i = M(tB.T[None], tA.T[None], dist_type=dt)            # __call__(target [1,f,n], source [1,f,m])
        out[f'call5_d_{dt}'] = d.numpy(); out[f'call5_i_{dt}'] = i.numpy()
        d, i = knn_module.KNN(1)(tB.T[None], tA.T[None], dist_type=dt)
        out[f'call1_d_{dt}'] = d.numpy(); out[f'call1_i_{dt}'] = i.numpy()
    out['nn_i_only'] = M.find_nn_gpu(tA, tB, return_distance=False).numpy()
    K = rng.uniform(0, 3, (500, 3)).astype(np.float32)
    d, i = M.find_knn_gpu(torch.from_numpy(K), torch.from_numpy(K))
    out.update(K=K, knn3_d=d.numpy(), knn3_i=i.numpy())
    np.random.seed(77)
    i0, i1 = M.find_corr(tA, tB, subsample_size=256, mutual=True)
    out.update(corr_i0=i0, corr_i1=i1)
    np.random.seed(78)
    i0, i1 = M.find_corr(tA, tB, subsample_size=-1, mutual=False)
    out.update(corr_nm_i0=i0, corr_nm_i1=i1)
    save('knn_api', **out)
    # batch_create: a pure re-labelling of slices (NB the 0/1 exchange)
    ex = ref_est.extractor_localtrans.__new__(ref_est.extractor_localtrans)
    f = [rng.standard_normal((9, 4, 6)) for _ in range(4)]        # float64 on purpose: the method casts to float32
    idx = rng.integers(0, 60, 9)
    b = ex.batch_create(f[0], f[1], f[2], f[3], idx, 2, 7)
    save('batch_create', f0_fcgf=f[0], f1_fcgf=f[1], f0_yomo=f[2], f1_yomo=f[3], index_pre=idx, **{k: v.numpy() for k, v in b.items()})


def gen_knn_wide():
    """modified_knn_matcher at a feature width and a k the pipeline never uses (utils/knn_search.py:13-162 accepts anything): F = 16 with k = 12, and
    F = 7 with k = 1, both distance types; exact duplicates among the targets (first-index ties)."""
    rng = np.random.default_rng(29)
    out = {}
    A = rng.standard_normal((257, 16)).astype(np.float32)
    B = rng.standard_normal((400, 16)).astype(np.float32)
    B[50:90] = A[:40]; B[150:190] = A[:40]
    A7 = rng.standard_normal((130, 7)).astype(np.float32); B7 = rng.standard_normal((333, 7)).astype(np.float32); B7[10:40] = A7[:30]
    out.update(A=A, B=B, A7=A7, B7=B7)
    tA, tB = torch.from_numpy(A), torch.from_numpy(B)
    M = knn_module.KNN(12)
    for dt in ('L2', 'SquareL2'):
        d, i = M.find_knn_gpu(tA, tB, nn_max_n=100, dist_type=dt)
        out[f'knn_d_{dt}'] = d.numpy(); out[f'knn_i_{dt}'] = i.numpy()
        d, i = M(tB.T[None], tA.T[None], dist_type=dt)
        out[f'call12_d_{dt}'] = d.numpy(); out[f'call12_i_{dt}'] = i.numpy()
        d, i = knn_module.KNN(1).find_nn_gpu(torch.from_numpy(A7), torch.from_numpy(B7), nn_max_n=64, dist_type=dt)
        out[f'nn7_d_{dt}'] = d.numpy(); out[f'nn7_i_{dt}'] = i.numpy()
    save('knn_wide', **out)


def gen_nms():
    rng = np.random.default_rng(10)
    out = {}
    n = 600
    keys = rng.uniform(0, 3, (n, 3))
    raw = rng.standard_normal(n).astype(np.float32)
    scores = raw.copy(); scores[np.argsort(raw)] = np.arange(n) / n      # detector.py:45-46 rank scores
    scores = scores.astype(np.float32)
    out['keys'] = keys; out['scores'] = scores
    for num in [700, 600, 400, 150, 20]:
        idx = ref_mat.NMS_sample(num, 5).sample(keys, scores)
        out[f'idx_{num}'] = np.asarray(idx, np.int64)
    save('nms', **out)


def gen_des2r(cfg):
    T = tables()
    ex = ref_est.extractor_dr_index(cfg)
    rng = np.random.default_rng(11)
    B = 96
    d2 = rng.standard_normal((B, 32, 60)).astype(np.float32); d2 /= np.sqrt((d2 * d2).sum(1, keepdims=True))
    a = rng.integers(0, 60, B)
    # d1 such that Des2R(d1,d2)==a : d1[:,:,P[a,g]] ~ d2[:,:,g]
    d1 = np.empty_like(d2)
    for b in range(B):
        d1[b][:, T.P[a[b]]] = d2[b]
    d1 += 0.3 * rng.standard_normal(d1.shape).astype(np.float32)
    t1, t2 = torch.from_numpy(d1), torch.from_numpy(d2)
    idx = ex.Batch_Des2R_torch(t1, t2).numpy()
    perm = torch.from_numpy(T.P.reshape(-1))
    cor = torch.einsum('bfag,bfg->ba', t1[:, :, perm].reshape(B, 32, 60, 60), t2).numpy()
    save('des2r', d1=d1, d2=d2, idx=idx, cor=cor, planted=a)


def gen_et(cfg):
    net = name2network['ET_test'](cfg); synth.seeded_state_dict(net, ET_SEED); net.eval()
    rng = np.random.default_rng(12)
    B = 40

    def f():
        x = rng.standard_normal((B, 32, 60)).astype(np.float32)
        return x / np.sqrt((x * x).sum(1, keepdims=True))
    b0, b1, a0, a1 = f(), f(), f(), f()
    pre = rng.integers(0, 60, B).astype(np.int64)
    batch = {'before_eqv0': torch.from_numpy(b0.copy()), 'before_eqv1': torch.from_numpy(b1.copy()),
             'after_eqv0': torch.from_numpy(a0.copy()), 'after_eqv1': torch.from_numpy(a1.copy()),
             'pre_idx': torch.from_numpy(pre.copy())}
    with torch.no_grad():
        out = net(batch)
    q = out['quaternion_pre'].numpy()
    Rs = np.stack([ref_reval.matrix_from_quaternion(q[i]) for i in range(B)])
    save('et_forward', before_eqv0=b0, before_eqv1=b1, after_eqv0=a0, after_eqv1=a1, pre_idx=pre,
         quaternion=q, pre_idxs=out['pre_idxs'].numpy(), R_from_q=Rs, seed=np.int64(ET_SEED))


def gen_ransac(cfg):
    """yohoo_ransac / refiner on hand-built inputs, incl. the per-hypothesis overlap vector."""
    T = tables()
    rng = np.random.default_rng(13)
    out = {}
    for tag, (M, outl, f32scores) in {'ones': (400, 0.5, False), 'f32': (257, 0.7, True)}.items():
        Rgt = T.R[17]; tgt = np.array([0.2, -0.4, 0.1])
        k1 = rng.uniform(0, 3, (M, 3))
        k0 = k1 @ Rgt.T + tgt + 0.01 * rng.standard_normal((M, 3))
        bad = rng.random(M) < outl
        k0[bad] = rng.uniform(0, 3, (bad.sum(), 3))
        scores = rng.uniform(0.1, 1.0, M).astype(np.float32) if f32scores else np.ones(M)
        # hypotheses: per-correspondence local transforms (noisy rotations about the truth)
        Trans = np.zeros((M, 3, 4))
        for i in range(M):
            g = 17 if rng.random() < 0.3 else int(rng.integers(0, 60))
            ax = rng.standard_normal(3); ax /= np.linalg.norm(ax); ang = 0.05 * rng.standard_normal()
            Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
            R = dR @ T.R[g]
            Trans[i, :, :3] = R; Trans[i, :, 3] = k0[i] - k1[i] @ R.T
        rs = ref_est.yohoo_ransac(make_cfg_like(cfg, RM=f32scores))
        ov = np.array([rs.overlap_cal(k0, k1, Trans[i], scores) for i in range(M)])
        masks = np.stack([np.sum(np.square(k0 - ref_utils.transform_points(k1, Trans[i])), -1) < 0.1 * 0.1 for i in range(M)])
        best = int(np.argmax(ov))          # first max == strict '>' scan
        r1 = rs.refiner.Refine_trans(k0, k1, Trans[best], scores, inlinerdist=0.2)
        r2 = rs.refiner.Refine_trans(k0, k1, r1, scores, inlinerdist=0.1)
        out.update({f'{tag}_k0': k0, f'{tag}_k1': k1, f'{tag}_scores': scores, f'{tag}_Trans': Trans, f'{tag}_overlap': ov,
                    f'{tag}_masks': masks, f'{tag}_best': np.int64(best), f'{tag}_refine1': r1, f'{tag}_refine2': r2})
    # single-inlier refine edge (R = I expected)
    k1 = rng.uniform(0, 3, (5, 3)); k0 = rng.uniform(5, 8, (5, 3)); k0[2] = k1[2]
    Tn = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    r = ref_est.refiner().Refine_trans(k0, k1, Tn, np.ones(5), inlinerdist=0.1)
    out.update({'single_k0': k0, 'single_k1': k1, 'single_T': Tn, 'single_refined': r})
    # two-inlier edge: H has rank 1, U V^T is whatever LAPACK's null-space basis gives
    k1 = rng.uniform(0, 3, (6, 3)); k0 = rng.uniform(5, 8, (6, 3)); k0[1] = k1[1] + 0.01; k0[4] = k1[4] - 0.02
    r = ref_est.refiner().Refine_trans(k0, k1, Tn, np.ones(6), inlinerdist=0.1)
    out.update({'rank1_k0': k0, 'rank1_k1': k1, 'rank1_T': Tn, 'rank1_refined': r})
    save('ransac', **out)


def make_cfg_like(cfg, **kw):
    d = dict(vars(cfg)); d.update(kw); return NS(**d)


def gen_quat():
    rng = np.random.default_rng(14)
    q = rng.standard_normal((50, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    Rs = np.stack([ref_reval.matrix_from_quaternion(q[i]) for i in range(50)])
    T = tables()
    A = np.stack([T.R[int(rng.integers(0, 60))] for _ in range(50)])
    Rn = Rs @ A + 1e-3 * rng.standard_normal((50, 3, 3))
    diffs = np.array([ref_reval.compute_R_diff(A[i], Rn[i]) for i in range(50)])
    qm = np.stack([ref_reval.quaternion_from_matrix(Rn[i]) for i in range(50)])
    save('quat', q=q, R=Rs, A=A, Rn=Rn, rdiff=diffs, qfrommat=qm)


# ====================================================================================================
def gen_config1():
    """SURVEY 8(d) config 1 (N = 256, g = 7, t = (0.3, -0.2, 0.5)) through the reference's mutual matcher + yohoo estimator; the inputs are rebuilt
    from the recipe (roreg_amd.synth.config1_pair), only their checksum and the reference's outputs are stored."""
    import hashlib
    ds = synth.config1_pair()
    h = hashlib.sha256()
    for a in ds.feats + ds._kps:
        h.update(np.ascontiguousarray(a).tobytes())
    run_pipeline('config1', RD=False, RM=False, ET='yohoo', keynum=256, n_kpts=256, n_clouds=2, keep_yoho='sample', ds=ds,
                 extra={'inputs_sha256': np.frombuffer(h.hexdigest().encode(), np.uint8)})


def run_pipeline(tag, RD, RM, ET, keynum, n_kpts=128, n_clouds=3, seed=5, match_n=0.5, keep_yoho=True, ds=None, extra=None):
    """End-to-end reference run on a small synthetic scene; captures every inter-stage file."""
    root = tempfile.mkdtemp(prefix='golden_')
    try:
        cfg = make_cfg(root, RD=RD, RM=RM, ET=ET, keynum=keynum, match_n=match_n)
        ds = synth.make_scene(seed, n_clouds=n_clouds, n_kpts=n_kpts, overlap=0.6, name='synth/scene0') if ds is None else ds
        ds.write_inputs(cfg.output_cache_fn)
        out = {'n_kpts': np.int64(n_kpts), 'n_clouds': np.int64(n_clouds), 'scene_seed': np.int64(seed),
               'keynum': np.int64(keynum), 'RD': np.bool_(RD), 'RM': np.bool_(RM), 'match_n': np.float64(match_n)}
        name2extractor['yoho_des'](cfg).run(ds)
        if RD:
            name2detector['yoho_det'](cfg).run(ds)
        np.random.seed(1234)
        name2matcher['yoho_mat' if RM else 'matmul'](cfg).run(ds, keynum)
        np.random.seed(4321)
        name2estimator[ET](cfg).run(ds, keynum, 1000) if ET == 'yohoo' else run_yohoc_inline(cfg, ds, keynum)
        base = f'{cfg.output_cache_fn}/{ds.name}'
        for pc in ds.pc_ids:
            y = np.load(f'{base}/YOHO_Output_Group_feature/{pc}.npy')
            if keep_yoho == 'sample':
                out[f'yoho_sample_{pc}'] = y[::8]
            elif keep_yoho:
                out[f'yoho_{pc}'] = y
            else:   # same scene + same GF weights as the first pipeline: stored once there
                first = np.load(os.path.join(OUT, 'pipeline_mutual_yohoo.npz'))
                assert np.array_equal(first[f'yoho_{pc}'], y)
            if RD:
                out[f'det_{pc}'] = np.load(f'{base}/det_score/{pc}.npy')
        md = f'{base}/match_{keynum}'
        for a, b in ds.pair_ids:
            out[f'match_{a}_{b}'] = np.load(f'{md}/{a}-{b}.npy')
            out[f'mscore_{a}_{b}'] = np.load(f'{md}/scores/{a}-{b}.npy')
            out[f'dr_{a}_{b}'] = np.load(f'{md}/DR_index/{a}-{b}.npy')
            if ET == 'yohoo':
                out[f'transpre_{a}_{b}'] = np.load(f'{md}/Trans_pre/{a}-{b}.npy')
            r = np.load(f'{md}/{ET}/1000iters/{a}-{b}.npz', allow_pickle=True)
            out[f'trans_{a}_{b}'] = r['trans']; out[f'recall_{a}_{b}'] = np.int64(r['recalltime'])
        out['pre_log'] = np.frombuffer(open(f'{md}/{ET}/1000iters/pre.log', 'rb').read(), np.uint8)
        # evaluator metrics on the same files
        ev = yoho_evaluator.__new__(yoho_evaluator)
        ev.cfg = cfg; ev.keynum = keynum; ev.max_iter = 1000; ev.ET = ET
        fmr, ir = ev.fmr_ir_scene(ds)
        rr, rre, rte = ev.rr_scene(ds)
        out.update(fmr=np.float64(fmr), ir=np.float64(ir), rr=np.float64(rr), rre=np.float64(rre), rte=np.float64(rte))
        out.update(extra or {})
        save(f'pipeline_{tag}', **out)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def run_yohoc_inline(cfg, ds, keynum):
    """yohoc without the multiprocessing.Pool (estimator.py:258-262) so the global RNG stream is the
    parent's: Rindex, then ransac_once per pair in order, then pre.log."""
    est = name2estimator['yohoc'](cfg)
    est.rind_extractor.Rindex(ds, keynum)
    sd = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}/yohoc/1000iters'
    os.makedirs(sd, exist_ok=True)
    for pair in ds.pair_ids:
        est.ransacer.ransac_once(ds, keynum, 1000, pair)
    ref_est.R_pre_log(ds, sd)


def gen_match_ot(cfg):
    net = name2network['RM_test'](cfg)
    ck = torch.load(f'{REF}/checkpoints/FCGF/RM/model_best.pth')
    net.load_state_dict(ck['network_state_dict'], strict=True); net.eval()
    ds = synth.make_scene(21, n_clouds=2, n_kpts=120, overlap=0.6)
    # GF-like inputs: unit-normalised eqv features
    f0 = ds.feats[0][:112]; f1 = ds.feats[1][:120]
    f0 = f0 / np.sqrt((f0 * f0).sum(1, keepdims=True)); f1 = f1 / np.sqrt((f1 * f1).sum(1, keepdims=True))
    k0 = ds.get_kps('0')[:112].astype(np.float32); k1 = ds.get_kps('1')[:120].astype(np.float32)
    batch = {'feats0': torch.from_numpy(f1[None].copy()), 'feats1': torch.from_numpy(f0[None].copy()),
             'keys0': torch.from_numpy(k1[None].copy()), 'keys1': torch.from_numpy(k0[None].copy())}
    with torch.no_grad():
        r = net(batch)
    out = {'feats0': f1[None], 'feats1': f0[None], 'keys0': k1[None], 'keys1': k0[None]}
    for k, v in r.items():
        out['out_' + k] = v.numpy()
    save('match_ot', **out)


def gen_rr_cal():
    """utils/RR_cal.benchmark on a fabricated Redwood-format scene (gt.log, gt.info, pre.log).  nibabel is not installed, so
    nibabel.quaternions.mat2quat is served by the reference's own utils.r_eval.quaternion_from_matrix (same K-matrix method)."""
    import utils.RR_cal as ref_rr
    sys.modules['nibabel.quaternions'].mat2quat = lambda r: ref_reval.quaternion_from_matrix(r)
    ref_rr.nq = sys.modules['nibabel.quaternions']
    rng = np.random.default_rng(15)
    root = tempfile.mkdtemp(prefix='golden_rr_')
    try:
        n_frag = 7
        pairs = [(i, j) for i in range(n_frag) for j in range(i + 1, n_frag) if rng.random() < 0.7]
        scene_dir = f'{root}/origin/synth/scene0/PointCloud'
        os.makedirs(scene_dir)
        T = tables()
        gt_lines, info_lines, pre_lines = [], [], []
        for (i, j) in pairs:
            Rg = T.R[int(rng.integers(0, 60))]; tg = rng.uniform(-1, 1, 3)
            G = np.eye(4); G[:3, :3] = Rg; G[:3, 3] = tg
            gt_lines.append(f'{i}\t {j}\t {n_frag}\t\n' + ''.join('\t'.join(f'{v:.8f}' for v in G[r]) + '\n' for r in range(4)))
            A = rng.standard_normal((6, 6)); info = A @ A.T * 50 + np.eye(6) * 200
            info_lines.append(f'{i}\t{j}\t{n_frag}\n' + ''.join('\t'.join(f'{v:.6f}' for v in info[r]) + '\n' for r in range(6)))
            # estimate: GT perturbed (small for most pairs, large for some)
            big = rng.random() < 0.3
            ax = rng.standard_normal(3); ax /= np.linalg.norm(ax); ang = (0.5 if big else 0.01) * rng.standard_normal()
            Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
            E = np.eye(4); E[:3, :3] = dR @ Rg; E[:3, 3] = tg + (0.5 if big else 0.01) * rng.standard_normal(3)
            pre_lines.append(f'{i}\t{j}\t{n_frag}\n' + ''.join(f'{E[r][0]}\t{E[r][1]}\t{E[r][2]}\t{E[r][3]}\n' for r in range(3)) + '0.0\t0.0\t0.0\t1.0\n')
        gt_txt = ''.join(gt_lines); info_txt = ''.join(info_lines); pre_txt = ''.join(pre_lines)
        open(f'{scene_dir}/gt.log', 'w').write(gt_txt); open(f'{scene_dir}/gt.info', 'w').write(info_txt)
        cfg = NS(output_cache_fn=f'{root}/cache', tau_3=0.2)
        pre_dir = f'{cfg.output_cache_fn}/synth/scene0/match_128/yohoo/1000iters'
        os.makedirs(pre_dir); open(f'{pre_dir}/pre.log', 'w').write(pre_txt)
        ds = NS(name='synth/scene0', gt_dir=f'{scene_dir}/gt.log')
        datasets = {'wholesetname': 'synth', 'scene0': ds}
        rr, flags, errors = ref_rr.benchmark(cfg, datasets, 128, 1000, yoho_sign='yohoo')
        result_txt = open(f'{cfg.output_cache_fn}/synth/Eval_results/yohoo_RR/1000iters/result.txt').read()
        save('rr_cal', gt_log=np.frombuffer(gt_txt.encode(), np.uint8), gt_info=np.frombuffer(info_txt.encode(), np.uint8),
             pre_log=np.frombuffer(pre_txt.encode(), np.uint8), rr=np.float64(rr), flags=np.array(flags['synth/scene0']),
             errors=np.array(errors['synth/scene0']), result_txt=np.frombuffer(result_txt.encode(), np.uint8))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    if len(sys.argv) > 1:                                          # python tools/gen_golden.py knn_api ... : only the named fixtures
        for name in sys.argv[1:]:
            print(name); globals()[f'gen_{name}']()
        return
    root = tempfile.mkdtemp(prefix='golden_cfg_')
    try:
        cfg = make_cfg(root)
        print('weights'); gen_weights()
        print('gf'); gen_gf(cfg)
        print('rd'); gen_rd(cfg)
        print('knn'); gen_knn()
        print('knn_api'); gen_knn_api()
        print('nms'); gen_nms()
        print('des2r'); gen_des2r(cfg)
        print('et'); gen_et(cfg)
        print('ransac'); gen_ransac(cfg)
        print('quat'); gen_quat()
        print('match_ot'); gen_match_ot(cfg)
        print('rr_cal'); gen_rr_cal()
    finally:
        shutil.rmtree(root, ignore_errors=True)
    print('pipelines')
    run_pipeline('mutual_yohoo', RD=False, RM=False, ET='yohoo', keynum=128)
    run_pipeline('rd_mutual_yohoc', RD=True, RM=False, ET='yohoc', keynum=96, keep_yoho=False)
    run_pipeline('rd_rm_yohoo', RD=True, RM=True, ET='yohoo', keynum=96, keep_yoho=False)


if __name__ == '__main__':
    main()
