"""Full-size (BASELINE-size) parity fixtures: tests/golden/full_*.npz.

Same rules as tools/gen_golden.py (build container only; the reference is imported from /root/reference and run on CPU; only
vectors are written).  The INPUTS of every case are rebuilt from a seed by roreg_amd/synth.py (portable arithmetic only), so the
fixtures hold just the reference's small outputs: index lists, packed inlier masks, transforms and strided samples of the big tensors.

    python tools/gen_golden_full.py [stages] [ransac] [ransac_ties] [match_ot] [pipeline] [pipeline_rd_rm] [pipeline_rd_rm_o60] [pipeline_rd_rm_o60_s1..3] [pipeline_rd_rm_k5000] [match_ot_5000] [match_ot_3000x1000] [match_ot_1200x4000] [yohoc] [rd]        (no argument = all; ~10 minutes on 8 cores)
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg                       # noqa: E402  (shims + reference imports; changes cwd to /root/reference)
from gen_golden import synth, tables, name2network, name2extractor, name2matcher, name2estimator, ref_est, ref_utils, save, REF   # noqa: E402

STAGE_SEED, RANSAC_SEED, OT_SEED, PIPE_SEED = 31, 41, 51, 61


def _i16(a):
    a = np.asarray(a)
    assert a.min() >= -1 and a.max() < 32768
    return a.astype(np.int16)


def gen_stages():
    """matcher -> Des2R -> ET/Trans_pre -> one-shot RANSAC of the reference on a 5000-keypoint near-tie pair, stage outputs stored."""
    root = tempfile.mkdtemp(prefix='golden_full_')
    try:
        cfg = gg.make_cfg(root, keynum=5000)
        cfg.bs_ET = 500
        ds = synth.make_neartie_scene(STAGE_SEED, n_clouds=2, n_kpts=5000)
        ds.write_inputs(cfg.output_cache_fn)
        base = f'{cfg.output_cache_fn}/{ds.name}'
        os.makedirs(f'{base}/YOHO_Output_Group_feature')
        for pc, f in zip(ds.pc_ids, ds.feats):                       # the extractor stage is skipped: eqv := the scene's group features
            np.save(f'{base}/YOHO_Output_Group_feature/{pc}.npy', f)
        np.random.seed(1234)
        name2matcher['matmul'](cfg).run(ds, 5000)
        np.random.seed(4321)
        name2estimator['yohoo'](cfg).run(ds, 5000, 1000)
        md = f'{base}/match_5000'
        m = np.load(f'{md}/0-1.npy'); dr = np.load(f'{md}/DR_index/0-1.npy'); tp = np.load(f'{md}/Trans_pre/0-1.npy')
        r = np.load(f'{md}/yohoo/1000iters/0-1.npz', allow_pickle=True)
        print('   matches', m.shape, 'recalltime', int(r['recalltime']))
        save('full_stages', scene_seed=np.int64(STAGE_SEED), match=_i16(m), dr=dr.astype(np.int8), transpre=tp,
             trans=r['trans'], recalltime=np.int64(r['recalltime']))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def gen_ransac():
    out = {}
    for tag, f32s in (('ones', False), ('f32', True)):
        k0, k1, scores, Trans, hyp = synth.make_ransac_case(RANSAC_SEED + int(f32s), M=5000, H=1000, f32_scores=f32s)
        rs = ref_est.yohoo_ransac(gg.make_cfg_like(gg.NS(ransac_ird=0.1, RM=f32s, match_n=0.5, output_cache_fn='/tmp', SO3_related_files=f'{REF}/utils/group_related')))
        ov = np.array([rs.overlap_cal(k0, k1, Trans[i], scores) for i in hyp])
        masks = np.stack([np.sum(np.square(k0 - ref_utils.transform_points(k1, Trans[i])), -1) < 0.1 * 0.1 for i in hyp])
        best = int(np.argmax(ov))
        r1 = rs.refiner.Refine_trans(k0, k1, Trans[hyp[best]], scores, inlinerdist=0.2)
        r2 = rs.refiner.Refine_trans(k0, k1, r1, scores, inlinerdist=0.1)
        print('   ', tag, 'best', best, 'overlap', ov[best], 'inliers', int(masks[best].sum()))
        out.update({f'{tag}_seed': np.int64(RANSAC_SEED + int(f32s)), f'{tag}_overlap': ov, f'{tag}_masks': np.packbits(masks, axis=1),
                    f'{tag}_best': np.int64(best), f'{tag}_refine1': r1, f'{tag}_refine2': r2})
    save('full_ransac', **out)


def gen_ransac_ties():
    """One-shot RANSAC of the reference on float32 scores that tie at float32 precision (synth.make_ransac_tie_case): the seed is searched
    until the reference's winner (float32 pairwise sums, float32 quotient, strict `>`) differs from the winner of a float64 accumulation of
    the same weights, so the fixture tells the two evaluation orders apart."""
    rs = ref_est.yohoo_ransac(gg.make_cfg_like(gg.NS(ransac_ird=0.1, RM=True, match_n=0.5, output_cache_fn='/tmp', SO3_related_files=f'{REF}/utils/group_related')))
    for seed in range(700, 760):
        k0, k1, scores, Trans, hyp = synth.make_ransac_tie_case(seed)
        ov = np.array([rs.overlap_cal(k0, k1, Trans[i], scores) for i in hyp])
        assert ov.dtype == np.float32
        masks = np.stack([np.sum(np.square(k0 - ref_utils.transform_points(k1, Trans[i])), -1) < 0.1 * 0.1 for i in hyp])
        ov64 = np.array([scores[m].astype(np.float64).sum() / scores.shape[0] for m in masks])
        best, best64 = int(np.argmax(ov)), int(np.argmax(ov64))
        ties = int((ov == ov[best]).sum())
        print('   seed', seed, 'best', best, 'float64-accumulation best', best64, 'hypotheses at the float32 maximum', ties)
        if best != best64 and ties >= 2:
            break
    else:
        raise SystemExit('no discriminating seed found')
    r1 = rs.refiner.Refine_trans(k0, k1, Trans[hyp[best]], scores, inlinerdist=0.2)
    r2 = rs.refiner.Refine_trans(k0, k1, r1, scores, inlinerdist=0.1)
    save('full_ransac_ties', seed=np.int64(seed), overlap=ov, masks=np.packbits(masks, axis=1), best=np.int64(best),
         best_of_float64_accumulation=np.int64(best64), refine1=r1, refine2=r2)


def gen_match_ot(n=2500, tag='full_match_ot', seed=OT_SEED, m_src=None, n_tgt=None):
    """`Match_ot.forward` of the reference with the shipped RM weights (network/rot_coh_match.py:339-390).  n = 2500 is yoho_mat's default
    keynum; n = 5000 (`full_match_ot_5000`) is what `Test.py --RM --keynum 5000` hands it (test/evaluator.py:20,46 -> test/matcher.py:152-185)."""
    cfg = gg.make_cfg(tempfile.mkdtemp(prefix='golden_full_cfg_'))
    net = name2network['RM_test'](cfg)
    ck = torch.load(f'{REF}/checkpoints/FCGF/RM/model_best.pth')
    net.load_state_dict(ck['network_state_dict'], strict=True); net.eval()
    ds = synth.make_scene(seed, n_clouds=2, n_kpts=n, overlap=0.6, coord_noise=0.005, portable=True)
    f0 = ds.feats[0]; f1 = ds.feats[1]
    f0 = f0 / np.sqrt((f0 * f0).sum(1, keepdims=True)); f1 = f1 / np.sqrt((f1 * f1).sum(1, keepdims=True))
    k0 = ds.get_kps('0').astype(np.float32); k1 = ds.get_kps('1').astype(np.float32)
    if m_src is not None:             # a ragged pair: the first m_src points of the source side against the first n_tgt of the target side
        f1, k1, f0, k0 = f1[:m_src], k1[:m_src], f0[:n_tgt], k0[:n_tgt]
    batch = {'feats0': torch.from_numpy(f1[None].copy()), 'feats1': torch.from_numpy(f0[None].copy()),
             'keys0': torch.from_numpy(k1[None].copy()), 'keys1': torch.from_numpy(k0[None].copy())}
    with torch.no_grad():
        r = net(batch)
    m0 = r['matches0'][0].numpy(); m1 = r['matches1'][0].numpy()
    print('   valid matches', int((m0 >= 0).sum()))
    Z = r['scores'][0].numpy()
    save(tag, scene_seed=np.int64(seed), n=np.int64(n), m_src=np.int64(f1.shape[0]), n_tgt=np.int64(f0.shape[0]), matches0=_i16(m0), matches1=_i16(m1),
         matching_scores0=r['matching_scores0'][0].numpy(), matching_scores1=r['matching_scores1'][0].numpy(),
         scores_sample=Z[::40, ::40].copy(), scores_lastrow=Z[-1, ::10].copy(), scores_lastcol=Z[::10, -1].copy(),
         source_final_sample=r['source_final'][0, :, ::25, 0].numpy(), target_final_sample=r['target_final'][0, :, ::25, 0].numpy())
    shutil.rmtree(cfg.base_dir, ignore_errors=True)


def gen_pipeline():
    """The reference end to end (GF -> mutual -> yohoo; seeded GF/ET weights) on three 5000-keypoint clouds."""
    root = tempfile.mkdtemp(prefix='golden_full_')
    try:
        cfg = gg.make_cfg(root, keynum=5000)
        cfg.bs_GF = 250; cfg.bs_ET = 500
        ds = synth.make_scene(PIPE_SEED, n_clouds=3, n_kpts=5000, overlap=0.6, coord_noise=0.005, name='synth/scene0', portable=True)
        ds.write_inputs(cfg.output_cache_fn)
        name2extractor['yoho_des'](cfg).run(ds)
        np.random.seed(1234)
        name2matcher['matmul'](cfg).run(ds, 5000)
        np.random.seed(4321)
        name2estimator['yohoo'](cfg).run(ds, 5000, 1000)
        base = f'{cfg.output_cache_fn}/{ds.name}'
        out = {'scene_seed': np.int64(PIPE_SEED)}
        for pc in ds.pc_ids:
            y = np.load(f'{base}/YOHO_Output_Group_feature/{pc}.npy')
            out[f'yoho_sample_{pc}'] = y[::250].copy()
            inv = y.mean(-1); inv = inv / (np.sqrt((inv * inv).sum(1, keepdims=True)) + 1e-5)
            out[f'yoho_absmax_{pc}'] = np.float32(np.abs(y).max())
        md = f'{base}/match_5000'
        for a, b in ds.pair_ids:
            out[f'match_{a}_{b}'] = _i16(np.load(f'{md}/{a}-{b}.npy'))
            out[f'dr_{a}_{b}'] = np.load(f'{md}/DR_index/{a}-{b}.npy').astype(np.int8)
            out[f'transpre_sample_{a}_{b}'] = np.load(f'{md}/Trans_pre/{a}-{b}.npy')[::16].copy()
            r = np.load(f'{md}/yohoo/1000iters/{a}-{b}.npz', allow_pickle=True)
            out[f'trans_{a}_{b}'] = r['trans']; out[f'recall_{a}_{b}'] = np.int64(r['recalltime'])
            print('   pair', a, b, 'matches', out[f'match_{a}_{b}'].shape[0], 'recalltime', int(r['recalltime']))
        save('full_pipeline', **out)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def gen_pipeline_rd_rm(tag='full_pipeline_rd_rm', overlap=0.2, seed=PIPE_SEED + 7, keynum=2500):
    """BASELINE config 4's chain in the reference, end to end at full size on a LOW-OVERLAP pair (20 % shared keypoints; a second fixture
    `_o60` at 60 %, where the trained matcher finds enough correct correspondences on synthetic descriptors to register the pair): GF (seeded
    weights) -> detector (shipped RD weights) -> rank scores -> NMS sampling of 2500 keypoints -> yoho_mat (shipped RM weights) -> one-shot
    RANSAC on the top-`match_n` = 0.5 matches (test/detector.py:26-47, test/matcher.py:11-42,152-210, test/estimator.py:405-443).
    Stored: the detector's rank order, both NMS samples, matches + scores, DR_index, a strided sample of Trans_pre, the result.
    `_o60_s1..3`: three more 60 % pairs (other seeds) for the comparison of the two GEMM kernels; `_k5000`: the same chain at
    `--keynum 5000`, SURVEY 3.1's hot path (NMS_sample at num == n still runs its k-NN branch and returns a permutation; Match_ot sees
    m = n = 5000)."""
    root = tempfile.mkdtemp(prefix='golden_full_')
    try:
        cfg = gg.make_cfg(root, RD=True, RM=True, keynum=keynum, match_n=0.5)
        cfg.bs_GF = 250; cfg.bs_ET = 500
        ds = synth.make_scene(seed, n_clouds=2, n_kpts=5000, overlap=overlap, coord_noise=0.005, name='synth/scene0', portable=True)
        ds.write_inputs(cfg.output_cache_fn)
        name2extractor['yoho_des'](cfg).run(ds)
        gg.name2detector['yoho_det'](cfg).run(ds)
        np.random.seed(1234)
        mat = name2matcher['yoho_mat'](cfg)
        mat.run(ds, keynum)
        np.random.seed(4321)
        name2estimator['yohoo'](cfg).run(ds, keynum, 1000)
        base = f'{cfg.output_cache_fn}/{ds.name}'
        out = {'scene_seed': np.int64(seed), 'overlap': np.float64(overlap), 'keynum': np.int64(keynum)}
        RefNMS = gg.ref_mat.NMS_sample                                # the reference's sampler, to store the two samples it drew
        for pc in ds.pc_ids:
            det = np.load(f'{base}/det_score/{pc}.npy')
            out[f'det_rank_{pc}'] = _i16(np.rint(det * det.shape[0]))
            out[f'nms_{pc}'] = _i16(RefNMS(keynum, 5).sample(ds.get_kps(pc), det))
            y = np.load(f'{base}/YOHO_Output_Group_feature/{pc}.npy')
            out[f'yoho_sample_{pc}'] = y[::250].copy()
        md = f'{base}/match_{keynum}'
        for a, b in ds.pair_ids:
            out[f'match_{a}_{b}'] = _i16(np.load(f'{md}/{a}-{b}.npy'))
            out[f'mscore_{a}_{b}'] = np.load(f'{md}/scores/{a}-{b}.npy')
            out[f'dr_{a}_{b}'] = np.load(f'{md}/DR_index/{a}-{b}.npy').astype(np.int8)
            out[f'transpre_sample_{a}_{b}'] = np.load(f'{md}/Trans_pre/{a}-{b}.npy')[::16].copy()
            r = np.load(f'{md}/yohoo/1000iters/{a}-{b}.npz', allow_pickle=True)
            out[f'trans_{a}_{b}'] = r['trans']; out[f'recall_{a}_{b}'] = np.int64(r['recalltime'])
            gt = ds.get_transform(a, b)
            print('   pair', a, b, 'matches', out[f'match_{a}_{b}'].shape[0], 'score dtype', out[f'mscore_{a}_{b}'].dtype, 'recalltime', int(r['recalltime']),
                  'RRE', gg.ref_reval.compute_R_diff(gt[:3, :3], r['trans'][:3, :3]), 'RTE', float(np.linalg.norm(gt[:3, 3] - r['trans'][:3, 3])))
        save(tag, **out)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def gen_yohoc():
    """The rotation-bin estimator (test/estimator.py:173-241, run without its Pool like tools/gen_golden.py does) on the matches and
    DR_index of the full-size near-tie pair (inputs = arrays of full_stages.npz + the seed-rebuilt keypoints)."""
    z = np.load(os.path.join(gg.OUT, 'full_stages.npz'))
    root = tempfile.mkdtemp(prefix='golden_full_')
    try:
        cfg = gg.make_cfg(root, keynum=5000, ET='yohoc')
        ds = synth.make_neartie_scene(int(z['scene_seed']), n_clouds=2, n_kpts=5000)
        md = f'{cfg.output_cache_fn}/{ds.name}/match_5000'
        os.makedirs(f'{md}/scores'); os.makedirs(f'{md}/DR_index'); os.makedirs(f'{md}/yohoc/1000iters')
        m = z['match'].astype(np.int64)
        np.save(f'{md}/0-1.npy', m); np.save(f'{md}/scores/0-1.npy', np.ones(m.shape[0])); np.save(f'{md}/DR_index/0-1.npy', z['dr'].astype(np.int64))
        est = name2estimator['yohoc'](cfg)
        np.random.seed(4321)
        est.ransacer.ransac_once(ds, 5000, 1000, ('0', '1'))
        r = np.load(f'{md}/yohoc/1000iters/0-1.npz', allow_pickle=True)
        print('   yohoc recalltime', int(r['recalltime']))
        save('full_yohoc', trans=r['trans'], recalltime=np.int64(r['recalltime']))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def gen_rd():
    """The detector with the shipped RD weights on a whole 5000-keypoint cloud (network/rot_detect.py:43-55, test/detector.py:45-46)."""
    cfg = gg.make_cfg(tempfile.mkdtemp(prefix='golden_full_cfg_'))
    net = name2network['RD_test'](cfg)
    ck = torch.load(f'{REF}/checkpoints/FCGF/RD/model_best.pth')
    net.load_state_dict(ck['network_state_dict'], strict=True); net.eval()
    ds = synth.make_scene(PIPE_SEED + 1, n_clouds=1, n_kpts=5000, overlap=0.6, portable=True)
    x = ds.feats[0] / np.sqrt((ds.feats[0] * ds.feats[0]).sum(1, keepdims=True))
    with torch.no_grad():
        raw = net({'feats': torch.from_numpy(x.copy())})['scores'].numpy()
    rank = raw.copy(); rank[np.argsort(raw)] = np.arange(raw.shape[0]) / raw.shape[0]
    save('full_rd', scene_seed=np.int64(PIPE_SEED + 1), raw=raw, rank=rank.astype(np.float32))
    shutil.rmtree(cfg.base_dir, ignore_errors=True)


if __name__ == '__main__':
    todo = sys.argv[1:] or ['ransac', 'ransac_ties', 'stages', 'match_ot', 'pipeline', 'pipeline_rd_rm', 'pipeline_rd_rm_o60', 'pipeline_rd_rm_o60_s1', 'pipeline_rd_rm_o60_s2', 'pipeline_rd_rm_o60_s3', 'pipeline_rd_rm_k5000', 'match_ot_5000', 'match_ot_3000x1000', 'match_ot_1200x4000', 'yohoc', 'rd']
    for name in todo:
        print(name)
        {'stages': gen_stages, 'ransac': gen_ransac, 'ransac_ties': gen_ransac_ties, 'match_ot': gen_match_ot, 'pipeline': gen_pipeline, 'pipeline_rd_rm': gen_pipeline_rd_rm, 'pipeline_rd_rm_o60': lambda: gen_pipeline_rd_rm('full_pipeline_rd_rm_o60', 0.6, PIPE_SEED + 8),
         'pipeline_rd_rm_o60_s1': lambda: gen_pipeline_rd_rm('full_pipeline_rd_rm_o60_s1', 0.6, PIPE_SEED + 21),
         'pipeline_rd_rm_o60_s2': lambda: gen_pipeline_rd_rm('full_pipeline_rd_rm_o60_s2', 0.6, PIPE_SEED + 22),
         'pipeline_rd_rm_o60_s3': lambda: gen_pipeline_rd_rm('full_pipeline_rd_rm_o60_s3', 0.6, PIPE_SEED + 23),
         'pipeline_rd_rm_k5000': lambda: gen_pipeline_rd_rm('full_pipeline_rd_rm_k5000', 0.6, PIPE_SEED + 9, keynum=5000),
         'match_ot_5000': lambda: gen_match_ot(5000, 'full_match_ot_5000', OT_SEED + 1),
         'match_ot_3000x1000': lambda: gen_match_ot(3000, 'full_match_ot_3000x1000', OT_SEED + 2, 3000, 1000),
         'match_ot_1200x4000': lambda: gen_match_ot(4000, 'full_match_ot_1200x4000', OT_SEED + 3, 1200, 4000), 'yohoc': gen_yohoc, 'rd': gen_rd}[name]()
