"""File-coupled stage classes (the reference's API) and the device-resident engine on the golden scene.  GPU only.
Stage-wise parity: each stage is fed the REFERENCE's output of the previous stage."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from roreg_amd import synth
from roreg_amd.parses.parses_test import default_config

pytestmark = pytest.mark.gpu


def _setup(tmp_path, z, **cfg_kw):
    from roreg_amd.network import name2network
    root = str(tmp_path)
    cfg = default_config(output_cache_fn=f'{root}/cache', model_fn=f'{root}/ckpt', base_dir=root, SO3_related_files=None,
                         keynum=int(z['keynum']), bs_GF=50, bs_ET=40, **cfg_kw)
    for kind, d, seed in [('GF_test', 'GF', 101), ('ET_test', 'ET', 202)]:
        net = name2network[kind](cfg)
        synth.seeded_state_dict(net, seed)
        os.makedirs(f'{root}/ckpt/{d}', exist_ok=True)
        torch.save({'best_para': 0, 'network_state_dict': net.state_dict()}, f'{root}/ckpt/{d}/model_best.pth')
    for d in ['RD', 'RM']:
        os.makedirs(f'{root}/ckpt/{d}', exist_ok=True)
        sd = {k: torch.from_numpy(v) for k, v in load_golden(f'weights_{d}').items()}
        torch.save({'best_para': 0, 'network_state_dict': sd}, f'{root}/ckpt/{d}/model_best.pth')
    ds = synth.make_scene(int(z['scene_seed']), n_clouds=int(z['n_clouds']), n_kpts=int(z['n_kpts']), overlap=0.6, name='synth/scene0')
    ds.write_inputs(cfg.output_cache_fn)
    return cfg, ds


def _put_yoho(cfg, ds, y):
    d = f'{cfg.output_cache_fn}/{ds.name}/YOHO_Output_Group_feature'
    os.makedirs(d, exist_ok=True)
    for pc in ds.pc_ids:
        np.save(f'{d}/{pc}.npy', y[f'yoho_{pc}'])


def test_stages_mutual_yohoo_stagewise(tmp_path):
    from roreg_amd.test import name2extractor, name2matcher, name2estimator
    from roreg_amd.test import _cache
    z = load_golden('pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, ET='yohoo')
    keynum = int(z['keynum'])
    base = f'{cfg.output_cache_fn}/{ds.name}'
    # stage 1
    name2extractor['yoho_des'](cfg).run(ds)
    for pc in ds.pc_ids:
        got = np.load(f'{base}/YOHO_Output_Group_feature/{pc}.npy')
        assert got.dtype == np.float32 and np.abs(got - z[f'yoho_{pc}']).max() < 1e-5
    _put_yoho(cfg, ds, z); _cache.clear()
    # stage 3
    np.random.seed(1234)
    name2matcher['matmul'](cfg).run(ds, keynum)
    md = f'{base}/match_{keynum}'
    for a, b in ds.pair_ids:
        m = np.load(f'{md}/{a}-{b}.npy'); s = np.load(f'{md}/scores/{a}-{b}.npy')
        assert m.dtype == np.int64 and np.array_equal(m, z[f'match_{a}_{b}'])
        assert s.dtype == np.float64 and np.array_equal(s, z[f'mscore_{a}_{b}'])
    # stage 4
    est = name2estimator['yohoo'](cfg)
    est.rind_extractor.Rindex(ds, keynum)
    for a, b in ds.pair_ids:
        dr = np.load(f'{md}/DR_index/{a}-{b}.npy')
        assert dr.dtype == np.int64 and np.array_equal(dr, z[f'dr_{a}_{b}'])
    est.localT_extractor.Rt_pre(ds, keynum)
    for a, b in ds.pair_ids:
        T = np.load(f'{md}/Trans_pre/{a}-{b}.npy')
        assert T.dtype == np.float64 and T.shape == z[f'transpre_{a}_{b}'].shape
        assert np.abs(T - z[f'transpre_{a}_{b}']).max() < 2e-4
        np.save(f'{md}/Trans_pre/{a}-{b}.npy', z[f'transpre_{a}_{b}'])      # identical inputs for the RANSAC stage
    np.random.seed(4321)
    est.ransacer.ransac(ds, keynum, 1000)
    for a, b in ds.pair_ids:
        r = np.load(f'{md}/yohoo/1000iters/{a}-{b}.npz')
        assert int(r['recalltime']) == int(z[f'recall_{a}_{b}'])
        assert np.abs(r['trans'] - z[f'trans_{a}_{b}']).max() < 1e-8
    got = np.array(open(f'{md}/yohoo/1000iters/pre.log').read().split(), float)
    want = np.array(z['pre_log'].tobytes().decode().split(), float)
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-8


def test_config1_recipe_end_to_end_on_the_device(tmp_path):
    """BASELINE configs[0] / SURVEY 8(d) config 1 with its exact recipe (N = 256, group element 7, t = (0.3, -0.2, 0.5)) through the stage
    classes END TO END (no stage is fed the reference's intermediates) and through the device-resident engine: the reference's 256 matches,
    Des2R indices, recalltime and transform (1e-7, = the ground truth), FMR = IR = RR = 1."""
    from roreg_amd.network import name2network
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.test import name2extractor, name2matcher, name2estimator, _cache
    z = load_golden('pipeline_config1')
    root = str(tmp_path)
    cfg = default_config(output_cache_fn=f'{root}/cache', model_fn=f'{root}/ckpt', base_dir=root, SO3_related_files=None, keynum=256, ET='yohoo')
    nets = {}
    for kind, d, seed in [('GF_test', 'GF', 101), ('ET_test', 'ET', 202)]:
        net = name2network[kind](cfg)
        synth.seeded_state_dict(net, seed)
        os.makedirs(f'{root}/ckpt/{d}', exist_ok=True)
        torch.save({'best_para': 0, 'network_state_dict': net.state_dict()}, f'{root}/ckpt/{d}/model_best.pth')
        nets[d] = net
    ds = synth.config1_pair()
    ds.write_inputs(cfg.output_cache_fn)
    _cache.clear()
    name2extractor['yoho_des'](cfg).run(ds)
    base = f'{cfg.output_cache_fn}/{ds.name}'
    for pc in ds.pc_ids:
        assert np.abs(np.load(f'{base}/YOHO_Output_Group_feature/{pc}.npy')[::8] - z[f'yoho_sample_{pc}']).max() < 1e-5
    np.random.seed(1234)
    name2matcher['matmul'](cfg).run(ds, 256)
    md = f'{base}/match_256'
    m = np.load(f'{md}/0-1.npy')
    assert np.array_equal(m, z['match_0_1']) and np.array_equal(ds.perm[m[:, 1]], m[:, 0])
    np.random.seed(4321)
    name2estimator['yohoo'](cfg).run(ds, 256, 1000)
    assert np.array_equal(np.load(f'{md}/DR_index/0-1.npy'), z['dr_0_1'])
    tp = np.load(f'{md}/Trans_pre/0-1.npy')
    assert np.abs(tp[:, :, :3] - z['transpre_0_1'][:, :, :3]).max() < 1e-4 and np.abs(tp - z['transpre_0_1']).max() < 4e-4
    r = np.load(f'{md}/yohoo/1000iters/0-1.npz')
    gt = np.eye(4); gt[:3] = ds.get_transform('0', '1').astype(np.float64)
    assert int(r['recalltime']) == int(z['recall_0_1'])
    assert np.abs(r['trans'] - z['trans_0_1']).max() < 1e-7 and np.abs(r['trans'] - gt).max() < 1e-6
    # the device-resident engine on the same pair (the reference's global generator calls in the reference's order)
    eng = RegistrationEngine(cfg, nets['GF'], nets['ET'])
    np.random.seed(1234)
    res = eng.run_scene({0: ds.feats[0], 1: ds.feats[1]}, {0: ds.get_kps('0'), 1: ds.get_kps('1')}, ds.pair_ids, keynum=256, max_iter=1000, keep_matches=True)
    assert res[0].n_match == 256 and np.abs(res[0].trans - gt).max() < 1e-6


def test_stages_rd_mutual_yohoc_stagewise(tmp_path):
    from roreg_amd.test import name2detector, name2matcher, name2estimator
    z = load_golden('pipeline_rd_mutual_yohoc')
    y = load_golden('pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, RD=True, ET='yohoc')
    keynum = int(z['keynum'])
    base = f'{cfg.output_cache_fn}/{ds.name}'
    _put_yoho(cfg, ds, y)
    name2detector['yoho_det'](cfg).run(ds)
    n = int(z['n_kpts'])
    for pc in ds.pc_ids:
        got = np.load(f'{base}/det_score/{pc}.npy')
        assert got.dtype == np.float32
        assert np.abs(got - z[f'det_{pc}']).max() <= 4.0 / n          # rank noise of the ill-conditioned std (see oracle test)
        np.save(f'{base}/det_score/{pc}.npy', z[f'det_{pc}'])
    name2matcher['matmul'](cfg).run(ds, keynum)
    md = f'{base}/match_{keynum}'
    for a, b in ds.pair_ids:
        assert np.array_equal(np.load(f'{md}/{a}-{b}.npy'), z[f'match_{a}_{b}'])
    est = name2estimator['yohoc'](cfg)
    est.rind_extractor.Rindex(ds, keynum)
    for a, b in ds.pair_ids:
        assert np.array_equal(np.load(f'{md}/DR_index/{a}-{b}.npy'), z[f'dr_{a}_{b}'])
    np.random.seed(4321)
    est.ransacer.ransac(ds, keynum, 1000)
    for a, b in ds.pair_ids:
        r = np.load(f'{md}/yohoc/1000iters/{a}-{b}.npz')
        assert int(r['recalltime']) == int(z[f'recall_{a}_{b}'])
        assert np.abs(r['trans'] - z[f'trans_{a}_{b}']).max() < 1e-8


def test_evaluator_metrics_on_reference_files(tmp_path):
    """fmr / ir / rr computed by the evaluator mirror from the reference's own match + result files."""
    from roreg_amd.test.evaluator import yoho_evaluator
    z = load_golden('pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, ET='yohoo')
    keynum = int(z['keynum'])
    md = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}'
    os.makedirs(f'{md}/scores'); os.makedirs(f'{md}/yohoo/1000iters')
    for a, b in ds.pair_ids:
        np.save(f'{md}/{a}-{b}.npy', z[f'match_{a}_{b}']); np.save(f'{md}/scores/{a}-{b}.npy', z[f'mscore_{a}_{b}'])
        np.savez(f'{md}/yohoo/1000iters/{a}-{b}.npz', trans=z[f'trans_{a}_{b}'], recalltime=z[f'recall_{a}_{b}'])
    ev = yoho_evaluator(cfg)
    fmr, ir = ev.fmr_ir_scene(ds)
    rr, rre, rte = ev.rr_scene(ds)
    assert abs(fmr - float(z['fmr'])) < 1e-12 and abs(ir - float(z['ir'])) < 1e-12
    assert abs(rr - float(z['rr'])) < 1e-12 and abs(rre - float(z['rre'])) < 1e-9 and abs(rte - float(z['rte'])) < 1e-9


def test_engine_equals_file_coupled_stages(tmp_path):
    """The device-resident engine and the file-coupled stages run the same kernels in the same order on the same
    RNG stream: matches and the selected hypothesis are identical; the transform differs only by the closing 3x3
    SVD (device Jacobi vs host LAPACK), i.e. at 1e-15."""
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.test import name2extractor, name2matcher, name2estimator
    z = load_golden('pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, ET='yohoo')
    keynum = int(z['keynum'])
    np.random.seed(99)
    name2extractor['yoho_des'](cfg).run(ds)
    name2matcher['matmul'](cfg).run(ds, keynum)
    name2estimator['yohoo'](cfg).run(ds, keynum, 1000)
    md = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}'
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    eng = RegistrationEngine(cfg, gf, et)
    np.random.seed(99)
    res = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keynum=keynum, max_iter=1000, keep_matches=True)
    for r in res:
        want = np.load(f'{md}/yohoo/1000iters/{r.id0}-{r.id1}.npz')
        assert np.array_equal(r.matches.cpu().numpy(), np.load(f'{md}/{r.id0}-{r.id1}.npy'))
        assert r.recalltime == int(want['recalltime'])
        assert np.abs(r.trans - want['trans']).max() < 1e-10


@pytest.mark.parametrize('RD,RM,ET', [(False, False, 'yohoo'), (True, True, 'yohoo'), (False, False, 'yohoc')])
def test_engine_writer_files_equal_the_stage_classes_files(tmp_path, RD, RM, ET):
    """with a StageFileWriter the engine emits the reference's inter-stage files (matches, scores, DR_index, Trans_pre)
    asynchronously; they are byte for byte the files the file-coupled stage classes write from the same generator stream."""
    import filecmp
    from roreg_amd.engine import RegistrationEngine, StageFileWriter
    from roreg_amd.network import name2network
    from roreg_amd.test import name2extractor, name2detector, name2matcher, name2estimator
    z = load_golden('pipeline_rd_rm_yohoo' if RM else 'pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, ET=ET, RD=RD, RM=RM)
    keynum = int(z['keynum'])
    np.random.seed(99)
    name2extractor['yoho_des'](cfg).run(ds)
    if RD:
        name2detector['yoho_det'](cfg).run(ds)
    name2matcher['yoho_mat' if RM else 'matmul'](cfg).run(ds, keynum)
    name2estimator[ET](cfg).run(ds, keynum, 1000)
    ref_dir = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}'
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    rd = rm = None
    if RD:
        rd = name2network['RD_test'](cfg); rd.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RD').items()})
    if RM:
        rm = name2network['RM_test'](cfg); rm.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()})
    from types import SimpleNamespace as NS
    cfg2 = NS(**{**vars(cfg), 'output_cache_fn': f'{tmp_path}/cache_engine'})
    eng = RegistrationEngine(cfg2, gf, et if ET == 'yohoo' else None, rd_net=rd, rm_net=rm)
    w = StageFileWriter(cfg2, ds.name, keynum)
    np.random.seed(99)
    res = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keynum=keynum, max_iter=1000, writer=w)
    w.close()
    got_dir = f'{cfg2.output_cache_fn}/{ds.name}/match_{keynum}'
    kinds = ['', 'scores/', 'DR_index/'] + (['Trans_pre/'] if ET == 'yohoo' else [])
    for a, b in ds.pair_ids:
        for k in kinds:
            assert filecmp.cmp(f'{got_dir}/{k}{a}-{b}.npy', f'{ref_dir}/{k}{a}-{b}.npy', shallow=False), (k, a, b)
    for r in res:                                                       # and the registration itself is the stages'
        want = np.load(f'{ref_dir}/{ET}/1000iters/{r.id0}-{r.id1}.npz')
        assert r.recalltime == int(want['recalltime'])
        if np.isfinite(want['trans']).all():
            assert np.abs(r.trans - want['trans']).max() < 1e-10


@pytest.mark.parametrize('RD,RM,ET', [(False, False, 'yohoo'), (True, True, 'yohoo'), (False, False, 'yohoc'), (True, False, 'yohoo')])
def test_evaluator_on_the_engine_writes_the_stage_chains_files(tmp_path, monkeypatch, RD, RM, ET):
    """`yoho_evaluator.process_scene` (test/evaluator.py:39-48: what an unchanged Test.py calls) runs on the device-resident engine when its
    stages are the built-in classes; every file of the contract -- extractor outputs, detector scores, matches, scores, DR_index, Trans_pre
    byte for byte, the result .npz by content, pre.log as text -- equals what the chain of stage.run() calls (ROREG_EVALUATOR=stages) writes
    from the same generator stream; a second call finds the per-cloud files and reuses them (the reference's skip rule)."""
    import filecmp
    import glob
    from types import SimpleNamespace as NS
    from roreg_amd.test import _cache
    from roreg_amd.test.evaluator import yoho_evaluator
    z = load_golden('pipeline_rd_rm_yohoo' if RM else 'pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, ET=ET, RD=RD, RM=RM, max_iter=1000)
    keynum = int(z['keynum'])
    roots = {}
    for route in ('stages', 'engine'):
        c = NS(**{**vars(cfg), 'output_cache_fn': f'{tmp_path}/cache_{route}'})
        ds.write_inputs(c.output_cache_fn)
        monkeypatch.setenv('ROREG_EVALUATOR', route)
        _cache.clear()
        ev = yoho_evaluator(c)
        assert ev._engine_route() == (route == 'engine')
        np.random.seed(77)
        ev.process_scene(ds)
        roots[route] = f'{c.output_cache_fn}/{ds.name}'
        if route == 'engine':
            before = {f: os.stat(f).st_mtime_ns for f in glob.glob(f'{roots[route]}/YOHO_Output_Group_feature/*.npy') + glob.glob(f'{roots[route]}/det_score/*.npy')}
            np.random.seed(77)
            ev.process_scene(ds)                                          # per-cloud files exist now: reused, not rewritten
            assert before and all(os.stat(f).st_mtime_ns == t for f, t in before.items())
    a, b = roots['stages'], roots['engine']
    rel = sorted(os.path.relpath(f, a) for f in glob.glob(f'{a}/**/*.npy', recursive=True) if 'Input_Group_feature' not in f)
    assert rel == sorted(os.path.relpath(f, b) for f in glob.glob(f'{b}/**/*.npy', recursive=True) if 'Input_Group_feature' not in f)
    assert any(r.startswith('YOHO_Output_Group_feature') for r in rel) and any('DR_index' in r for r in rel) and (not RD or any(r.startswith('det_score') for r in rel))
    for r in rel:
        assert filecmp.cmp(f'{a}/{r}', f'{b}/{r}', shallow=False), r
    rdir = f'match_{keynum}/{ET}/1000iters'
    for p0, p1 in ds.pair_ids:
        x, y = np.load(f'{a}/{rdir}/{p0}-{p1}.npz'), np.load(f'{b}/{rdir}/{p0}-{p1}.npz')
        assert sorted(x.files) == sorted(y.files)
        for k in x.files:
            assert x[k].dtype == y[k].dtype and np.array_equal(x[k], y[k], equal_nan=True), (p0, p1, k)
    assert open(f'{a}/{rdir}/pre.log').read() == open(f'{b}/{rdir}/pre.log').read()


@pytest.mark.parametrize('kind', ['mutual+yohoo', 'mutual+yohoc', 'rd+rm+yohoo'])
def test_run_scenes_pipelined_equals_one_scene_at_a_time(kind):
    """engine.run_scenes (the next scene's extraction and matcher enqueued before the host waits for this scene's match counts; downloads on
    a side stream) gives bitwise the results of run_scene scene by scene, for every estimator / matcher combination (per-pair seeds)."""
    import zlib
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    RD = RM = kind.startswith('rd')
    cfg = default_config(keynum=96, max_iter=300, ET='yohoc' if kind.endswith('yohoc') else 'yohoo', RD=RD, RM=RM)
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    rd = rm = None
    if RD:
        rd = name2network['RD_test'](cfg); rd.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RD').items()})
        rm = name2network['RM_test'](cfg); rm.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()})
    eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
    jobs = []
    for q, n_clouds in enumerate((4, 3, 5, 2)):
        ds = synth.make_scene(40 + q, n_clouds=n_clouds, n_kpts=128 + 16 * q, overlap=0.6)
        seeds = [zlib.crc32(f'{q}:{a}:{b}'.encode()) for a, b in ds.pair_ids]
        jobs.append((ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, dict(keynum=96, max_iter=300, keep_matches=True, pair_seeds=seeds)))
    one = [eng.run_scene(f, k, p, **kw) for f, k, p, kw in jobs]
    piped = eng.run_scenes(jobs)
    assert [len(r) for r in piped] == [len(r) for r in one]
    for ra, rb in zip(one, piped):
        for a, b in zip(ra, rb):
            assert (a.id0, a.id1, a.n_match, a.recalltime) == (b.id0, b.id1, b.n_match, b.recalltime)
            assert np.array_equal(a.trans, b.trans, equal_nan=True) and torch.equal(a.matches, b.matches)


def test_run_plan_splits_a_lone_scene_without_changing_results():
    """distributed.run_plan on a rank that holds ONE scene: the pair list is halved into four jobs that share the scene's cloud cache and
    are pipelined (so the scene's host synchronisations hide behind its own other half); results bitwise those of engine.run_scene on the
    whole list, in the list's order."""
    import zlib
    from roreg_amd import distributed as D
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    cfg = default_config(keynum=128, max_iter=300, ET='yohoo')
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    eng = RegistrationEngine(cfg, gf, et)
    ds = synth.make_scene(9, n_clouds=9, n_kpts=160, overlap=0.6)                    # 36 pairs
    keys = [ds.get_kps(i) for i in ds.pc_ids]
    seeds = [zlib.crc32(f'{a}:{b}'.encode()) for a, b in ds.pair_ids]
    want = eng.run_scene(ds.feats, keys, ds.pair_ids, keynum=128, max_iter=300, keep_matches=True, pair_seeds=seeds)
    calls = []
    orig = eng.run_scenes
    eng.run_scenes = lambda jobs: (calls.append([len((j() if callable(j) else j)[2]) for j in jobs]), orig(jobs))[1]
    got = D.run_plan(eng, [('s', 0, len(ds.pair_ids))], lambda s: (ds.feats, keys, ds.pair_ids, seeds), min_jobs=4, min_pairs=4,
                     keynum=128, max_iter=300, keep_matches=True)
    assert calls == [[9, 9, 9, 9]] and len(got) == 1 and got[0][:3] == ('s', 0, 36)
    for a, b in zip(want, got[0][3]):
        assert (a.id0, a.id1, a.n_match, a.recalltime) == (b.id0, b.id1, b.n_match, b.recalltime)
        assert np.array_equal(a.trans, b.trans, equal_nan=True) and torch.equal(a.matches, b.matches)


def test_engine_yohoc_equals_file_coupled_stages(tmp_path):
    """The rotation-bin estimator inside the device-resident engine against the file-coupled yohoc stages on the same
    generator stream: same matches, same winning try, same transform."""
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.test import name2extractor, name2matcher, name2estimator
    z = load_golden('pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, ET='yohoc')
    keynum = int(z['keynum'])
    np.random.seed(77)
    name2extractor['yoho_des'](cfg).run(ds)
    name2matcher['matmul'](cfg).run(ds, keynum)
    name2estimator['yohoc'](cfg).run(ds, keynum, 1000)
    md = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}'
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    eng = RegistrationEngine(cfg, gf, None)
    np.random.seed(77)
    res = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keynum=keynum, max_iter=1000, keep_matches=True)
    for r in res:
        want = np.load(f'{md}/yohoc/1000iters/{r.id0}-{r.id1}.npz')
        assert np.array_equal(r.matches.cpu().numpy(), np.load(f'{md}/{r.id0}-{r.id1}.npy'))
        assert r.recalltime == int(want['recalltime'])
        assert np.abs(r.trans - want['trans']).max() < 1e-10


@pytest.mark.parametrize('n_kpts,keynum,RD,RM,ET', [(40, 24, False, True, 'yohoo'), (40, 24, False, True, 'yohoc'), (40, 50, True, True, 'yohoc'),
                                                     (250, 150, True, False, 'yohoc'), (250, 260, False, True, 'yohoo')])
def test_engine_equals_stages_on_small_and_degenerate_scenes(tmp_path, n_kpts, keynum, RD, RM, ET):
    """Cases from tools/soak_engine_vs_stages.py (all 48 combinations pass there): tiny clouds where the rotation-coherence matcher
    finds fewer than three matches (the dummy correspondence must be the same sampled row in both paths), refinements on three or
    coplanar inliers (rank-2 cross-covariance: both paths must end in the same LAPACK call), keynum above and below the cloud size."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from soak_engine_vs_stages import run_case
    ok, worst, msgs = run_case(n_kpts, keynum, RD, RM, ET, root=str(tmp_path))
    assert ok, msgs


def test_knn_module_api_shapes():
    from roreg_amd.utils.knn_search import knn_module
    z = load_golden('knn')
    d, idx = knn_module.KNN(1)(torch.from_numpy(z['a_target'].T[None].copy()), torch.from_numpy(z['a_source'].T[None].copy()))
    assert tuple(d.shape) == tuple(z['a_d'].shape) and tuple(idx.shape) == tuple(z['a_idx'].shape)
    assert not d.is_cuda and idx.dtype == torch.int64
    assert np.array_equal(idx.numpy(), z['a_idx'])
    K = torch.from_numpy(z['k5_keys'].T[None].copy())
    d5, idx5 = knn_module.KNN(5)(K, K)
    assert tuple(idx5.shape) == tuple(z['k5_idx'].shape) and np.array_equal(idx5.numpy(), z['k5_idx'])
    assert tuple(d5.shape) == (1, 5, 1, 777)


def test_nms_sample_matches_reference():
    from roreg_amd.test.matcher import NMS_sample
    z = load_golden('nms')
    for num in [700, 600, 400, 150, 20]:
        assert np.array_equal(NMS_sample(num, 5).sample(z['keys'], z['scores']), z[f'idx_{num}'])


def test_group_feature_assembly_with_plugin_backbone(group):
    """N3: rotated-cloud backbone + nearest-point lookup assembles the [N,32,60] input of the path (toy backbone)."""
    from oracle import ref_numpy as O
    from roreg_amd.testset import assemble_group_features
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1, 1, (900, 3)); kps = pts[rng.permutation(900)[:70]] + 0.01 * rng.standard_normal((70, 3))
    Wp = rng.standard_normal((3, 32)).astype(np.float32)

    def backbone(xyz):                      # deterministic stand-in: every 3rd point, features = tanh(xyz @ W)
        d = xyz[::3]
        return d, np.tanh(d @ Wp).astype(np.float32)
    got = assemble_group_features(backbone, pts, kps)
    assert got.shape == (70, 32, 60) and got.dtype == np.float32
    for g in [0, 7, 59]:
        xg = (pts @ group.R[g].T).astype(np.float32); kg = (kps @ group.R[g].T).astype(np.float32)
        d, f = backbone(xg)
        _, idx = O.knn(d, kg, 1)
        assert np.array_equal(got[:, :, g], f[idx])


def test_engine_rd_rm_equals_file_coupled_stages(tmp_path):
    """--RD --RM --ET yohoo: the device-resident engine against the file-coupled stage classes on one RNG stream."""
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.test import name2extractor, name2detector, name2matcher, name2estimator
    z = load_golden('pipeline_rd_rm_yohoo')
    cfg, ds = _setup(tmp_path, z, RD=True, RM=True, ET='yohoo')
    keynum = int(z['keynum'])
    np.random.seed(5)
    name2extractor['yoho_des'](cfg).run(ds)
    name2detector['yoho_det'](cfg).run(ds)
    name2matcher['yoho_mat'](cfg).run(ds, keynum)
    name2estimator['yohoo'](cfg).run(ds, keynum, 1000)
    md = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}'
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    rd = name2network['RD_test'](cfg); rd.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RD').items()})
    rm = name2network['RM_test'](cfg); rm.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()})
    eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
    np.random.seed(5)
    res = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keynum=keynum, max_iter=1000, keep_matches=True)
    for r in res:
        want = np.load(f'{md}/yohoo/1000iters/{r.id0}-{r.id1}.npz')
        assert np.array_equal(r.matches.cpu().numpy(), np.load(f'{md}/{r.id0}-{r.id1}.npy'))
        assert np.array_equal(r.scores, np.load(f'{md}/scores/{r.id0}-{r.id1}.npy'))
        assert r.recalltime == int(want['recalltime'])
        if np.isfinite(want['trans']).all():
            assert np.abs(r.trans - want['trans']).max() < 1e-10


@pytest.mark.parametrize('RD,RM,ET', [(False, False, 'yohoo'), (False, False, 'yohoc'), (True, True, 'yohoo'), (True, False, 'yohoc')])
def test_distributed_driver_equals_evaluator(tmp_path, RD, RM, ET):
    """run_distributed.evaluate (engine + result-table gather, world size 1) gives the metrics of the file-coupled evaluator, for
    both estimators and with the detector / rotation-coherence matcher on."""
    from roreg_amd import run_distributed as RD_
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.test.evaluator import yoho_evaluator
    z = load_golden('pipeline_mutual_yohoo')
    cfg, ds0 = _setup(tmp_path, z, ET=ET, RD=RD, RM=RM, testset='synth')
    ds1 = synth.make_scene(77, n_clouds=3, n_kpts=int(z['n_kpts']), overlap=0.6, name='synth/scene1')
    ds1.write_inputs(cfg.output_cache_fn)
    datasets = {'wholesetname': 'synth', 'scene0': ds0, 'scene1': ds1}
    for d in (ds0, ds1):
        d.gt_dir = f'{tmp_path}/nonexistent/{d.name}/gt.log'
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    rd = rm = None
    if RD:
        rd = name2network['RD_test'](cfg); rd.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RD').items()})
    if RM:
        rm = name2network['RM_test'](cfg); rm.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()})
    np.random.seed(11)
    got = RD_.evaluate(cfg, datasets, RegistrationEngine(cfg, gf, et if ET == 'yohoo' else None, rd_net=rd, rm_net=rm), rank=0, world=1)
    got_files = {(s, a, b): np.load(f'{cfg.output_cache_fn}/{datasets[s].name}/match_{cfg.keynum}/{ET}/1000iters/{a}-{b}.npz')['trans']
                 for s in ('scene0', 'scene1') for a, b in datasets[s].pair_ids}
    # reference-shaped evaluator on a fresh cache
    import shutil
    shutil.rmtree(cfg.output_cache_fn); ds0.write_inputs(cfg.output_cache_fn); ds1.write_inputs(cfg.output_cache_fn)
    ev = yoho_evaluator(cfg)
    np.random.seed(11)
    for d in (ds0, ds1):
        ev.process_scene(d)
    fm, ir, rr = [], [], []
    for d in (ds0, ds1):
        f, i = ev.fmr_ir_scene(d); r, _, _ = ev.rr_scene(d)
        fm.append(f); ir.append(i); rr.append(r)
    assert abs(got['fmr'] - np.mean(fm)) < 1e-12 and abs(got['ir'] - np.mean(ir)) < 1e-12 and abs(got['rr'] - np.mean(rr)) < 1e-12
    for (s, a, b), T in got_files.items():
        want = np.load(f'{cfg.output_cache_fn}/{datasets[s].name}/match_{cfg.keynum}/{ET}/1000iters/{a}-{b}.npz')['trans']
        if np.isfinite(want).all():
            assert np.abs(T - want).max() < 1e-10


def test_evaluate_world_1_equals_world_2_on_device(tmp_path):
    """The real engine behind run_distributed.evaluate(seed=...): two ranks (gloo, both on GPU 0; a scene's pair list is cut across them,
    so its clouds are extracted in different batches on each rank) give bit-identical per-pair results to one rank -- per-keypoint block
    scales (no batch-composition dependence) + per-pair generator streams."""
    from test_host_logic import run_distributed_worlds
    one, two = run_distributed_worlds(tmp_path, 'real', timeout=900)
    assert int(two['split']) == 1
    assert sorted(one.files) == sorted(two.files)
    for k in one.files:
        if k != 'split':
            assert np.array_equal(one[k], two[k]), k


def test_rccl_path_on_one_gpu_forced_collectives(tmp_path):
    """The RCCL code path executed on the one GPU there is (tests/_nccl_worker.py, a fresh process): init_process_group('nccl', world_size 1,
    device_id) + barrier(device_ids); gather_table's device float64 all_gather_into_tensor; EqvExchange's batch_isend_irecv with device
    payloads sent from the rank to itself -- payload still being produced by queued kernels when the send is issued, receive buffer
    poisoned with NaN, consumer enqueued behind wait() with no host synchronisation; run_plan with the real engine through that exchange:
    table, received `eqv` and every pair's result bitwise equal to the plain pass and to the gloo-staged exchange.  Every section runs
    under the hang watchdog (status 1 instead of a hung box)."""
    from test_host_logic import run_forced_collectives
    rc, log = run_forced_collectives('nccl', tmp_path / 'forced.npz', timeout=600)
    assert rc == 0 and 'ok' in log, log
    z = np.load(tmp_path / 'forced.npz')
    assert z['table'].shape == (37, 21) and int(z['raw_exchange_bytes']) == 5000 * 32 * 60 * 4
    assert z['rows'].shape == (25, 20) and np.isfinite(z['rows']).all()


def test_config3_full_benchmark_shape_through_the_distributed_driver(tmp_path):
    """BASELINE config 3's shape on one GPU: 8 scenes with the 3DMatch station counts [60,60,60,55,57,37,66,38] (dataops/dataset.py:152) =
    433 clouds and 1623 pairs through run_distributed.evaluate (shard plan, engine, result table, metrics), at 256 keypoints per cloud so
    that the feature files stay small.  Every pair comes back once, registers, and the plan for 8 ranks is balanced on this very shape."""
    from roreg_amd import run_distributed as RD_, distributed as D
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    root = str(tmp_path)
    cfg = default_config(output_cache_fn=f'{root}/cache', model_fn=f'{root}/ckpt', base_dir=root, SO3_related_files=None, keynum=256, ET='yohoo',
                         testset='synth', max_iter=1000)
    datasets = {'wholesetname': 'synth'}
    for i, (name, nc, npairs) in enumerate(zip(synth.THREEDMATCH_SCENES, synth.THREEDMATCH_CLOUDS, synth.THREEDMATCH_PAIRS)):
        ds = synth.make_scene(300 + i, n_clouds=nc, n_kpts=256, overlap=0.6, name=f'synth/{name}', pair_ids=synth.scene_pair_list(nc, npairs, 900 + i))
        ds.write_inputs(cfg.output_cache_fn)
        ds.gt_dir = f'{root}/nonexistent/{ds.name}/gt.log'
        datasets[name] = ds
    assert sum(len(datasets[s].pair_ids) for s in synth.THREEDMATCH_SCENES) == 1623 and sum(len(datasets[s].pc_ids) for s in synth.THREEDMATCH_SCENES) == 433
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    out = RD_.evaluate(cfg, datasets, RegistrationEngine(cfg, gf, et), rank=0, world=1, seed=3)
    assert out['pairs'] == 1623
    assert out['rr'] > 0.9 and out['fmr'] > 0.95                        # synthetic scenes with 60 % overlap register
    n_files = sum(len([f for f in os.listdir(f'{cfg.output_cache_fn}/{datasets[s].name}/match_256/yohoo/1000iters') if f.endswith('.npz')])
                  for s in synth.THREEDMATCH_SCENES)
    assert n_files == 1623
    # the 8-GPU shard plan of this shape: every pair exactly once, loads within 25 % of the mean (pair + cloud-extraction cost units)
    pc = {s: len(datasets[s].pair_ids) for s in synth.THREEDMATCH_SCENES}; cc = {s: len(datasets[s].pc_ids) for s in synth.THREEDMATCH_SCENES}
    plan = D.shard_scenes(pc, 8, cc, pair_lists={s: datasets[s].pair_ids for s in pc})
    seen = sorted((s, a, b) for r in plan for s, a, b in r)
    for s in pc:
        rs = [(a, b) for t, a, b in seen if t == s]
        assert rs[0][0] == 0 and rs[-1][1] == pc[s] and all(rs[i][1] == rs[i + 1][0] for i in range(len(rs) - 1))
    loads = [sum((b - a) + 6.0 * len({i for pr in datasets[s].pair_ids[a:b] for i in pr}) for s, a, b in r) for r in plan]
    assert max(loads) < 1.3 * np.mean(loads), loads             # extraction is replicated when a scene is cut: ~80 % efficiency at 8 ranks


def test_dropin_end_to_end_on_a_demo_layout(tmp_path, monkeypatch):
    """The reference's entry point flow (parse flags -> yoho_evaluator(cfg).run()) through the drop-in aliases, on a dataset laid out
    like data/origin_data/demo/kitchen (binary PLY clouds, keypoint index files, gt.log) with checkpoints under --model_fn."""
    import sys
    from roreg_amd import dropin
    from roreg_amd.network import name2network
    z = load_golden('pipeline_mutual_yohoo')
    root = str(tmp_path)
    scene = synth.make_scene(31, n_clouds=2, n_kpts=160, overlap=0.7, name='demo/kitchen')
    base = f'{root}/data/origin_data/demo/kitchen'
    os.makedirs(f'{base}/PointCloud'); os.makedirs(f'{base}/Keypoints')
    rng = np.random.default_rng(0)
    for i in range(2):
        kps = scene.get_kps(str(i)).astype(np.float32)
        extra = rng.uniform(0, 3, (300, 3)).astype(np.float32)
        pts = np.concatenate([extra, kps], 0)                       # keypoints are indices into the cloud
        rec = np.zeros(pts.shape[0], dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')])
        rec['x'], rec['y'], rec['z'] = pts[:, 0], pts[:, 1], pts[:, 2]
        hdr = f'ply\nformat binary_little_endian 1.0\nelement vertex {pts.shape[0]}\nproperty float x\nproperty float y\nproperty float z\nend_header\n'
        open(f'{base}/PointCloud/cloud_bin_{i}.ply', 'wb').write(hdr.encode() + rec.tobytes())
        np.savetxt(f'{base}/Keypoints/cloud_bin_{i}Keypoints.txt', np.arange(300, 300 + kps.shape[0]))
    gt = scene.get_transform('0', '1')
    with open(f'{base}/PointCloud/gt.log', 'w') as f:
        f.write('0\t 1\t 2\t\n' + ''.join('\t'.join(repr(float(v)) for v in gt[r]) + '\n' for r in range(3)) + '0.0\t0.0\t0.0\t1.0\n')
    cache = f'{root}/data/YOHO_FCGF/Testset'
    scene.write_inputs(cache)
    cfg0 = default_config()
    for kind, d, seed in [('GF_test', 'GF', 101), ('ET_test', 'ET', 202)]:
        net = name2network[kind](cfg0); synth.seeded_state_dict(net, seed)
        os.makedirs(f'{root}/ckpt/{d}')
        torch.save({'best_para': 0, 'network_state_dict': net.state_dict()}, f'{root}/ckpt/{d}/model_best.pth')
    saved = {k: sys.modules.get(k) for k in dropin._ALIASES}
    monkeypatch.setattr(sys, 'argv', ['Test.py', '--testset', 'demo', '--ET', 'yohoo', '--keynum', '160', '--max_iter', '1000',
                                      '--base_dir', f'{root}/data', '--origin_data_dir', f'{root}/data/origin_data', '--output_cache_fn', cache,
                                      '--model_fn', f'{root}/ckpt', '--SO3_related_files', f'{root}/no_such_dir'])
    try:
        dropin.install()
        import parses.parses_test as parses_test                     # what Test.py does (Test.py:3-4,21-23)
        from test.evaluator import yoho_evaluator
        config, _ = parses_test.get_config()
        np.random.seed(3)
        out = yoho_evaluator(config).run()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    log = open(f'{root}/data/results.log').read()
    assert log.startswith('demo-yoho_des-nodet-matmul-yohoo-160keys-1000iters\n') and 'registration recall(predator)    : 1.00000' in log
    assert out['fmr'] == 1.0 and out['ir'] > 0.8 and out['rr'] == 1.0
    T = np.load(f'{cache}/demo/kitchen/match_160/yohoo/1000iters/0-1.npz')['trans']
    assert np.abs(T[:3] - gt).max() < 1e-3
    assert os.path.exists(f'{base}/Keypoints_PC/cloud_bin_0Keypoints.npy')


def test_pinned_pool_and_staging_ring_reuse_their_buffers():
    """Steady state allocates no pinned memory (a pinned allocation with kernels in flight stalled the launching thread for 5-90 ms): a released
    buffer comes back for the next request of its class, and the staging ring's uploads of alternating sizes settle on pooled buffers while the
    uploaded values stay exact."""
    from roreg_amd import hip
    pool = hip.PinnedPool(max_bytes=1 << 26)
    a = pool.acquire(100_000)
    assert a.is_pinned() and a.shape[0] == hip.PinnedPool.size_class(100_000)
    ptr = a.data_ptr()
    pool.release(a); del a
    b = pool.acquire(99_000)
    assert b.data_ptr() == ptr                                   # (same class: the same buffer)
    big = pool.acquire(1 << 27)                                  # beyond the cap: handed out, not kept
    pool.release(big); pool.release(b)
    assert pool.held == hip.PinnedPool.size_class(99_000)
    rng = np.random.default_rng(0)
    ring = hip._StagingRing(slots=4)
    for rep in range(6):
        for n in (37, 3_000_000, 11, 500_000, 2_000_001):
            x = rng.integers(-9, 9, n).astype(np.int64)
            d = ring.upload(x)
            assert d.dtype == torch.int64 and torch.equal(d.cpu(), torch.from_numpy(x))
    before = hip.pinned_pool.held
    for n in (37, 3_000_000, 11, 500_000, 2_000_001) * 3:
        ring.upload(np.zeros(n, np.int64))
    torch.cuda.synchronize()
    assert hip.pinned_pool.held <= before + 8 * 3_000_000 * 2    # (the pool does not keep growing)
