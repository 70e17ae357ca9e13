"""Soak check (GPU): the device-resident engine against the file-coupled stage classes over scene sizes, keynum and every
detector / matcher / estimator combination, on one generator stream each.  Prints one line per case; exits non-zero on a mismatch.
   python tools/soak_engine_vs_stages.py [n_kpts keynum]"""
import os, sys, tempfile, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch


def run_case(n_kpts, keynum, RD, RM, ET, root=None, n_clouds=4):
    """-> (ok, worst |dT| over the pairs both paths solve, [messages])."""
    from conftest import load_golden
    from roreg_amd import synth
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.parses.parses_test import default_config
    from roreg_amd.test import name2extractor, name2detector, name2matcher, name2estimator, _cache
    root = root or tempfile.mkdtemp()
    cfg = default_config(output_cache_fn=f'{root}/cache', model_fn=f'{root}/ckpt', base_dir=root, SO3_related_files=None,
                         keynum=keynum, bs_GF=50, bs_ET=40, RD=RD, RM=RM, ET=ET)
    nets = {}
    for kind, d, seed in [('GF_test', 'GF', 101), ('ET_test', 'ET', 202)]:
        net = name2network[kind](cfg); synth.seeded_state_dict(net, seed); nets[d] = net
        os.makedirs(f'{root}/ckpt/{d}', exist_ok=True)
        torch.save({'best_para': 0, 'network_state_dict': net.state_dict()}, f'{root}/ckpt/{d}/model_best.pth')
    for d in ['RD', 'RM']:
        os.makedirs(f'{root}/ckpt/{d}', exist_ok=True)
        sd = {k: torch.from_numpy(v) for k, v in load_golden(f'weights_{d}').items()}
        torch.save({'best_para': 0, 'network_state_dict': sd}, f'{root}/ckpt/{d}/model_best.pth')
        nets[d] = name2network[f'{d}_test'](cfg); nets[d].load_state_dict(sd)
    ds = synth.make_scene(n_kpts + keynum, n_clouds=n_clouds, n_kpts=n_kpts, overlap=0.6, name='synth/scene0')
    ds.write_inputs(cfg.output_cache_fn)
    _cache.clear()
    np.random.seed(11)
    name2extractor['yoho_des'](cfg).run(ds)
    if RD:
        name2detector['yoho_det'](cfg).run(ds)
    name2matcher['yoho_mat' if RM else 'matmul'](cfg).run(ds, keynum)
    name2estimator[ET](cfg).run(ds, keynum, 1000)
    md = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}'
    eng = RegistrationEngine(cfg, nets['GF'], nets['ET'], rd_net=nets['RD'] if RD else None, rm_net=nets['RM'] if RM else None)
    np.random.seed(11)
    res = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keynum=keynum, max_iter=1000, keep_matches=True)
    ok, worst, msgs = True, 0.0, []
    for r in res:
        want = np.load(f'{md}/{ET}/1000iters/{r.id0}-{r.id1}.npz')
        stage_m = np.load(f'{md}/{r.id0}-{r.id1}.npy')
        m_ok = np.array_equal(r.matches.cpu().numpy(), stage_m)
        r_ok = r.recalltime == int(want['recalltime'])
        if not (m_ok and r_ok):
            msgs.append(f'pair {r.id0}-{r.id1}: matches equal {m_ok} (engine {r.matches.shape[0]}, stage {stage_m.shape[0]}), '
                        f'recalltime engine {r.recalltime} stage {int(want["recalltime"])}')
        ok &= m_ok and r_ok
        if np.isfinite(want['trans']).all() and int(want['recalltime']) != 50000:
            d = float(np.abs(r.trans - want['trans']).max()); worst = max(worst, d)
            if d >= 1e-9:
                msgs.append(f'pair {r.id0}-{r.id1}: |dT| {d:.2e} with {stage_m.shape[0]} matches')
                ok = False
    return ok, worst, msgs


if __name__ == '__main__':
    cases = [(n, k) for n in (40, 250, 700) for k in (int(n * 0.6), n + 10)]
    if len(sys.argv) > 2:
        cases = [(int(sys.argv[1]), int(sys.argv[2]))]
    bad = 0
    for (n_kpts, keynum), RD, RM, ET in itertools.product(cases, (False, True), (False, True), ('yohoo', 'yohoc')):
        ok, worst, msgs = run_case(n_kpts, keynum, RD, RM, ET)
        print(f'N={n_kpts} keynum={keynum} RD={RD} RM={RM} ET={ET}: {"ok" if ok else "MISMATCH"}  max |dT| {worst:.1e}', flush=True)
        for m in msgs:
            print('   ' + m)
        bad += not ok
    sys.exit(1 if bad else 0)
