"""BASELINE config 5: group features stored and streamed as bfloat16, float32 accumulation (--dtype bf16).

Parity is defined on the bf16-ROUNDED tensors: every stage is compared with the oracle fed the same rounded inputs, so indices are
bit-exact again and floating-point outputs keep their float32 tolerances.  The only new rounding steps are (R1) the FCGF-like input
when it is taken in and (R2) the extractor's output when it is stored.  GPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_numpy as O
from roreg_amd import synth
from roreg_amd.parses.parses_test import default_config

pytestmark = pytest.mark.gpu


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bf16_round(a):
    """float32 ndarray -> the float32 values a bfloat16 tensor of it holds (round to nearest even)."""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def test_extractor_bf16_io_vs_oracle_on_rounded_input(group):
    """GF on a bfloat16 input, output stored as bfloat16, against bf16(oracle(bf16(x))): equal except where the float32 value sits on a
    bfloat16 rounding boundary (then one bf16 ulp apart), and always within one ulp."""
    from roreg_amd.network import name2network
    net = name2network['GF_test'](default_config())
    sd = {k: v.numpy() for k, v in synth.seeded_state_dict(net, 101).items()}
    rng = np.random.default_rng(3)
    x = rng.standard_normal((300, 32, 60)).astype(np.float32)
    x /= np.sqrt((x * x).sum(1, keepdims=True))
    xb = bf16_round(x)
    got = net.PartI_net(cu(x).to(torch.bfloat16), out_dtype=torch.bfloat16)
    assert got['eqv'].dtype == torch.bfloat16
    want32 = O.gf_forward(xb, sd, group.Nei)['eqv']
    g = got['eqv'].float().cpu().numpy(); w = bf16_round(want32)
    ulp = np.abs(w) * 2.0 ** -7 + 2e-6                                       # >= one bfloat16 ulp of the value, + the float32 evaluation noise near 0
    assert (np.abs(g - w) <= ulp).all()
    assert np.mean(g != w) < 0.01
    assert np.abs(g - want32).max() < 2.0 ** -8                                # i.e. the stored value is the float32 result to bf16 precision


def test_pair_stages_bit_exact_on_bf16_descriptors(group):
    """Matcher descriptor, mutual matches and Des2R on bfloat16-stored features equal the oracle's on the same rounded values, bit for
    bit; the ET input assembly copies the stored values; local transforms within the float32 tolerance."""
    from roreg_amd import hip
    from roreg_amd.network import name2network
    ds = synth.make_scene(11, n_clouds=2, n_kpts=700, overlap=0.6)
    e0, e1 = bf16_round(ds.feats[0]), bf16_round(ds.feats[1])                     # the stored descriptors (used as 'eqv' and as 'before')
    E0, E1 = cu(e0).to(torch.bfloat16), cu(e1).to(torch.bfloat16)
    assert np.array_equal(hip.inv_descriptor(E0).cpu().numpy(), O.inv_descriptor(e0))
    s = np.arange(700)
    want = O.mutual_match(e0, e1, s, s)
    buf, cnt = hip.mutual_match_batch([(hip.inv_descriptor(E0), hip.inv_descriptor(E1), None, None)])
    m = buf[0, :int(cnt.item())].cpu().numpy()
    assert np.array_equal(m, want) and m.shape[0] > 100
    r0, r1 = cu(m[:, 0].copy()), cu(m[:, 1].copy())
    dr = hip.des2r(E1, E0, rows1=r1, rows0=r0, coefs1=hip.feat_coefs(E1), coefs0=hip.feat_coefs(E0))
    assert np.array_equal(dr.cpu().numpy(), O.des2r(e1[m[:, 1]], e0[m[:, 0]], group.P))
    x = hip.et_gather(E0, E1, E0, E1, dr, rows0=r0, rows1=r1).cpu().numpy()
    P = group.P[dr.cpu().numpy()]
    assert np.array_equal(x[:, 32:64], e0[m[:, 0]]) and np.array_equal(x[:, 0:32], np.take_along_axis(e1[m[:, 1]], P[:, None, :], 2))
    et = name2network['ET_test'](default_config())
    sdn = {k: v.numpy() for k, v in synth.seeded_state_dict(et, 202).items()}
    q = et.trunk_and_head(cu(x))
    q = (q / torch.norm(q, dim=1)[:, None]).cpu().numpy()
    n = 64
    batch = {'before_eqv0': e1[m[:n, 1]], 'before_eqv1': e0[m[:n, 0]], 'after_eqv0': e1[m[:n, 1]], 'after_eqv1': e0[m[:n, 0]], 'pre_idx': dr.cpu().numpy()[:n]}
    assert np.abs(q[:n] - O.et_forward(batch, sdn, group.Nei, group.P)).max() < 1e-4


@pytest.mark.parametrize('RD,RM', [(False, False), (True, True)])
def test_engine_equals_stage_classes_in_bf16_mode(tmp_path, RD, RM):
    """--dtype bf16 end to end: the device-resident engine and the file-coupled stage classes (float32 .npy files holding bf16-representable
    values) give the same matches and registrations from the same generator stream, and the stored YOHO features are exactly bf16."""
    from conftest import load_golden
    from test_hip_pipeline import _setup
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from roreg_amd.test import name2extractor, name2detector, name2matcher, name2estimator
    z = load_golden('pipeline_rd_rm_yohoo' if RM else 'pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, ET='yohoo', RD=RD, RM=RM, dtype='bf16')
    keynum = int(z['keynum'])
    np.random.seed(99)
    name2extractor['yoho_des'](cfg).run(ds)
    y = np.load(f'{cfg.output_cache_fn}/{ds.name}/YOHO_Output_Group_feature/0.npy')
    assert y.dtype == np.float32 and np.array_equal(y, bf16_round(y))
    if RD:
        name2detector['yoho_det'](cfg).run(ds)
    name2matcher['yoho_mat' if RM else 'matmul'](cfg).run(ds, keynum)
    name2estimator['yohoo'](cfg).run(ds, keynum, 1000)
    md = f'{cfg.output_cache_fn}/{ds.name}/match_{keynum}'
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    rd = rm = None
    if RD:
        rd = name2network['RD_test'](cfg); rd.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RD').items()})
    if RM:
        rm = name2network['RM_test'](cfg); rm.load_state_dict({k: torch.from_numpy(v) for k, v in load_golden('weights_RM').items()})
    eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
    assert eng.feat_dtype == torch.bfloat16
    np.random.seed(99)
    res = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keynum=keynum, max_iter=1000, keep_matches=True)
    for r in res:
        want = np.load(f'{md}/yohoo/1000iters/{r.id0}-{r.id1}.npz')
        assert np.array_equal(r.matches.cpu().numpy(), np.load(f'{md}/{r.id0}-{r.id1}.npy'))
        assert r.recalltime == int(want['recalltime'])
        if np.isfinite(want['trans']).all():
            assert np.abs(r.trans - want['trans']).max() < 1e-10


def test_config5_outdoor_scene_with_bf16_descriptors(group):
    """BASELINE config 5 as stated: ETH-like scale (30 m extent, 5 cm coordinate noise, ransac_ird 0.5), 5000 keypoints, descriptors in
    bfloat16 end to end: every pair registers well inside the ETH success bounds, and the results agree with the float32 run to the
    registration tolerance (the descriptors differ by the bf16 rounding, so match lists differ slightly)."""
    from roreg_amd.engine import RegistrationEngine
    from roreg_amd.network import name2network
    from test_hip_edge_cases import _pose_error
    cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo', ransac_ird=0.5)
    gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
    et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
    ds = synth.make_scene(57, n_clouds=3, n_kpts=5000, overlap=0.6, coord_noise=0.05, extent=30.0)
    eng = RegistrationEngine(cfg, gf, et)
    out = {}
    for name in ('fp32', 'bf16'):
        eng.set_descriptor_dtype(name)
        np.random.seed(5)
        out[name] = eng.run_scene(ds.feats, [ds.get_kps(i) for i in ds.pc_ids], ds.pair_ids, keep_matches=True)
    for a, b in zip(out['fp32'], out['bf16']):
        rre, rte = _pose_error(b.trans, ds.get_transform(b.id0, b.id1))
        assert rre < 0.5 and rte < 0.1, (b.id0, b.id1, rre, rte)
        assert b.n_match > 2500
        ma = {tuple(r) for r in a.matches.cpu().numpy()}; mb = {tuple(r) for r in b.matches.cpu().numpy()}
        assert len(ma & mb) > 0.9 * len(ma)
