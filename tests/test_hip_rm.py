"""Rotation-coherence matcher (Match_ot) on HIP against the oracle and the reference's golden run.  GPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import match_ot_numpy as MO
from roreg_amd.parses.parses_test import default_config

pytestmark = pytest.mark.gpu


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope='module')
def rm():
    from roreg_amd.network import name2network
    net = name2network['RM_test'](default_config())
    sd = dict(load_golden('weights_RM'))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.eval()
    return net, sd


@pytest.mark.parametrize('m,n,k', [(300, 257, 16), (5000, 5000, 8), (40, 17, 16), (129, 700, 1)])
def test_topk_dot(m, n, k):
    from roreg_amd import hip
    rng = np.random.default_rng(m + n + k)
    A = rng.standard_normal((m, 32)).astype(np.float32); B = rng.standard_normal((n, 32)).astype(np.float32)
    B[n // 2] = B[1]                                     # an exact tie: lower index first
    idx, val = hip.topk_dot(cu(A), cu(B), k, want_val=True)
    idx = idx.cpu().numpy(); val = val.cpu().numpy()
    S = (A.astype(np.float64) @ B.T.astype(np.float64))
    want = MO.topk_rows(S.astype(np.float32), k)
    got_vals = np.take_along_axis(S, idx, 1)
    assert np.abs(val - got_vals).max() < 1e-4
    assert (np.diff(val, axis=1) <= 0).all()
    # sets agree except where fp32 vs fp64 dot products reorder near-ties at the boundary
    agree = np.mean([len(set(a) & set(b)) == k for a, b in zip(idx, want)])
    assert agree > 0.995
    rows_with_tie = [i for i in range(m) if 1 in idx[i] and n // 2 in idx[i]]
    for i in rows_with_tie:
        assert list(idx[i]).index(1) < list(idx[i]).index(n // 2)


def test_topk_on_the_matrix_cores_is_bitwise_the_vector_pipe_search(tmp_path):
    """roreg_topk_dot's default kernel forms the dot products as float32 MFMA chains (= the fmaf chains of the vector-pipe kernel, bit for bit) and
    selects per lane; against ROREG_TOPK_VALU=1 (a child process): indices AND values bitwise, k = 16 / 8 / 1, stacked ragged pairs (segments),
    exact ties (duplicated targets: lower index first) within and across the two half-wave lists and across slices, sizes on both sides of the
    32-row tile and the 128-source workgroup."""
    import subprocess, sys
    from roreg_amd import hip
    rng = np.random.default_rng(91)
    sizes = [(300, 257), (64, 190), (513, 2500), (2500, 33), (129, 5000), (17, 40)]
    A = [rng.standard_normal((m, 32)).astype(np.float32) for m, _ in sizes]
    B = [rng.standard_normal((n, 32)).astype(np.float32) for _, n in sizes]
    for b in B:
        n = b.shape[0]
        b[n // 2] = b[1]; b[n - 1] = b[0]; b[min(n - 1, 36)] = b[4]        # exact ties: same tile / other half-wave / far apart
    np.savez(tmp_path / 'in.npz', A=np.concatenate(A), B=np.concatenate(B), m=np.array([a for a, _ in sizes]), n=np.array([b for _, b in sizes]))
    code = ("import numpy as np, torch, sys\n"
            "from roreg_amd import hip\n"
            "z = np.load(sys.argv[1]); A = torch.from_numpy(z['A']).cuda(); B = torch.from_numpy(z['B']).cuda()\n"
            "out = {}\n"
            "for k in (16, 8, 1):\n"
            "    i, v = hip.topk_dot(A, B, k, want_val=True, segA=hip.Segments(z['m']), segB=hip.Segments(z['n']))\n"
            "    out[f'i{k}'] = i.cpu().numpy(); out[f'v{k}'] = v.cpu().numpy()\n"
            "    i, v = hip.topk_dot(A[:300].contiguous(), B[:257].contiguous(), k, want_val=True)\n"
            "    out[f'si{k}'] = i.cpu().numpy(); out[f'sv{k}'] = v.cpu().numpy()\n"
            "np.savez(sys.argv[2], **out)\n")
    env = dict(os.environ, ROREG_TOPK_VALU='1', PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(tmp_path / 'out.npz')
    cA, cB = cu(np.concatenate(A)), cu(np.concatenate(B))
    for k in (16, 8, 1):
        i, v = hip.topk_dot(cA, cB, k, want_val=True, segA=hip.Segments([a for a, _ in sizes]), segB=hip.Segments([b for _, b in sizes]))
        assert np.array_equal(i.cpu().numpy(), z[f'i{k}']) and np.array_equal(v.cpu().numpy(), z[f'v{k}']), k
        i, v = hip.topk_dot(cA[:300].contiguous(), cB[:257].contiguous(), k, want_val=True)
        assert np.array_equal(i.cpu().numpy(), z[f'si{k}']) and np.array_equal(v.cpu().numpy(), z[f'sv{k}']), k


def test_group_corr_transposed_is_r_indicator(group):
    from roreg_amd import hip
    rng = np.random.default_rng(5)
    own = rng.standard_normal((70, 32, 60)).astype(np.float32); other = rng.standard_normal((90, 32, 60)).astype(np.float32)
    nn = rng.integers(0, 90, 70)
    for s2t in [True, False]:
        want = MO.r_indicator(own, other[nn], group.P, s2t)
        if s2t:
            got = hip.group_corr(cu(own), cu(other), perm_rows=None, bcast_rows=cu(nn), transpose=True)
        else:
            got = hip.group_corr(cu(other), cu(own), perm_rows=cu(nn), bcast_rows=None, transpose=True)
        assert np.abs(got.cpu().numpy() - want).max() < 2e-4


def test_group_corr_as_one_matrix_product_per_point(group):
    """R_indicator as C = A^T B (one 60 x 32 x 60 float32 MFMA product per point) + the 60 coset sums cor[a] = sum_g C[T[a,g], g]
    (csrc/corr_mfma.hip) against the literal kernel and against float64: both table orientations, with and without row lists, a point count
    that is not a multiple of the workgroup's four, values spanning 1e-2 .. 1e2 per point; per-element error relative to |A||B| at the
    float32 level -- and no worse than the literal kernel's own."""
    from roreg_amd import hip
    rng = np.random.default_rng(18)
    A = (rng.standard_normal((403, 32, 60)) * 10.0 ** rng.integers(-2, 3, (403, 1, 1))).astype(np.float32)
    B = (rng.standard_normal((350, 32, 60)) * 10.0 ** rng.integers(-2, 3, (350, 1, 1))).astype(np.float32)
    Ad, Bd = cu(A), cu(B)
    rows = rng.integers(0, 350, 403)
    for tr in (True, False):
        T = group.P.T if tr else group.P                                         # T[a, g]
        for perm_first in (True, False):
            if perm_first:
                lit = hip.group_corr(Ad, Bd, perm_rows=None, bcast_rows=cu(rows), transpose=tr)
                with hip.matrix_core_layers():
                    got = hip.group_corr(Ad, Bd, perm_rows=None, bcast_rows=cu(rows), transpose=tr)
                X, Y = A.astype(np.float64), B[rows].astype(np.float64)
            else:
                lit = hip.group_corr(Bd, Ad, perm_rows=cu(rows), bcast_rows=None, transpose=tr)
                with hip.matrix_core_layers():
                    got = hip.group_corr(Bd, Ad, perm_rows=cu(rows), bcast_rows=None, transpose=tr)
                X, Y = B[rows].astype(np.float64), A.astype(np.float64)
            ref = np.einsum('bfag,bfg->ba', X[:, :, T], Y)                           # sum_f sum_g X[f, T[a,g]] Y[f,g]
            scale = (np.sqrt((X * X).sum((1, 2))) * np.sqrt((Y * Y).sum((1, 2))))[:, None]
            e_new = float((np.abs(got.double().cpu().numpy() - ref) / scale).max()); e_lit = float((np.abs(lit.double().cpu().numpy() - ref) / scale).max())
            assert e_new < 2e-6 and e_new < 2 * e_lit + 2e-7, (tr, perm_first, e_new, e_lit)


def test_group_corr_irrep_equals_literal_to_rounding(group):
    """R_indicator in the irrep domain (both table orientations, with and without row lists) against the literal float32 kernel: equal to
    the float32 rounding level of a 1920-term sum (the literal kernel itself is bit-exact against the oracle, test above)."""
    from roreg_amd import hip
    rng = np.random.default_rng(8)
    A = rng.standard_normal((400, 32, 60)).astype(np.float32); B = rng.standard_normal((350, 32, 60)).astype(np.float32)
    Ad, Bd = cu(A), cu(B)
    ca, cb = hip.feat_coefs(Ad), hip.feat_coefs(Bd)
    rows = cu(rng.integers(0, 350, 400))
    for tr in (True, False):
        lit = hip.group_corr(Ad, Bd, perm_rows=None, bcast_rows=rows, transpose=tr)
        irr = hip.group_corr(Ad, Bd, perm_rows=None, bcast_rows=rows, transpose=tr, perm_coefs=ca, bcast_coefs=cb)
        scale = float(torch.linalg.vector_norm(Ad, dim=(1, 2)).max() * torch.linalg.vector_norm(Bd, dim=(1, 2)).max())
        assert float((lit - irr).abs().max()) < 5e-6 * scale
        lit = hip.group_corr(Bd, Ad, perm_rows=rows, bcast_rows=None, transpose=tr)
        irr = hip.group_corr(Bd, Ad, perm_rows=rows, bcast_rows=None, transpose=tr, perm_coefs=cb, bcast_coefs=ca)
        assert float((lit - irr).abs().max()) < 5e-6 * scale


def test_mlp_instnorm_and_attention(rm):
    from roreg_amd import hip
    net, sd = rm
    rng = np.random.default_rng(6)
    blk = net.Graph.merge_blocks[0].self_graph_s
    name = 'Graph.merge_blocks.0.self_graph_s'
    x = rng.standard_normal((513, 96)).astype(np.float32)
    assert np.abs(blk.val_en(cu(x)).cpu().numpy() - MO.mlp_2layer(x, sd, name + '.val_en')).max() < 1e-4
    x3 = rng.standard_normal((700, 3)).astype(np.float32) * 10
    assert np.abs(blk.pos_en(cu(x3)).cpu().numpy() - MO.mlp_2layer(x3, sd, name + '.pos_en')).max() < 1e-4
    xc = rng.standard_normal((333, 120)).astype(np.float32)
    assert np.abs(blk.ambiguity(cu(xc)).cpu().numpy() - MO.mlp_2layer(xc, sd, name + '.ambiguity')).max() < 1e-4
    m, k = 200, 16
    q = rng.standard_normal((m, 32)).astype(np.float32); table = rng.standard_normal((m, 32)).astype(np.float32)
    val = rng.standard_normal((m, k, 32)).astype(np.float32); idx = rng.integers(0, m, (m, k))
    want = MO.mha(q, table[idx], val, sd, name + '.self_attn')
    got = blk.self_attn(cu(q), cu(table), cu(val.reshape(m * k, 32)), cu(idx), k, True, False).cpu().numpy()
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize('cin,cout', [(32, 32), (96, 64), (120, 128), (64, 64), (96, 32), (120, 32), (64, 32)])
def test_linear_layers_on_the_matrix_cores_are_f32_accurate(cin, cout):
    """The 1x1 layers as fp16 hi + lo MFMAs against float64, PER ELEMENT relative to the row's own |W||x| scale: rows spanning 1e-3 .. 1e3 in
    magnitude in one call, rows beyond fp16's range (1e6: the power-of-two row scale), a ragged row count; error at the level of a float32
    fmaf chain (<= 6e-7 of sum |w||x| per element)."""
    from roreg_amd import hip
    g = torch.Generator(device='cuda').manual_seed(100 * cin + cout)
    L = 4133
    x = torch.randn((L, cin), device='cuda', generator=g)
    x *= torch.pow(10.0, torch.randint(-3, 4, (L, 1), device='cuda', generator=g).float())
    x[7] *= 1e3; x[100:110] *= 1e3                                            # 1e6-sized rows: beyond fp16
    W = torch.randn((cout, cin), device='cuda', generator=g) * 0.3; b = torch.randn(cout, device='cuda', generator=g)
    with hip.matrix_core_layers():
        y = hip.linear(x, W, b).double()
        assert torch.equal(hip.linear(x[:33].contiguous(), W, b), hip.linear(x, W, b)[:33])      # a row's result depends on that row alone
    ref = x.double() @ W.double().t() + b.double()
    scale = x.double().abs() @ W.double().abs().t() + b.double().abs()
    err = ((y - ref).abs() / scale).max().item()
    assert err < 6e-7, err


@pytest.mark.parametrize('cin,cout', [(32, 32), (96, 64), (120, 128), (64, 64), (96, 32), (120, 32), (64, 32), (3, 64), (3, 32)])
def test_fmaf_chains_on_the_matrix_cores_are_bitwise_the_vector_pipe_chains(cin, cout, rm):
    """The default 1x1 layers: one float32 fmaf chain per (row, output), inputs ascending, starting from the bias -- evaluated by
    v_mfma_f32_32x32x2_f32, which IS such a chain on gfx950 (csrc/linear_chain.hip, tools/probe/mfma_f32_order.hip).  BITWISE the
    vector-pipe kernels (hip.vector_pipe_layers) on rows spanning 1e-3 .. 1e6 in magnitude, float32 subnormals and exact zeros, ragged row
    counts on both sides of the 128-row tile and of the 65536-row switch of the vector-pipe kernels; and the InstanceNorm + ReLU + second
    conv + residual of mlp_2layer with per-pair statistics (segments cutting through row tiles)."""
    from roreg_amd import hip
    g = torch.Generator(device='cuda').manual_seed(1000 * cin + cout)
    for L in (1, 127, 129, 4133, 70001):
        x = torch.randn((L, cin), device='cuda', generator=g)
        x *= torch.pow(10.0, torch.randint(-3, 4, (L, 1), device='cuda', generator=g).float())
        if L > 200:
            x[7] *= 1e3; x[100:110] *= 1e-38; x[120:125] = 0.0                    # 1e6-sized rows, subnormal products, exact zeros
        W = torch.randn((cout, cin), device='cuda', generator=g) * 0.3; b = torch.randn(cout, device='cuda', generator=g)
        y = hip.linear(x, W, b)
        with hip.vector_pipe_layers():
            want = hip.linear(x, W, b)
        assert torch.equal(y, want), (cin, cout, L, float((y - want).abs().max()))
        ref = x.double() @ W.double().t() + b.double()
        scale = x.double().abs() @ W.double().abs().t() + b.double().abs()
        assert ((y.double() - ref).abs() / scale).max().item() < 4e-6
    if (cin, cout) in ((64, 32), (120, 128)):                                        # mlp_2layer (64 -> 64 -> 32 and 120 -> 128 -> 32) with per-pair statistics
        cmid = 64 if cin == 64 else 128
        sizes = [900, 37, 5000, 130, 2500]
        seg = hip.Segments(sizes)
        for mult in (1, 16):
            x = torch.randn((sum(sizes) * mult, cin), device='cuda', generator=g)
            W1 = torch.randn((cmid, cin), device='cuda', generator=g) * 0.2; b1 = torch.randn(cmid, device='cuda', generator=g)
            W2 = torch.randn((32, cmid), device='cuda', generator=g) * 0.2; b2 = torch.randn(32, device='cuda', generator=g)
            Wr = torch.randn((32, cin), device='cuda', generator=g) * 0.2; br = torch.randn(32, device='cuda', generator=g)
            if cin == 120:
                continue                                                          # (a 120 -> 128 first conv exists, its residual branch is 120 -> 32)
            y = hip.mlp_instnorm(x, W1, b1, W2, b2, Wr, br, seg=seg)
            with hip.vector_pipe_layers():
                want = hip.mlp_instnorm(x, W1, b1, W2, b2, Wr, br, seg=seg)
            assert torch.equal(y, want), (cin, mult)


@pytest.mark.parametrize('cin,cmid', [(3, 64), (64, 64), (96, 64), (120, 128), ('cat3', 64)])
def test_pipelined_chain_kernels_equal_round5s_kernels(cin, cmid):
    """Round 6 (csrc/linear_chain.hip, lc2_kernel): the float32 fmaf chains software-pipelined, mlp_2layer's first convolution and residual branch
    in one launch (roreg_mlp_head) with the InstanceNorm statistics of h taken from the tiles' float64 channel sums.  Against round 5's kernels
    (hip.round5_chain_layers: one launch per convolution, roreg_instnorm_stats): every plain layer bit for bit; the whole mlp_2layer bit for
    bit too on these inputs (the statistics are the same float64 sums in another association -- equal after rounding to float32 but for one
    value in ~1e8) -- on ragged stacked pairs whose tiles end inside a 128-row tile, k-expanded rows (mult 16 / 8), one-row and empty-ish
    pairs, rows spanning 1e-3 .. 1e3 with subnormals and zeros; a pair's result does not depend on what is stacked beside it."""
    from roreg_amd import hip
    g = torch.Generator(device='cuda').manual_seed(7 + (0 if cin == 'cat3' else cin))
    rnd = lambda *shape, s=1.0: torch.randn(shape, device='cuda', generator=g) * s
    sizes = [900, 37, 5000, 1, 130, 2500, 129]
    seg = hip.Segments(sizes)
    for mult in ((16, 8) if cin in (3, 'cat3') else (1,)):
        m = sum(sizes); L = m * mult
        if cin == 'cat3':
            pos = rnd(L, 32); table = rnd(m, 32); conf = rnd(m, 32)
            idx = torch.randint(0, m, (m, mult), device='cuda', generator=g)
            x = hip.Cat3Rows(pos, table, conf, idx); ci = 96
            x_plain = x.materialise().clone()
        else:
            ci = cin
            x = rnd(L, ci) * torch.pow(10.0, torch.randint(-3, 4, (L, 1), device='cuda', generator=g).float())
            x[7] *= 1e3; x[100:110] *= 1e-38; x[120:125] = 0.0
            x_plain = x
        W1 = rnd(cmid, ci, s=0.2); b1 = rnd(cmid); W2 = rnd(32, cmid, s=0.2); b2 = rnd(32); Wr = rnd(32, ci, s=0.2); br = rnd(32)
        y = hip.mlp_instnorm(x, W1, b1, W2, b2, Wr, br, seg=seg)
        with hip.round5_chain_layers():
            want = hip.mlp_instnorm(x_plain, W1, b1, W2, b2, Wr, br, seg=seg)
            want_h = hip.linear(x_plain, W1, b1)
        assert torch.equal(y, want), (cin, mult, float((y - want).abs().max()))
        assert torch.equal(hip.linear(x_plain, W1, b1), want_h)
        # one pair alone: the same rows, the same statistics
        q = 2
        o = int(seg.host[q]) * mult; n = sizes[q] * mult
        if cin == 'cat3':
            lo = int(seg.host[q])
            xa = hip.Cat3Rows(pos[o:o + n].contiguous(), table[lo:lo + sizes[q]].contiguous(), conf[lo:lo + sizes[q]].contiguous(), (idx[lo:lo + sizes[q]] % sizes[q]).contiguous())
            xs = hip.Cat3Rows(pos, table, conf, torch.cat([idx[:lo], idx[lo:lo + sizes[q]] % sizes[q] + lo, idx[lo + sizes[q]:]]).contiguous())
            ys = hip.mlp_instnorm(xs, W1, b1, W2, b2, Wr, br, seg=seg)
            ya = hip.mlp_instnorm(xa, W1, b1, W2, b2, Wr, br, seg=hip.Segments([sizes[q]]))
            assert torch.equal(ys[o:o + n], ya)
        else:
            ya = hip.mlp_instnorm(x[o:o + n].contiguous(), W1, b1, W2, b2, Wr, br, seg=hip.Segments([sizes[q]]))
            assert torch.equal(y[o:o + n], ya)
        # no segments: one InstanceNorm over everything
        y1 = hip.mlp_instnorm(x_plain[:4133].contiguous(), W1, b1, W2, b2, Wr, br)
        with hip.round5_chain_layers():
            w1 = hip.mlp_instnorm(x_plain[:4133].contiguous(), W1, b1, W2, b2, Wr, br)
        assert torch.equal(y1, w1)


@pytest.mark.parametrize('m,k', [(1, 16), (777, 16), (5000, 8), (40001, 16)])
def test_value_rows_assembled_in_the_layer_are_bitwise_the_materialised_rows(m, k, rm):
    """hip.value_input() no longer builds the attention blocks' [m k, 96] value-MLP input (rot_coh_match.py:95-119): linear() gets the three sources
    and roreg_linear_cat3 assembles a row while it stages it.  Both first layers (96 -> 64, 96 -> 32) and the whole mlp_2layer BITWISE those on
    the materialised rows -- ragged row counts, neighbour indices with repeats -- and the vector-pipe path (which materialises) agrees too."""
    from roreg_amd import hip
    g = torch.Generator(device='cuda').manual_seed(7 * m + k)
    n = max(m, 3)
    pos = torch.randn((m * k, 32), device='cuda', generator=g); table = torch.randn((n, 32), device='cuda', generator=g)
    conf = torch.randn((m, 32), device='cuda', generator=g)
    idx = torch.randint(0, n, (m, k), device='cuda', generator=g)
    rows = hip.value_input(pos, table, conf, idx)
    full = hip.value_input(pos, table, conf, idx, materialise=True)
    want = torch.cat([pos, table[idx.reshape(-1)], conf.repeat_interleave(k, 0)], 1)
    assert torch.equal(full, want)
    for cout in (64, 32):
        W = torch.randn((cout, 96), device='cuda', generator=g) * 0.3; b = torch.randn(cout, device='cuda', generator=g)
        assert torch.equal(hip.linear(rows, W, b), hip.linear(full, W, b))
    W1 = torch.randn((64, 96), device='cuda', generator=g) * 0.2; b1 = torch.randn(64, device='cuda', generator=g)
    W2 = torch.randn((32, 64), device='cuda', generator=g) * 0.2; b2 = torch.randn(32, device='cuda', generator=g)
    Wr = torch.randn((32, 96), device='cuda', generator=g) * 0.2; br = torch.randn(32, device='cuda', generator=g)
    y = hip.mlp_instnorm(hip.value_input(pos, table, conf, idx), W1, b1, W2, b2, Wr, br)
    assert torch.equal(y, hip.mlp_instnorm(full, W1, b1, W2, b2, Wr, br))
    with hip.vector_pipe_layers():
        assert torch.equal(y, hip.mlp_instnorm(hip.value_input(pos, table, conf, idx), W1, b1, W2, b2, Wr, br))


def test_mlp_tail_on_the_matrix_cores(rm):
    """mlp_2layer (conv -> InstanceNorm -> ReLU -> conv + residual conv; rot_coh_match.py:14-32) with the shipped final_mlp weights against
    a float64 torch evaluation of the same formula."""
    from roreg_amd import hip
    net, sd = rm
    mlp = net.final_mlp
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn((3001, 64), device='cuda', generator=g) * 2.0
    with torch.no_grad(), hip.matrix_core_layers():
        got = mlp(x).double()
    w = {k.split('final_mlp.')[1]: torch.from_numpy(v).double().cuda() for k, v in sd.items() if k.startswith('final_mlp.')}
    h = x.double() @ w['net.0.weight'][:, :, 0, 0].t() + w['net.0.bias']
    h = (h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + 1e-5)
    ref = torch.relu(h) @ w['net.3.weight'][:, :, 0, 0].t() + w['net.3.bias'] + x.double() @ w['res.weight'][:, :, 0, 0].t() + w['res.bias']
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_many_row_layers_equal_the_per_row_kernels(rm):
    """A 1x1 layer's result for a row depends on that row alone, whatever the number of rows in the call (several pairs stacked: >= 65536
    rows), in BOTH forms of the layer: the vector-pipe kernels (LDS-tiled at >= 65536 rows: same fmaf chains as the one-thread-per-row
    kernel of short calls) and the matrix-core kernel (fixed k order, no cross-row arithmetic, per-row scale) -- big calls are bitwise the
    concatenation of short ones; so is the row normalisation."""
    from roreg_amd import hip
    g = torch.Generator(device='cuda').manual_seed(3)
    L = 70001
    for cin, cout in [(32, 32), (96, 64), (120, 128), (64, 64), (96, 32), (120, 32), (64, 32)]:
        x = torch.randn((L, cin), device='cuda', generator=g); W = torch.randn((cout, cin), device='cuda', generator=g) * 0.2
        b = torch.randn(cout, device='cuda', generator=g)
        big = hip.linear(x, W, b)
        small = torch.cat([hip.linear(x[i:i + 9000].contiguous(), W, b) for i in range(0, L, 9000)])
        assert torch.equal(big, small), (cin, cout)
        with hip.matrix_core_layers():
            big_m = hip.linear(x, W, b)
            small_m = torch.cat([hip.linear(x[i:i + 9000].contiguous(), W, b) for i in range(0, L, 9000)])
        assert torch.equal(big_m, small_m) and (big_m - big).abs().max() < 1e-4, (cin, cout)
        ref = x.double() @ W.double().t() + b.double()
        assert (big.double() - ref).abs().max() < 1e-4
    x = torch.randn((L, 32), device='cuda', generator=g)
    assert torch.equal(hip.l2_normalize_rows(x), torch.cat([hip.l2_normalize_rows(x[i:i + 9000].contiguous()) for i in range(0, L, 9000)]))
    # mlp_2layer with per-pair statistics: 8 pairs of 9000 rows stacked == pair by pair
    net, sd = rm
    mlp = net.final_mlp
    x = torch.randn((72000, 64), device='cuda', generator=g)
    seg = hip.Segments([9000] * 8)
    for on in (False, True):
        with torch.no_grad(), hip.matrix_core_layers(on):
            big = mlp(x, seg=seg)
            small = torch.cat([mlp(x[i:i + 9000].contiguous()) for i in range(0, 72000, 9000)])
        assert torch.equal(big, small), on


def test_sinkhorn_and_readout():
    from roreg_amd import hip
    rng = np.random.default_rng(7)
    for m, n in [(200, 173), (64, 300)]:
        s = rng.standard_normal((m, 32)).astype(np.float32) * 0.5; t = rng.standard_normal((n, 32)).astype(np.float32) * 0.5
        t[:min(m, n) // 2] = s[:min(m, n) // 2] * 3           # planted strong matches
        Z, m0, m1, s0, s1 = hip.sinkhorn(cu(s), cu(t), 1.5, 100)
        want = MO.log_sinkhorn((s @ t.T).astype(np.float32), np.float32(1.5), 100)
        assert np.abs(Z.cpu().numpy() - want).max() < 2e-3
        w0, w1, ws0, ws1 = MO.readout(Z.cpu().numpy())
        assert np.array_equal(m0.cpu().numpy(), w0) and np.array_equal(m1.cpu().numpy(), w1)
        assert np.abs(s0.cpu().numpy() - ws0).max() < 1e-5 and np.abs(s1.cpu().numpy() - ws1).max() < 1e-5
        assert (w0 >= 0).sum() >= min(m, n) // 2 - 2


@pytest.mark.parametrize('recompute', [True, False])
def test_sinkhorn_batch_with_widely_spread_scores(recompute):
    """Final descriptors whose scores spread over several hundred: whole columns underflow in the one-pass (linear-domain column sums)
    iteration of the stacked Sinkhorn; the exact log-domain fallback for those columns keeps it equal to the one-pair path.  Both ways
    of running the iterations: scores recomputed on the matrix cores in every pass (default), or the materialised matrix re-read."""
    from roreg_amd import hip
    rng = np.random.default_rng(17)
    m, n = 300, 260
    src = rng.standard_normal((m, 32)).astype(np.float32)
    tgt = rng.standard_normal((n, 32)).astype(np.float32)
    src[:, 0] = np.abs(src[:, 0]) + 1.0                      # every source has a positive first component ...
    tgt[::7, 0] = -150.0                                     # ... so these target columns score <= -150 against every source
    tgt[3::7] *= 6.0                                          # and these reach several tens
    s, t = cu(src), cu(tgt)
    Z, m0, m1, s0, s1 = hip.sinkhorn(s, t, 1.0, 100)
    seg_s = hip.Segments([m, m]); seg_t = hip.Segments([n, n])
    b0, b1, bs0, bs1 = hip.sinkhorn_batch(torch.cat([s, s]), torch.cat([t, t]), seg_s, seg_t, 1.0, 100, recompute=recompute)
    assert float(Z.max() - Z.min()) > 200 and bool(torch.isfinite(Z).all())
    for k in range(2):
        assert torch.equal(b0[k * m:(k + 1) * m], m0) and torch.equal(b1[k * n:(k + 1) * n], m1)
        assert float((bs0[k * m:(k + 1) * m] - s0).abs().max()) < 1e-4 and float((bs1[k * n:(k + 1) * n] - s1).abs().max()) < 1e-4


@pytest.mark.parametrize('iters', [100, 1, 0])
def test_sinkhorn_recomputed_on_the_matrix_cores_equals_the_materialised_iterations(iters):
    """The stacked Sinkhorn whose passes recompute <s_i, t_j> with fp16 hi + lo MFMAs (csrc/ot_flash.hip) against (a) the same call reading the
    materialised matrix, (b) the one-pair kernel, (c) the oracle's log-domain iteration: ragged pairs in ONE call, sizes on and around the
    32-row tile (3 x 5, 31 x 32, 32 x 31, 33 x 64 ...), descriptors from 0.02 to units (low pieces in fp16's subnormal range, scores up to
    ~100, where one float32 ulp of the score is already 1e-5 of the coupling), after 100, 1 and 0 iterations.  Matches identical, matching
    scores to 5e-5 (absolute; they are probabilities)."""
    from roreg_amd import hip
    rng = np.random.default_rng(23)
    sizes = [(3, 5), (31, 32), (32, 31), (33, 64), (200, 173), (64, 300), (1, 1), (97, 1)]
    S, T = [], []
    for q, (m, n) in enumerate(sizes):
        scale = [0.5, 0.02, 1.0, 0.05, 0.5, 0.7, 0.5, 0.5][q]
        s = rng.standard_normal((m, 32)).astype(np.float32) * scale; t = rng.standard_normal((n, 32)).astype(np.float32) * scale
        k = min(m, n) // 2
        t[:k] = s[:k] * 3                                     # planted strong matches
        S.append(s); T.append(t)
    seg_s = hip.Segments([m for m, _ in sizes]); seg_t = hip.Segments([n for _, n in sizes])
    cs, ct = cu(np.concatenate(S)), cu(np.concatenate(T))
    a0, a1, as0, as1 = hip.sinkhorn_batch(cs, ct, seg_s, seg_t, 1.5, iters, recompute=True)
    b0, b1, bs0, bs1 = hip.sinkhorn_batch(cs, ct, seg_s, seg_t, 1.5, iters, recompute=False)
    o0 = o1 = 0
    for q, ((m, n), s, t) in enumerate(zip(sizes, S, T)):
        Z, m0, m1, s0, s1 = hip.sinkhorn(cu(s), cu(t), 1.5, iters)                  # the literal two-pass log-domain kernel
        want = MO.log_sinkhorn((s @ t.T).astype(np.float32), np.float32(1.5), iters)
        assert np.abs(Z.cpu().numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max() / 20), (q, iters)
        w0, w1, ws0, ws1 = MO.readout(want)
        # After ONE iteration a column's two best rows can tie to the last bit (the oracle itself then disagrees with a float64 evaluation):
        # rows / columns are compared where the oracle's arg-max is decided by more than 1e-3 on both sides of the mutual check.
        P = want[:m, :n]
        gap = lambda A: (np.sort(A, 1)[:, -1] - np.sort(A, 1)[:, -2]) if A.shape[1] > 1 else np.full(A.shape[0], np.inf)
        row_ok, col_ok = gap(P) > 1e-3, gap(P.T) > 1e-3
        keep0 = row_ok & col_ok[P.argmax(1)]; keep1 = col_ok & row_ok[P.argmax(0)]
        for name, (g0, g1, gs0) in (('recomputed', (a0, a1, as0)), ('materialised', (b0, b1, bs0)), ('literal', (None, None, None))):
            h0 = (m0 if g0 is None else g0[o0:o0 + m]).cpu().numpy(); h1 = (m1 if g1 is None else g1[o1:o1 + n]).cpu().numpy()
            hs = (s0 if gs0 is None else gs0[o0:o0 + m]).cpu().numpy()
            assert np.array_equal(h0[keep0], w0[keep0]) and np.array_equal(h1[keep1], w1[keep1]), (name, q, (m, n), iters, h0.tolist(), w0.tolist())
            sure = keep0 & (h0 == w0) & np.isfinite(ws0)                         # (iters = 0: exp of raw scores up to 100 overflows to inf on both sides)
            assert (np.abs(hs[sure] - ws0[sure]) / np.maximum(1.0, np.abs(ws0[sure]))).max(initial=0.0) < 5e-5, (name, q, iters)   # (iters = 0: exp of raw scores, up to 1e11)
        o0 += m; o1 += n


@pytest.mark.parametrize('mode', [True, 'coop'])
def test_sinkhorn_whole_iteration_kernel_sizes_and_stacking(mode):
    """The forms of the recomputed iteration (csrc/ot_flash.hip) in ONE stacked call: of_iter_kernel (target clouds up to 2559 points, 1 ... 10
    column tiles per wave), and beyond that two passes per iteration (mode True, the default) or two cooperating workgroups per strip up to
    5119 points (mode 'coop'); source clouds on both sides of 2559 points (more than 80 strip sums per column); ragged pairs (tiles a pair
    does not have read the pad tile; every kernel skips the pairs of the other form).  (a) A pair's matches AND scores are bitwise the same
    stacked and alone -- alone the kernels are other instantiations (tiles per wave follow the call's longest target cloud) and the group
    needs other launches, so this is the independence of the stacking that the forms promise; (b) against the materialised iteration:
    matches identical where the arg-max is decided, scores to 5e-5; (c) a call with a target cloud above 5119 points takes the two-pass
    form in both modes and meets the same bar."""
    from roreg_amd import hip
    rng = np.random.default_rng(41)
    sizes = [(2500, 2500), (700, 1200), (1200, 700), (20, 300), (2500, 90), (3000, 1000), (5000, 2400), (1000, 3000), (2600, 5000), (300, 2600), (64, 5119)]
    S, T = [], []
    for m, n in sizes:
        s = rng.standard_normal((m, 32)).astype(np.float32) * 0.5; t = rng.standard_normal((n, 32)).astype(np.float32) * 0.5
        k = min(m, n) // 2
        t[:k] = s[:k] * 3 + rng.standard_normal((k, 32)).astype(np.float32) * 0.05
        S.append(s); T.append(t)

    def run(idx, recompute):
        recompute = mode if recompute else False
        seg_s = hip.Segments([sizes[q][0] for q in idx]); seg_t = hip.Segments([sizes[q][1] for q in idx])
        out = hip.sinkhorn_batch(cu(np.concatenate([S[q] for q in idx])), cu(np.concatenate([T[q] for q in idx])), seg_s, seg_t, 1.5, 100, recompute=recompute)
        return [x.cpu().numpy() for x in out], seg_s.host, seg_t.host

    (a0, a1, as0, as1), hs, ht = run(range(len(sizes)), True)
    (b0, b1, bs0, bs1), _, _ = run(range(len(sizes)), False)
    for q, (m, n) in enumerate(sizes):
        (c0, c1, cs0, cs1), _, _ = run([q], True)
        o0, o1 = hs[q], ht[q]
        assert np.array_equal(a0[o0:o0 + m], c0) and np.array_equal(a1[o1:o1 + n], c1), (q, 'matches stacked vs alone')
        assert np.array_equal(as0[o0:o0 + m], cs0) and np.array_equal(as1[o1:o1 + n], cs1), (q, 'scores stacked vs alone')
        same = a0[o0:o0 + m] == b0[o0:o0 + m]
        assert same.mean() > 0.995 and (a0[o0:o0 + m] >= 0).sum() >= min(m, n) // 2 - 5, (q, same.mean())          # (near-ties may fall either way between two arithmetics)
        assert np.abs(as0[o0:o0 + m][same] - bs0[o0:o0 + m][same]).max() < 5e-5, q
    # (c) one target cloud of 5200 points: the whole call runs of_pass_kernel twice per iteration
    s = rng.standard_normal((300, 32)).astype(np.float32) * 0.5; t = rng.standard_normal((5200, 32)).astype(np.float32) * 0.5
    t[:150] = s[:150] * 3
    seg_s = hip.Segments([300, sizes[1][0]]); seg_t = hip.Segments([5200, sizes[1][1]])
    cs, ct = cu(np.concatenate([s, S[1]])), cu(np.concatenate([t, T[1]]))
    d = [x.cpu().numpy() for x in hip.sinkhorn_batch(cs, ct, seg_s, seg_t, 1.5, 100, recompute=mode)]
    e = [x.cpu().numpy() for x in hip.sinkhorn_batch(cs, ct, seg_s, seg_t, 1.5, 100, recompute=False)]
    same = d[0] == e[0]
    assert same.mean() > 0.995 and (d[0][:300] >= 0).sum() >= 145 and np.abs(d[2][same] - e[2][same]).max() < 5e-5
    # the 700 x 1200 pair, whichever form ran it: same matches, scores to 2e-5
    m, n = sizes[1]
    assert (d[0][300:] == a0[hs[1]:hs[1] + m]).mean() > 0.995 and np.abs(d[2][300:] - as0[hs[1]:hs[1] + m])[d[0][300:] == a0[hs[1]:hs[1] + m]].max() < 2e-5


@pytest.mark.parametrize('mode', [True, 'coop'])
def test_sinkhorn_early_exit_of_converged_pairs_equals_all_iterations(mode):
    """Round 6 (include/roreg_hip.h, roreg_sinkhorn_early_exit): a pair's iterations stop once one of them moved none of its potentials by more
    than 2 .. 4 float32 units in the last place -- the float32 fixed point; the reference's loop (rot_coh_match.py:289-292) runs on to 100 and
    only flips last bits.  Stacked ragged pairs through every form of the iteration (whole-iteration kernel up to 2559 target points, two
    passes / cooperating workgroups beyond, odd and even stopping iterations for the double-buffered row potentials): matches identical to
    the all-iterations run, scores to 2e-5; the statistics show the saving; a pair's result is bitwise the same stacked and alone (the
    flags are per pair); with 1 or 0 iterations nothing changes; widely spread scores (slow movers) still agree."""
    from roreg_amd import hip
    rng = np.random.default_rng(61)
    sizes = [(2500, 2500), (700, 1200), (20, 300), (2500, 90), (3000, 1000), (1000, 3000), (2600, 5000), (64, 5119), (5000, 5000), (33, 64)]
    S, T = [], []
    for q, (m, n) in enumerate(sizes):
        sc = 0.5 if q == 4 else (0.25 if q % 3 else 0.2)              # (scores up to ~10 settle in 7 .. 40 iterations -- the matcher's own final descriptors do, at
                                                                      #  scores within [-18, 7] -- other scales stop elsewhere; pair 4, scores up to ~45, never settles)
        s = rng.standard_normal((m, 32)).astype(np.float32) * sc; t = rng.standard_normal((n, 32)).astype(np.float32) * sc
        k = min(m, n) // 2
        t[:k] = s[:k] * 3 + rng.standard_normal((k, 32)).astype(np.float32) * 0.05
        S.append(s); T.append(t)

    def run(idx, on, iters=100, alpha=1.5):
        seg_s = hip.Segments([sizes[q][0] for q in idx]); seg_t = hip.Segments([sizes[q][1] for q in idx])
        hip.sinkhorn_iteration_stats()
        with hip.sinkhorn_early_exit(on):
            out = hip.sinkhorn_batch(cu(np.concatenate([S[q] for q in idx])), cu(np.concatenate([T[q] for q in idx])), seg_s, seg_t, alpha, iters, recompute=mode)
        return [x.cpu().numpy() for x in out], seg_s.host, seg_t.host, hip.sinkhorn_iteration_stats()

    full, hs, ht, st_full = run(range(len(sizes)), False)
    early, _, _, st_early = run(range(len(sizes)), True)
    assert st_full == (100 * len(sizes), len(sizes))
    assert st_early[1] == len(sizes) and 100 + 2 * (len(sizes) - 1) <= st_early[0] < 100 + 50 * (len(sizes) - 1), st_early    # (pair 4 runs them all)
    print(f'[early exit, mode {mode}] iterations run: {st_early[0]} of {st_full[0]}')
    for k in range(2):
        assert np.array_equal(full[k], early[k]), k
    for k in (2, 3):
        assert np.abs(full[k] - early[k]).max() < 2e-5, k
    stops = set()
    for q in (1, 3, 4, 6, 8):                                         # stacked == alone, bit for bit, with the flags on
        one, _, _, st1 = run([q], True)
        m, n = sizes[q]
        assert np.array_equal(early[0][hs[q]:hs[q] + m], one[0]) and np.array_equal(early[1][ht[q]:ht[q] + n], one[1])
        assert np.array_equal(early[2][hs[q]:hs[q] + m], one[2]) and np.array_equal(early[3][ht[q]:ht[q] + n], one[3])
        assert (st1[0] == 100) if q == 4 else (2 <= st1[0] < 60), (q, st1)
        stops.add(st1[0])
    print(f'[early exit] iterations run by pairs 1, 3, 4, 6, 8 alone: {sorted(stops)}')
    for iters in (0, 1, 2, 3):                                        # nothing can stop before iteration 1 has been looked at
        a = run([1, 9], False, iters)[0]; b = run([1, 9], True, iters)[0]
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), iters
    # widely spread scores: columns that underflow, potentials of several hundred -- both settings agree to the usual bar
    m, n = 300, 260
    src = rng.standard_normal((m, 32)).astype(np.float32); tgt = rng.standard_normal((n, 32)).astype(np.float32)
    src[:, 0] = np.abs(src[:, 0]) + 1.0; tgt[::7, 0] = -150.0; tgt[3::7] *= 6.0
    seg_s = hip.Segments([m]); seg_t = hip.Segments([n])
    with hip.sinkhorn_early_exit(False):
        a = [x.cpu().numpy() for x in hip.sinkhorn_batch(cu(src), cu(tgt), seg_s, seg_t, 1.0, 100, recompute=mode)]
    with hip.sinkhorn_early_exit(True):
        b = [x.cpu().numpy() for x in hip.sinkhorn_batch(cu(src), cu(tgt), seg_s, seg_t, 1.0, 100, recompute=mode)]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.abs(a[2] - b[2]).max() < 1e-4 and np.abs(a[3] - b[3]).max() < 1e-4


@pytest.mark.parametrize('fvar,m,n', [(6, 2500, 2500), (6, 2500, 5000), (7, 3000, 4000), (7, 5000, 5000)])
def test_sinkhorn_stabilised_redo_of_a_strip_gives_the_same_result(tmp_path, fvar, m, n):
    """of_iter_kernel redoes a strip with the row maxima as stabilisers when a row's sum leaves (1e-35, 1e35) -- which finite, sanely scaled
    input never provokes.  ROREG_OT_FVAR=6 (read once per process: a child process) sends EVERY strip through that path: same matches,
    scores to 2e-5 against the normal path (n = 5000: the cooperating workgroups, three exchanges per iteration).
    ROREG_OT_FVAR=7: the two halves of a strip never see each other's words, i.e. every workgroup takes the bounded wait's fall-back
    (the partner's sums recomputed locally) -- BITWISE the normal path's matches and scores."""
    import subprocess, sys
    from roreg_amd import hip
    rng = np.random.default_rng(43)
    s = rng.standard_normal((m, 32)).astype(np.float32) * 0.5; t = rng.standard_normal((n, 32)).astype(np.float32) * 0.5
    t[:1000] = s[:1000] * 3 + rng.standard_normal((1000, 32)).astype(np.float32) * 0.05
    np.savez(tmp_path / 'in.npz', s=s, t=t)
    code = ("import numpy as np, torch, sys\n"
            "from roreg_amd import hip\n"
            "z = np.load(sys.argv[1]); s = torch.from_numpy(z['s']).cuda(); t = torch.from_numpy(z['t']).cuda()\n"
            "seg_s = hip.Segments([s.shape[0]]); seg_t = hip.Segments([t.shape[0]])\n"
            "a0, a1, as0, as1 = hip.sinkhorn_batch(s, t, seg_s, seg_t, 1.5, 100, recompute='coop')\n"
            "np.savez(sys.argv[2], a0=a0.cpu().numpy(), a1=a1.cpu().numpy(), as0=as0.cpu().numpy(), as1=as1.cpu().numpy())\n")
    env = dict(os.environ, ROREG_OT_FVAR=str(fvar), PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(tmp_path / 'out.npz')
    a0, a1, as0, as1 = [x.cpu().numpy() for x in hip.sinkhorn_batch(cu(s), cu(t), hip.Segments([m]), hip.Segments([n]), 1.5, 100, recompute='coop')]
    if fvar == 7:
        assert np.array_equal(z['a0'], a0) and np.array_equal(z['a1'], a1) and np.array_equal(z['as0'], as0) and np.array_equal(z['as1'], as1)
    same = z['a0'] == a0
    assert same.mean() > 0.998 and (a0 >= 0).sum() >= 990, same.mean()
    assert np.abs(z['as0'][same] - as0[same]).max() < 2e-5 and np.abs(z['as1'][z['a1'] == a1] - as1[z['a1'] == a1]).max() < 2e-5
    if n > 2559:                                                   # and the two forms a long target cloud can take agree like two arithmetics do
        b0, b1, bs0, bs1 = [x.cpu().numpy() for x in hip.sinkhorn_batch(cu(s), cu(t), hip.Segments([m]), hip.Segments([n]), 1.5, 100, recompute=True)]
        same = a0 == b0
        assert same.mean() > 0.998 and np.abs(as0[same] - bs0[same]).max() < 2e-5


def test_sinkhorn_readout_without_the_matrix_is_bitwise_the_matrix_readout(tmp_path):
    """The read-out of the recomputed path comes straight from the descriptors (ot_argmax_mfma_kernel: Z0 as float32 MFMA chains = ot_build_kernel's
    fmaf chains, packed (value, ~index) keys).  Against the same call with the two (m+1) x (n+1) matrices built and scanned
    (ROREG_OT_READOUT_MFMA=0, a child process): matches0 / matches1 / both score vectors BITWISE, on ragged stacked pairs with DUPLICATED
    descriptors on both sides (exact ties: the first index must win in rows and in columns) and a pair with fewer points than a tile."""
    import subprocess, sys
    from roreg_amd import hip
    rng = np.random.default_rng(77)
    sizes = [(700, 650), (33, 5000), (5000, 97), (20, 20), (2500, 2500)]
    S, T = [], []
    for m, n in sizes:
        s = rng.standard_normal((m, 32)).astype(np.float32) * 0.5; t = rng.standard_normal((n, 32)).astype(np.float32) * 0.5
        k = min(m, n) // 2
        t[:k] = s[:k] * 3
        if m > 40 and n > 40:
            t[n - 10:] = t[:10]; s[m - 7:] = s[3:10]                  # duplicated targets and sources: exactly tied scores
        S.append(s); T.append(t)
    np.savez(tmp_path / 'in.npz', s=np.concatenate(S), t=np.concatenate(T), m=np.array([a for a, _ in sizes]), n=np.array([b for _, b in sizes]))
    code = ("import numpy as np, torch, sys\n"
            "from roreg_amd import hip\n"
            "z = np.load(sys.argv[1]); s = torch.from_numpy(z['s']).cuda(); t = torch.from_numpy(z['t']).cuda()\n"
            "out = hip.sinkhorn_batch(s, t, hip.Segments(z['m']), hip.Segments(z['n']), 1.5, 100, recompute=True)\n"
            "np.savez(sys.argv[2], *[x.cpu().numpy() for x in out])\n")
    env = dict(os.environ, ROREG_OT_READOUT_MFMA='0', PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(tmp_path / 'out.npz')
    got = hip.sinkhorn_batch(cu(np.concatenate(S)), cu(np.concatenate(T)), hip.Segments([a for a, _ in sizes]), hip.Segments([b for _, b in sizes]), 1.5, 100, recompute=True)
    for q, g in enumerate(got):
        assert np.array_equal(g.cpu().numpy(), z[f'arr_{q}']), q
    assert int((got[0] >= 0).sum()) > 1000


def test_sinkhorn_recomputed_survives_non_finite_and_huge_descriptors():
    """No fault and no hang on NaN / inf / 1e6-sized descriptors (the fp16 operands overflow: those pairs' results are meaningless, as the
    reference's would be); a healthy pair stacked beside them is unaffected."""
    from roreg_amd import hip
    rng = np.random.default_rng(5)
    good_s = rng.standard_normal((70, 32)).astype(np.float32); good_t = rng.standard_normal((90, 32)).astype(np.float32)
    good_t[:30] = good_s[:30] * 2
    bad_s = good_s.copy(); bad_s[3, 5] = np.nan; bad_s[9, 1] = np.inf
    huge_s = good_s * 1e6
    seg_s = hip.Segments([70, 70, 70]); seg_t = hip.Segments([90, 90, 90])
    cs = cu(np.concatenate([good_s, bad_s, huge_s])); ct = cu(np.concatenate([good_t, good_t, good_t]))
    a0, a1, as0, as1 = hip.sinkhorn_batch(cs, ct, seg_s, seg_t, 1.0, 100, recompute=True)
    torch.cuda.synchronize()
    Z, m0, m1, s0, s1 = hip.sinkhorn(cu(good_s), cu(good_t), 1.0, 100)
    assert torch.equal(a0[:70], m0) and torch.equal(a1[:90], m1) and float((as0[:70] - s0).abs().max()) < 2e-5
    assert int(a0.min()) >= -1 and int(a0.max()) < 90


def test_match_ot_forward_vs_reference_golden(rm):
    net, sd = rm
    z = load_golden('match_ot')
    batch = {k: torch.from_numpy(z[k]) for k in ['feats0', 'feats1', 'keys0', 'keys1']}
    with torch.no_grad():
        out = net(batch)
    # 1e-4 on every floating-point output (measured: 2.5e-6 on the final descriptors, 1.5e-5 on the log-couplings of
    # magnitude up to 17, 2e-7 on the matching scores; the reference's own float32-vs-float64 noise on this case is 2e-6 / 1e-5 / 1e-7,
    # tests/golden/match_ot_noise.json)
    assert np.abs(out['source_final'].cpu().numpy() - z['out_source_final']).max() < 1e-4
    assert np.abs(out['target_final'].cpu().numpy() - z['out_target_final']).max() < 1e-4
    assert np.abs(out['scores'].cpu().numpy() - z['out_scores']).max() < 1e-4
    assert np.array_equal(out['matches0'].cpu().numpy(), z['out_matches0'])
    assert np.array_equal(out['matches1'].cpu().numpy(), z['out_matches1'])
    assert np.abs(out['matching_scores0'].cpu().numpy() - z['out_matching_scores0']).max() < 1e-4
    assert np.abs(out['matching_scores1'].cpu().numpy() - z['out_matching_scores1']).max() < 1e-4
    assert out['scores'].shape == z['out_scores'].shape and out['matches0'].dtype == torch.int64
    assert np.abs(out['scores_other'].cpu().numpy() - z['out_scores_other']).max() < 1e-4


@pytest.mark.parametrize('mfma_layers', [False, True])
def test_match_ot_stacked_pairs_equal_the_per_pair_forward(rm, mfma_layers):
    """Several ragged pairs through ONE pass of the network (segmented neighbour search, InstanceNorm statistics, context maximum,
    Sinkhorn) against forward() pair by pair.  Default: the stacked path runs forward()'s kernels -- matches AND scores BITWISE the
    per-pair forward()'s (one arithmetic behind both).  mfma_layers (ROREG_LINEAR_MFMA=1, opt-in): its 1x1 layers and R_indicator on the
    matrix cores -- float32-accurate, another rounding: identical matches on these pairs, scores to 2e-5.  The golden pair is one of them
    and is checked against the reference directly; a pair's result does not depend on the pairs stacked beside it."""
    net, sd = rm
    object.__setattr__(net, 'matrix_core_layers', mfma_layers)
    z = load_golden('match_ot')
    rng = np.random.default_rng(11)
    pairs = [(cu(z['feats0'][0]), cu(z['feats1'][0]), cu(z['keys0'][0]), cu(z['keys1'][0]))]
    for m, n in [(300, 257), (64, 190), (513, 512)]:
        f0 = rng.standard_normal((m, 32, 60)).astype(np.float32); f0 /= np.linalg.norm(f0, axis=1, keepdims=True)
        perm = rng.permutation(m)[:n] if n <= m else rng.integers(0, m, n)
        f1 = (f0[perm] + 0.05 * rng.standard_normal((n, 32, 60))).astype(np.float32)
        k0 = rng.uniform(0, 3, (m, 3)).astype(np.float32); k1 = (k0[perm] + 0.01 * rng.standard_normal((n, 3))).astype(np.float32)
        pairs.append((cu(f0), cu(f1), cu(k0), cu(k1)))
    with torch.no_grad():
        got = net.match_many(pairs)
        for (f0, f1, k0, k1), (m0, s0) in zip(pairs, got):
            want = net({'feats0': f0[None], 'feats1': f1[None], 'keys0': k0[None], 'keys1': k1[None]})
            assert torch.equal(m0, want['matches0'][0])
            if mfma_layers:
                assert (s0 - want['matching_scores0'][0]).abs().max() < 2e-5
            else:
                assert torch.equal(s0, want['matching_scores0'][0])
    assert np.array_equal(got[0][0].cpu().numpy(), z['out_matches0'][0])
    assert np.abs(got[0][1].cpu().numpy() - z['out_matching_scores0'][0]).max() < 1e-4
    assert int((got[1][0] >= 0).sum()) > 20                      # the synthetic pairs do produce matches
    with torch.no_grad():
        alone = net.match_many(pairs[2:3]); reordered = net.match_many([pairs[3], pairs[2], pairs[0]])
    assert torch.equal(alone[0][0], got[2][0]) and torch.equal(alone[0][1], got[2][1])                  # bitwise, whatever the stacking
    assert torch.equal(reordered[1][0], got[2][0]) and torch.equal(reordered[1][1], got[2][1]) and torch.equal(reordered[2][1], got[0][1])
    object.__setattr__(net, 'matrix_core_layers', None)


def test_stage_yoho_mat_and_yohoo_with_rm_scores(tmp_path):
    """--RD --RM --ET yohoo pipeline golden: matcher output (matches + float32 scores) and the RM branch of yohoo_ransac."""
    from test_hip_pipeline import _setup, _put_yoho
    from roreg_amd.test import name2matcher, name2estimator
    z = load_golden('pipeline_rd_rm_yohoo')
    y = load_golden('pipeline_mutual_yohoo')
    cfg, ds = _setup(tmp_path, z, RD=True, RM=True, ET='yohoo')
    keynum = int(z['keynum'])
    base = f'{cfg.output_cache_fn}/{ds.name}'
    _put_yoho(cfg, ds, y)
    os.makedirs(f'{base}/det_score')
    for pc in ds.pc_ids:
        np.save(f'{base}/det_score/{pc}.npy', z[f'det_{pc}'])
    np.random.seed(1234)
    name2matcher['yoho_mat'](cfg).run(ds, keynum)
    md = f'{base}/match_{keynum}'
    for a, b in ds.pair_ids:
        m = np.load(f'{md}/{a}-{b}.npy'); s = np.load(f'{md}/scores/{a}-{b}.npy')
        assert m.dtype == np.int64 and np.array_equal(m, z[f'match_{a}_{b}'])
        assert s.dtype == np.float32 and np.abs(s - z[f'mscore_{a}_{b}']).max() < 1e-4
        np.save(f'{md}/scores/{a}-{b}.npy', z[f'mscore_{a}_{b}'])
    est = name2estimator['yohoo'](cfg)
    est.rind_extractor.Rindex(ds, keynum)
    for a, b in ds.pair_ids:
        assert np.array_equal(np.load(f'{md}/DR_index/{a}-{b}.npy'), z[f'dr_{a}_{b}'])
    os.makedirs(f'{md}/Trans_pre', exist_ok=True)
    for a, b in ds.pair_ids:
        np.save(f'{md}/Trans_pre/{a}-{b}.npy', z[f'transpre_{a}_{b}'])
    np.random.seed(4321)
    # the golden run seeded before the whole estimator stage; Rindex / Rt_pre draw nothing, so the stream is the same here
    est.ransacer.ransac(ds, keynum, 1000)
    for a, b in ds.pair_ids:
        r = np.load(f'{md}/yohoo/1000iters/{a}-{b}.npz')
        assert int(r['recalltime']) == int(z[f'recall_{a}_{b}'])
        want = z[f'trans_{a}_{b}']
        if np.isfinite(want).all():
            assert np.abs(r['trans'] - want).max() < 1e-5
