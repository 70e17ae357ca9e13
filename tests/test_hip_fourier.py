"""Irrep-domain (group-Fourier) evaluation of the group conv against the direct MFMA group conv, the oracle and the golden.  GPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_numpy as O
from roreg_amd import synth
from roreg_amd.parses.parses_test import default_config

pytestmark = pytest.mark.gpu


def test_transform_roundtrip_and_orthogonality():
    from roreg_amd import hip
    from roreg_amd.fourier import group_fourier
    gf = group_fourier()
    rng = np.random.default_rng(0)
    for B, C in [(37, 32), (200, 64)]:
        x = rng.standard_normal((B, C, 60)).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        coef = hip.ft_nonlin(B, C, x_spatial=xd)
        # coefficient (rho,i,l) of (b,c) sits at X_rho[(l*C + c), (b/32)*32*d + i*32 + b%32]; pad keypoints are zero
        want = x @ gf.F.T.astype(np.float32)                                   # [B,C,60(q)]
        views = hip.coef_views(coef, C, B)
        Bp = hip.coef_pitch(B)
        for q, (ri, i, l) in enumerate(gf.index):
            d = hip.IRREP_DIMS[ri]
            got = views[ri].view(d, C, Bp // 32, d, 32)[l, :, :, i, :].reshape(C, Bp).t().cpu().numpy()
            assert np.abs(got[:B] - want[:, :, q]).max() < 1e-5
            assert np.abs(got[B:]).max() == 0 if Bp > B else True
        back = hip.ft_nonlin(B, C, coef_in=coef, spatial_out=True).cpu().numpy()
        assert np.abs(back - x).max() < 1e-5


def test_fourier_layer_equals_direct_group_conv(group):
    """FT -> per-irrep GEMMs -> IFT reproduces the 13-stencil group conv (incl. bias)."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    rng = np.random.default_rng(1)
    B, C, Oc = 72, 64, 160
    conv = torch.nn.Conv2d(C, Oc, (1, 13))
    x = rng.standard_normal((B, C, 60)).astype(np.float32)
    want = O.group_conv(x, conv.weight.detach().numpy(), conv.bias.detach().numpy(), group.Nei)
    L = _Layer(conv)
    xd = torch.from_numpy(x).cuda()
    X = hip.ft_nonlin(B, C, x_spatial=xd)
    T = hip.irrep_gemm(X, L.wpack, C, Oc, B)
    got = hip.ft_nonlin(B, Oc, coef_in=T, bias=L.bias, spatial_out=True).cpu().numpy()
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def test_ft_nonlin_split_matches_f32_path(group):
    """The 3 x bf16 split transforms agree with the exact-f32 MFMA transforms at the f32 rounding level, for all three kernel
    variants (group -> irrep, irrep -> irrep with bias/BN/ReLU, irrep -> group with residual and a live-column map)."""
    from roreg_amd import hip
    rng = np.random.default_rng(8)
    B, C = 77, 48
    x = torch.from_numpy((rng.standard_normal((B, C, 60)) * np.exp(rng.standard_normal((B, C, 1)))).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).cuda()
    bn = (torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).cuda(), torch.from_numpy(rng.standard_normal(C).astype(np.float32)).cuda())
    X32 = hip.ft_nonlin(B, C, x_spatial=x); Xsp = hip.ft_nonlin(B, C, x_spatial=x, split=True)
    scale = float(X32.abs().max())
    assert float((X32 - Xsp).abs().max()) < 4e-7 * scale
    Y32 = hip.ft_nonlin(B, C, coef_in=X32, bias=bias, bn=bn); Ysp = hip.ft_nonlin(B, C, coef_in=X32, bias=bias, bn=bn, split=True)
    assert float((Y32 - Ysp).abs().max()) < 1e-6 * float(Y32.abs().max())
    gmap = np.full(60, -1, np.int32); live = rng.permutation(60)[:45]; gmap[live] = np.arange(45)
    gm = torch.from_numpy(gmap).cuda()
    Z32 = hip.ft_nonlin(B, C, coef_in=X32, bias=bias, spatial_out=True, g_map=gm, Lout=48, Lvalid=45)
    Zsp = hip.ft_nonlin(B, C, coef_in=X32, bias=bias, spatial_out=True, g_map=gm, Lout=48, Lvalid=45, split=True)
    assert float((Z32 - Zsp).abs().max()) < 1e-6 * float(Z32.abs().max())
    back = hip.ft_nonlin(B, C, coef_in=Xsp, resid_spatial=x, spatial_out=True, split=True)      # F^-1 F x + x = 2x
    assert float((back - 2 * x).abs().max()) < 2e-6 * float(x.abs().max())
    # fp16 x 2 with per-keypoint-column scales; coefficient outputs leave as fp16 hi/lo words under the keypoint's bound
    b0 = hip.row_bound(x)
    assert bool((b0[:B] >= hip.pack_coefs_f16x2(X32, C, B)[1][:B]).all()) and float(b0[B:].abs().max() if b0.numel() > B else 0) == 0
    X16 = hip.unpack_coefs_f16x2(hip.ft_nonlin(B, C, x_spatial=x, split='f16x2', out_bound=b0), b0, C, B)
    assert float((X32 - X16).abs().max()) < 6e-7 * scale
    yb = 1.01 * hip.pack_coefs_f16x2(Y32, C, B)[1]
    Y16 = hip.unpack_coefs_f16x2(hip.ft_nonlin(B, C, coef_in=X32, bias=bias, bn=bn, split='f16x2', out_bound=yb), yb, C, B)
    assert float((Y32 - Y16).abs().max()) < 1.5e-6 * float(Y32.abs().max())
    Z16, zmax = hip.ft_nonlin(B, C, coef_in=X32, bias=bias, spatial_out=True, g_map=gm, Lout=48, Lvalid=45, split='f16x2', want_rowmax=True)
    assert float((Z32 - Z16).abs().max()) < 1.5e-6 * float(Z32.abs().max())
    assert torch.equal(zmax, Z16.abs().amax((1, 2)))                      # the tracked per-keypoint scale is the exact maximum of what was written


def test_group_conv_split_is_f32_accurate(group):
    """Pruned 13-stencil conv (45 live columns -> 13) with BN+ReLU: the 3 x bf16 split kernel against float64 and the f32-MFMA kernel."""
    from roreg_amd import hip
    rng = np.random.default_rng(12)
    B, C, Oc, Lin, Lout = 37, 64, 256, 48, 13
    conv = torch.nn.Conv2d(C, Oc, (1, 13))
    bn = (torch.rand(C) + 0.5, torch.randn(C) * 0.1, torch.randn(C) * 0.1, torch.rand(C) + 0.5)
    layer = hip.ConvLayer(conv.weight, conv.bias, bn)
    x = (rng.standard_normal((B, C, Lin)) * np.exp(rng.standard_normal((B, C, 1)))).astype(np.float32)
    x[:, :, 45:] = 0
    gather = np.stack([rng.permutation(45)[:13] for _ in range(Lout)]).astype(np.int32)
    xd = torch.from_numpy(x).cuda(); gd = torch.from_numpy(gather).cuda()
    y32 = hip.group_conv(xd, layer, gather=gd).double().cpu().numpy()
    ysp = hip.group_conv(xd, layer, gather=gd, split=True).double().cpu().numpy()
    g, b_, m, v = [t.double().numpy() for t in bn]
    scale = g / np.sqrt(v + 1e-5); shift = b_ - m * scale
    a = np.maximum(x.astype(np.float64) * scale[None, :, None] + shift[None, :, None], 0.0)
    ref = np.einsum('ock,bcjk->boj', conv.weight.detach().double().numpy()[:, :, 0, :], a[:, :, gather]) + conv.bias.detach().double().numpy()[None, :, None]
    s = np.abs(ref).max()
    amax = xd.abs().amax((1, 2))
    y16t, omax = hip.group_conv(xd, layer, gather=gd, in_rowmax=amax, want_rowmax=True)
    assert torch.equal(omax, y16t.abs().amax((1, 2)))                      # the kernel tracks the exact per-row maximum of what it wrote
    sub = slice(5, 19)                                                     # a row's result does not depend on the other rows of the launch
    assert torch.equal(hip.group_conv(xd[sub].contiguous(), layer, gather=gd, in_rowmax=amax[sub].contiguous()), y16t[sub])
    y16 = y16t.double().cpu().numpy()
    e32 = np.abs(y32 - ref).max() / s; esp = np.abs(ysp - ref).max() / s; e16 = np.abs(y16 - ref).max() / s
    assert e32 < 2e-6 and esp < 2e-6 and e16 < 2e-6, (e32, esp, e16)
    assert esp < 3 * e32 + 2e-7 and e16 < 3 * e32 + 2e-7, (e32, esp, e16)
    # the LDS slot order of the input columns (hip.group_conv lds_order, tools/lds_perm_search.py) is an execution hint: any injective
    # order at any stride gives bitwise the same output, in both split modes, also on a launch of several column tiles
    big = torch.from_numpy((rng.standard_normal((700, C, Lin)) * np.exp(rng.standard_normal((700, C, 1)))).astype(np.float32)).cuda()
    bmax = big.abs().amax((1, 2))
    want_sp = hip.group_conv(big, layer, gather=gd, split=True); want_16 = hip.group_conv(big, layer, gather=gd, in_rowmax=bmax)
    for stride in (45, 48, 53, 64):
        order = np.full(Lin, -1, np.int32); order[:45] = rng.permutation(stride)[:45]
        od = (torch.from_numpy(order).cuda(), stride)
        assert torch.equal(hip.group_conv(big, layer, gather=gd, split=True, lds_order=od), want_sp), stride
        assert torch.equal(hip.group_conv(big, layer, gather=gd, in_rowmax=bmax, lds_order=od), want_16), stride


def test_dense_split_is_f32_accurate():
    """Dense layer on row-major activations (ET trunk tail / head): ragged B and O, with and without BN+ReLU / residual, against float64."""
    from roreg_amd import hip
    rng = np.random.default_rng(13)
    for B, K, Oc, act, res in [(300, 6656, 256, True, True), (77, 256, 512, False, False), (130, 128, 4, True, False), (1, 512, 128, True, True)]:
        W = (rng.standard_normal((Oc, K)) / np.sqrt(K)).astype(np.float32); bias = rng.standard_normal(Oc).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, K).astype(np.float32) if act else None; sh = rng.standard_normal(K).astype(np.float32) if act else None
        x = (rng.standard_normal((B, K)) * np.exp(rng.standard_normal((B, 1)))).astype(np.float32)
        r = rng.standard_normal((B, Oc)).astype(np.float32) if res else None
        layer = hip.DenseSplitLayer(W, bias, sc, sh)
        xd = torch.from_numpy(x).cuda(); rd = torch.from_numpy(r).cuda() if res else None
        got = hip.dense_split(xd, layer, residual=rd).double().cpu().numpy()
        rmax = xd.abs().amax(1)
        g16, omax = hip.dense_split(xd, layer, residual=rd, in_rowmax=rmax, want_rowmax=True)
        assert torch.equal(omax, g16.abs().amax(1))
        if B > 40:                                                          # batch-composition independence of a row's result
            sub = slice(17, 40)
            assert torch.equal(hip.dense_split(xd[sub].contiguous(), layer, residual=rd[sub].contiguous() if res else None, in_rowmax=rmax[sub].contiguous()), g16[sub])
        a = x.astype(np.float64)
        if act:
            a = np.maximum(a * sc.astype(np.float64) + sh.astype(np.float64), 0.0)
        ref = a @ W.astype(np.float64).T + bias.astype(np.float64) + (r.astype(np.float64) if res else 0.0)
        tol = 3e-6 * max(1.0, np.abs(ref).max())
        assert np.abs(got - ref).max() < tol and np.abs(g16.double().cpu().numpy() - ref).max() < tol, (B, K, Oc)


def test_gemm_epilogue_residual_is_exact(group):
    """Out = W.X + Add in the GEMM epilogue is bitwise the separately computed sum (both GEMM kernels)."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    rng = np.random.default_rng(7)
    B, C, Oc = 100, 64, 96
    L = _Layer(torch.nn.Conv2d(C, Oc, (1, 13)))
    X = torch.from_numpy(rng.standard_normal(hip.coef_size(C, B)).astype(np.float32)).cuda()
    A = torch.from_numpy(rng.standard_normal(hip.coef_size(Oc, B)).astype(np.float32)).cuda()
    for sp in (None, L.wsplit):
        plain = hip.irrep_gemm(X, L.wpack, C, Oc, B, split=sp)
        fused = hip.irrep_gemm(X, L.wpack, C, Oc, B, split=sp, add=A)
        assert torch.equal(fused, plain + A)
    Xp, xb = hip.pack_coefs_f16x2(X, C, B)
    plain = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb)
    assert torch.equal(hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb, add=A), plain + A)


def test_gf_fourier_vs_direct_vs_golden(group):
    from roreg_amd.network import name2network
    z = load_golden('gf_forward')
    net = name2network['GF_test'](default_config())
    sd = synth.seeded_state_dict(net, int(z['seed']))
    x = torch.from_numpy(z['x'])
    net.PartI_net.mode = 'fourier'
    f = net(x)
    net.PartI_net.mode = 'direct'
    d = net(x)
    assert np.abs(f['eqv'].cpu().numpy() - d['eqv'].cpu().numpy()).max() < 1e-5
    assert np.abs(f['eqv'].cpu().numpy() - z['eqv']).max() < 1e-5
    assert np.abs(f['inv'].cpu().numpy() - z['inv']).max() < 1e-5
    # ragged batch sizes (not multiples of 4 / 32 / 256)
    rng = np.random.default_rng(2)
    for B in [1, 5, 257, 1250]:
        xb = torch.from_numpy(rng.standard_normal((B, 32, 60)).astype(np.float32))
        net.PartI_net.mode = 'fourier'; a = net(xb)['eqv'].cpu().numpy()
        net.PartI_net.mode = 'direct'; b = net(xb)['eqv'].cpu().numpy()
        assert np.abs(a - b).max() < 1e-5, B


def test_split_bf16_gemm_is_f32_accurate(group):
    """3 x bf16 split GEMM (six cross products, f32 accumulate) against float64: error at the level of the exact-f32 MFMA kernel."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    rng = np.random.default_rng(4)
    B, C, Oc = 64, 256, 512
    conv = torch.nn.Conv2d(C, Oc, (1, 13))
    L = _Layer(conv)
    x = rng.standard_normal((B, C, 60)).astype(np.float32) * np.abs(rng.standard_normal((B, C, 1))).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    X = hip.ft_nonlin(B, C, x_spatial=xd)
    xb = hip.row_bound(xd)
    Xp = hip.ft_nonlin(B, C, x_spatial=xd, split='f16x2', out_bound=xb)
    T32 = hip.irrep_gemm(X, L.wpack, C, Oc, B)
    Tsp = hip.irrep_gemm(X, L.wpack, C, Oc, B, split=L.wsplit)
    T16 = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb)
    y32 = hip.ft_nonlin(B, Oc, coef_in=T32, bias=L.bias, spatial_out=True).double().cpu().numpy()
    ysp = hip.ft_nonlin(B, Oc, coef_in=Tsp, bias=L.bias, spatial_out=True).double().cpu().numpy()
    y16 = hip.ft_nonlin(B, Oc, coef_in=T16, bias=L.bias, spatial_out=True).double().cpu().numpy()
    ref = np.einsum('ock,bcgk->bog', conv.weight.detach().double().numpy()[:, :, 0, :], x.astype(np.float64)[:, :, group.Nei]) + \
        conv.bias.detach().double().numpy()[None, :, None]
    scale = np.abs(ref).max()
    e32 = np.abs(y32 - ref).max() / scale; esp = np.abs(ysp - ref).max() / scale; e16 = np.abs(y16 - ref).max() / scale
    assert e32 < 2e-6 and esp < 2e-6 and e16 < 2e-6, (e32, esp, e16)
    assert esp < 3 * e32 + 1e-7 and e16 < 3 * e32 + 1e-7, (e32, esp, e16)


@pytest.mark.parametrize('span_bits', [12, 8])
def test_f16x2_small_channels_beside_large_ones_keep_their_own_accuracy(group, span_bits):
    """PER-ELEMENT accuracy under the per-keypoint block scale (the other accuracy tests take a maximum over the whole output tensor): in
    every keypoint half of the input channels are 2^span times smaller than the other half, and the weights are block diagonal, so half of
    the OUTPUT channels are 2^span times smaller than the others and depend on the small inputs alone.  Their errors, measured against
    float64 relative to THEIR OWN magnitude, must stay at the f32-input kernel's level (<= 3 x): the fp16 hi + lo split is relative to the
    keypoint's bound, which the large channels set."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    rng = np.random.default_rng(40 + span_bits)
    B, C, Oc = 96, 256, 512
    conv = torch.nn.Conv2d(C, Oc, (1, 13))
    with torch.no_grad():
        conv.weight[:Oc // 2, C // 2:] = 0.0; conv.weight[Oc // 2:, :C // 2] = 0.0; conv.bias.zero_()
    L = _Layer(conv)
    x = rng.standard_normal((B, C, 60)).astype(np.float32)
    x[:, :C // 2] *= np.float32(2.0 ** -span_bits)                                   # the small half, in the SAME keypoints as the large half
    xd = torch.from_numpy(x).cuda()
    X = hip.ft_nonlin(B, C, x_spatial=xd)
    xb = hip.row_bound(xd)
    Xp = hip.ft_nonlin(B, C, x_spatial=xd, split='f16x2', out_bound=xb)
    T32 = hip.irrep_gemm(X, L.wpack, C, Oc, B)
    Tsp = hip.irrep_gemm(X, L.wpack, C, Oc, B, split=L.wsplit)
    T16 = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb)
    ref = np.einsum('ock,bcgk->bog', conv.weight.detach().double().numpy()[:, :, 0, :], x.astype(np.float64)[:, :, group.Nei])
    small = ref[:, :Oc // 2]
    assert np.abs(small).max() < 2.0 ** (3 - span_bits) * np.abs(ref[:, Oc // 2:]).max()
    err = {}
    for name, T in (('f32', T32), ('bf16x3', Tsp), ('f16x2', T16)):
        y = hip.ft_nonlin(B, Oc, coef_in=T, bias=L.bias, spatial_out=True).double().cpu().numpy()
        err[name] = (float(np.abs(y[:, :Oc // 2] - small).max() / np.abs(small).max()), float(np.abs(y[:, Oc // 2:] - ref[:, Oc // 2:]).max() / np.abs(ref[:, Oc // 2:]).max()))
    print(f'span 2^{span_bits}: (small-channel, large-channel) errors relative to their own scale: {err}')
    assert err['f32'][0] < 4e-6, err
    assert err['f16x2'][0] < 3 * err['f32'][0] + 1e-7 and err['f16x2'][1] < 3 * err['f32'][1] + 1e-7, err
    assert err['bf16x3'][0] < 3 * err['f32'][0] + 1e-7, err


def test_f16x2_block_scale_survives_outliers(group):
    """One coefficient 10^4 times larger than the rest sets the block scale of ITS keypoint in the fp16 x 2 GEMM; the keypoint's other columns
    must keep their accuracy (the hi/lo split keeps 22 bits within ~11 binades of the bound), and other keypoints are not affected at all."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    rng = np.random.default_rng(3)
    B, C, Oc = 256, 256, 512
    L = _Layer(torch.nn.Conv2d(C, Oc, (1, 13)))
    X = torch.from_numpy(rng.standard_normal(hip.coef_size(C, B)).astype(np.float32)).cuda()
    v = hip.coef_views(X, C, B)
    v[4][:, 7] *= 1e4
    Xp, xb = hip.pack_coefs_f16x2(X, C, B)
    T16 = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb)
    T32 = hip.irrep_gemm(X, L.wpack, C, Oc, B)
    ref = L.dense[4].astype(np.float64) @ v[4].double().cpu().numpy()
    cols = [c for c in range(600) if c != 7]
    scale = np.abs(ref[:, cols]).max(0)
    err = {}
    for name, T in (('f16', T16), ('f32', T32)):
        g = hip.coef_views(T, Oc, B)[4].double().cpu().numpy()[:ref.shape[0]]
        err[name] = float((np.abs(g[:, cols] - ref[:, cols]).max(0) / scale).max())
    assert err['f16'] < 3 * err['f32'] + 1e-7 and err['f16'] < 4e-6, err


@pytest.mark.parametrize('C,Oc', [(256, 512), (512, 256)])
@pytest.mark.parametrize('B', [40, 256, 1000])
def test_plane_layout_gemm_is_bitwise_the_word_layout_gemm(group, C, Oc, B):      # (+ the 16x16x32 kernel against both, to rounding)
    """ft_nonlin(planes=True) writes exactly the bits of the word layout, re-arranged into hi / lo planes, and the GEMM fed from the planes
    by LDS-DMA (irrep_gemm_xdma_kernel) returns bit for bit what the word-layout kernel returns -- coefficients, the residual add and the
    propagated bound -- for a batch below one column tile, an exact multiple, and a ragged one."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    rng = np.random.default_rng(100 * B + C)
    torch.manual_seed(C)
    L = _Layer(torch.nn.Conv2d(C, Oc, (1, 13)))
    x = torch.from_numpy((rng.standard_normal((B, C, 60)) * np.exp(rng.standard_normal((B, C, 1)))).astype(np.float32)).cuda()
    T = hip.ft_nonlin(B, C, x_spatial=x)                                             # float32 coefficients, pad keypoints zero
    bias = torch.from_numpy(rng.standard_normal(C).astype(np.float32)).cuda()
    bn = (torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).cuda(), torch.from_numpy(rng.standard_normal(C).astype(np.float32)).cuda())
    Y = hip.ft_nonlin(B, C, coef_in=T, bias=bias, bn=bn, split=True)
    yb = 1.01 * hip.pack_coefs_f16x2(Y, C, B)[1]
    Xw = hip.ft_nonlin(B, C, coef_in=T, bias=bias, bn=bn, split='f16x2', out_bound=yb)
    Xp = hip.ft_nonlin(B, C, coef_in=T, bias=bias, bn=bn, split='f16x2', out_bound=yb, planes=True)
    assert torch.equal(Xp.view(torch.int32), hip.words_to_planes(Xw, C, B).view(torch.int32))
    bn2 = (torch.from_numpy(rng.uniform(0.5, 1.5, Oc).astype(np.float32)).cuda(), torch.from_numpy(rng.standard_normal(Oc).astype(np.float32)).cuda())
    nb = hip.next_bound(bn2, L.bias)
    add = torch.from_numpy(rng.standard_normal(hip.coef_size(Oc, B)).astype(np.float32)).cuda()
    for kw in ({}, {'add': add}):
        Tw, bw = hip.irrep_gemm(Xw, None, C, Oc, B, f16x2=L.wsplit2, x_bound=yb, next_bound=nb, **kw)
        Tp, bp = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=yb, next_bound=nb, x_planes=1, **kw)      # 32x32x16 MFMAs: the same sums in the same order
        assert torch.equal(Tw.view(torch.int32), Tp.view(torch.int32)) and torch.equal(bw.view(torch.int32), bp.view(torch.int32))
        # the 16x16x32 kernel (the default): 32 k per MFMA instead of 16 -- the same operands, another association of the f32 sums:
        # coefficients to 2e-6 of the tensor's scale (the f16x2 GEMM's own error is 7e-7..9e-7 of it), the propagated bound to 1e-5
        T16, b16 = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=yb, next_bound=nb, x_planes=2, **kw)
        assert float((T16 - Tw).abs().max()) <= 2e-6 * float(Tw.abs().max()), float((T16 - Tw).abs().max()) / float(Tw.abs().max())
        assert float(((b16 - bw).abs() / bw.abs().clamp_min(1e-30)).max()) <= 1e-5
        assert torch.equal(T16.view(torch.int32), hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=yb, next_bound=nb, x_planes=True, **kw)[0].view(torch.int32)) == hip.MFMA16
        # ... launched as persistent workgroups that walk the tile list (roreg_gemm_persistent): the same MFMA sequence per element
        with hip.gemm_persistent(True):
            Tq, bq = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=yb, next_bound=nb, x_planes=2, **kw)
        assert torch.equal(T16.view(torch.int32), Tq.view(torch.int32)) and torch.equal(b16.view(torch.int32), bq.view(torch.int32))


@pytest.mark.parametrize('C,Oc,B,resid', [(256, 512, 20000, False), (512, 256, 19993, True), (32, 256, 9000, False)])
def test_persistent_gemm_launch_is_bitwise_the_per_tile_launch(group, C, Oc, B, resid):
    """The two other launch forms of the 16x16x32 GEMM -- hip.gemm_persistent(1): one workgroup per CU walks its share of the tile list (several
    tiles each at these sizes, tiles of all five irreps, ragged last column tiles) and requests the next tile's operands while it stores the
    finished one; (2): four-wave workgroups on 256 x 128 half tiles, two per CU, one K32 weight buffer and two barriers per step (a ragged last
    column tile's right half may be empty) -- return coefficients and the propagated bound bit for bit those of the launch with one eight-wave
    workgroup per tile, with and without the residual; twice each, so that a launch also starts from whatever the previous one left in LDS."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    torch.manual_seed(B)
    L = _Layer(torch.nn.Conv2d(C, Oc, (1, 13)))
    x = torch.randn(hip.coef_size(C, B), device='cuda') * torch.exp(torch.randn(hip.coef_size(C, B), device='cuda'))
    Xp, xb = hip.pack_coefs_f16x2(x, C, B)
    Xp = hip.words_to_planes(Xp, C, B)
    add = torch.randn(hip.coef_size(Oc, B), device='cuda') if resid else None
    nb = (torch.rand(Oc, device='cuda') + 0.5, torch.rand(Oc, device='cuda'))
    with hip.gemm_persistent(False):
        T0, b0 = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb, x_planes=2, add=add)
    for form in (1, 2, 1, 2):                                 # 1: persistent workgroups; 2: half tiles (256 x 128), two 4-wave workgroups per CU
        with hip.gemm_persistent(form):
            T1, b1 = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb, x_planes=2, add=add)
        assert torch.equal(T0.view(torch.int32), T1.view(torch.int32)) and torch.equal(b0.view(torch.int32), b1.view(torch.int32)), form


def test_gemm_bound_propagation(group):
    """The bound a GEMM epilogue hands to the next transform really bounds that transform's coefficients, per keypoint, and is not
    absurdly loose (the split keeps full accuracy while bound / max <= ~2^11)."""
    from roreg_amd import hip
    from roreg_amd.network.gf_fourier import _Layer
    rng = np.random.default_rng(21)
    B, C, Oc = 200, 64, 256
    L = _Layer(torch.nn.Conv2d(C, Oc, (1, 13)))
    x = torch.from_numpy((rng.standard_normal((B, C, 60)) * np.exp(2 * rng.standard_normal((B, 1, 1)))).astype(np.float32)).cuda()
    bn = (torch.from_numpy(rng.uniform(0.5, 1.5, Oc).astype(np.float32)).cuda(), torch.from_numpy(rng.standard_normal(Oc).astype(np.float32)).cuda())
    xb = hip.row_bound(x)
    Xp = hip.ft_nonlin(B, C, x_spatial=x, split='f16x2', out_bound=xb)
    T, nb = hip.irrep_gemm(Xp, None, C, Oc, B, f16x2=L.wsplit2, x_bound=xb, next_bound=hip.next_bound(bn, L.bias))
    Y = hip.ft_nonlin(B, Oc, coef_in=T, bias=L.bias, bn=bn, split=True)              # float32 coefficients of the next transform
    actual = hip.pack_coefs_f16x2(Y, Oc, B)[1]
    assert bool((nb >= actual).all())
    assert float((nb[:B] / actual[:B].clamp_min(1e-30)).max()) < 2.0 ** 9
    # and the split written under it decodes to the float32 coefficients
    Y16 = hip.unpack_coefs_f16x2(hip.ft_nonlin(B, Oc, coef_in=T, bias=L.bias, bn=bn, split='f16x2', out_bound=nb), nb, Oc, B)
    kp_scale = actual[hip._keypoint_of_columns(5, hip.coef_pitch(B))]                 # per-keypoint magnitude, irrep of dimension 5
    err = (hip.coef_views(Y16, Oc, B)[4] - hip.coef_views(Y, Oc, B)[4]).abs().amax(0) / kp_scale.clamp_min(1e-30)
    assert float(err.max()) < 2e-6


def test_extractor_is_batch_invariant(group):
    """A cloud's features are bit-identical whether it is extracted alone or inside a batch with other clouds, in every matrix-core mode
    (the reference's bs_GF independence, test/extractor.py:51-58): all block scales are per keypoint."""
    from roreg_amd.network import name2network
    net = name2network['GF_test'](default_config())
    synth.seeded_state_dict(net, 101)
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((333, 32, 60))).astype(np.float32)
    big = (50.0 * rng.standard_normal((700, 32, 60))).astype(np.float32)                # much larger magnitudes than `a`
    net.PartI_net.mode = 'fourier'
    net(torch.from_numpy(a[:8]))
    for mode in ('f16x2', 'bf16x3', 'f32'):
        net.PartI_net._fourier.gemm = mode
        alone = net(torch.from_numpy(a))['eqv']
        both = net(torch.from_numpy(np.concatenate([big[:401], a, big[401:]])))['eqv'][401:401 + 333]
        assert torch.equal(alone, both), mode
    net.PartI_net._fourier.gemm = 'f16x2'


def test_et_is_batch_invariant(group):
    from roreg_amd.network import name2network
    z = load_golden('et_forward')
    net = name2network['ET_test'](default_config())
    synth.seeded_state_dict(net, int(z['seed']))
    keys = ('before_eqv0', 'before_eqv1', 'after_eqv0', 'after_eqv1', 'pre_idx')
    rng = np.random.default_rng(6)
    for mode in ('f16x2', 'bf16x3', 'f32'):
        net.gemm = mode
        full = net({k: torch.from_numpy(z[k].copy()) for k in keys})['quaternion_pre']
        sub = slice(7, 23)
        scaled = {k: torch.from_numpy(z[k][sub].copy()) for k in keys}
        part = net(scaled)['quaternion_pre']
        assert torch.equal(part, full[sub]), mode
    net.gemm = 'f16x2'


def test_gf_both_gemm_modes_vs_golden(group):
    """The extractor in all three matrix-core modes (fp16 x 2 = default, bf16 x 3, f32-input MFMA) against the reference's output."""
    from roreg_amd.network import name2network
    z = load_golden('gf_forward')
    net = name2network['GF_test'](default_config())
    synth.seeded_state_dict(net, int(z['seed']))
    x = torch.from_numpy(z['x'])
    net.PartI_net.mode = 'fourier'
    net(x)                                              # builds the plan
    out = {}
    for mode in ('f32', 'bf16x3', 'f16x2'):
        net.PartI_net._fourier.gemm = mode
        out[mode] = net(x)['eqv'].cpu().numpy()
        assert np.abs(out[mode] - z['eqv']).max() < 1e-5, mode
    assert np.abs(out['f32'] - out['bf16x3']).max() < 5e-6 and np.abs(out['f32'] - out['f16x2']).max() < 5e-6


def test_et_both_gemm_modes_vs_golden(group):
    from roreg_amd.network import name2network
    z = load_golden('et_forward')
    net = name2network['ET_test'](default_config())
    synth.seeded_state_dict(net, int(z['seed']))
    q = {}
    for split in ('f32', 'bf16x3', 'f16x2'):
        net.gemm = split
        batch = {k: torch.from_numpy(z[k].copy()) for k in ('before_eqv0', 'before_eqv1', 'after_eqv0', 'after_eqv1', 'pre_idx')}
        q[split] = net(batch)['quaternion_pre'].cpu().numpy()
        assert np.abs(q[split] - z['quaternion']).max() < 1e-4, split
    assert np.abs(q['f32'] - q['bf16x3']).max() < 2e-5 and np.abs(q['f32'] - q['f16x2']).max() < 2e-5


@pytest.mark.parametrize('B', [37, 3000])
def test_packed_trunk_operand_equals_the_float_path(group, B):
    """The ET trunk convolution's packed operand (roreg_ft_nonlin_packed -> roreg_group_conv_f16x2_packed, round 5) against the float path it
    replaces, stage by stage on the same coefficients: (a) the words decode to ReLU(BN(h)) of the float inverse transform within 2^-19 of the
    row's bound (fp16 hi + lo resolve 22 bits below the block scale; the propagated bound sits <= 3 bits above the row maximum here);
    (b) the raw short-cut column is BITWISE column g = 0 of h; (c) the propagated bound really bounds every row; (d) the convolution on the
    words equals the float-input convolution to 2e-5 of the output scale; (e) a row's result does not depend on the other rows."""
    from roreg_amd import hip
    from roreg_amd.network import name2network
    net = name2network['ET_test'](default_config())
    synth.seeded_state_dict(net, 202)
    net.gemm = 'f16x2'
    res = net.PartII_SO3_Conv_layers[0]
    layer, bn0 = net._fourier_init()
    bn_t, nb_t = net._trunk_bn()
    ga, gb, gc, p0, gmap = net._pruned_gathers()
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn((B, 128, 60), device='cuda', generator=g)
    x = x / x.norm(dim=1, keepdim=True)
    x[3] *= 50.0; x[5] *= 1e-3                                            # rows of very different magnitude: the block scale is per row
    b0 = hip.row_bound(x, bn=bn0)
    X0 = hip.ft_nonlin(B, 128, x_spatial=x, bn=bn0, split='f16x2', out_bound=b0, planes=hip.use_planes(256))
    T0, b1 = hip.irrep_gemm(X0, None, 128, 256, B, f16x2=layer.wsplit2, x_bound=b0, next_bound=nb_t, x_planes=hip.use_planes(256))
    h, ah = hip.ft_nonlin(B, 256, coef_in=T0, bias=layer.bias, spatial_out=True, g_map=gmap, Lout=net.LIVE_PAD, Lvalid=45, split='f16x2', want_rowmax=True)
    hw, h0 = hip.ft_nonlin_packed(B, 256, T0, layer.bias, bn_t, b1, g_map=gmap, Lout=net.LIVE_PAD, Lvalid=45, raw_g=0)
    assert torch.equal(h0, h[:, :, p0])                                                             # (b)
    act = torch.relu(h * bn_t[0][None, :, None] + bn_t[1][None, :, None]); act[:, :, 45:] = 0
    assert bool((act.abs().amax(dim=(1, 2)) <= b1[:B]).all())                                       # (c)
    e = hip.bound_exp(b1[:B]).double()
    w = hw.to(torch.int64) & 0xffffffff
    hi = (w & 0xffff).to(torch.int16).view(torch.float16).double(); lo = ((w >> 16) & 0xffff).to(torch.int16).view(torch.float16).double()
    dec = (hi + lo) * torch.pow(2.0, -e)[:, None, None]
    err = (dec - act.double()).abs().amax(dim=(1, 2)) / b1[:B].double()
    assert float(err.max()) < 2.0 ** -19, float(err.max())                                           # (a)
    headroom = torch.log2(b1[:B].double() / act.abs().amax(dim=(1, 2)).double().clamp_min(1e-300))
    print(f'bound headroom over the row maximum: {float(headroom.min()):.2f} .. {float(headroom.max()):.2f} bits')
    m_ref, _ = res._b_in(h, gather=gb, in_rowmax=ah, want_rowmax=True, lds_order=net._trunk_lds_order())
    m_pk, am = hip.group_conv_packed(hw, res._b_in.plan(), b1, gb, want_rowmax=True, lds_order=net._trunk_lds_order())
    scale = m_ref.abs().amax(dim=(1, 2)).clamp_min(1e-30)
    assert float(((m_pk - m_ref).abs().amax(dim=(1, 2)) / scale).max()) < 2e-5                      # (d)
    assert torch.allclose(am, m_pk.abs().amax(dim=(1, 2)))
    sub = slice(2, 9)                                                                                # (e) seven rows alone: bitwise the same
    xs = x[sub].contiguous(); nb = xs.shape[0]
    b0s = hip.row_bound(xs, bn=bn0)
    X0s = hip.ft_nonlin(nb, 128, x_spatial=xs, bn=bn0, split='f16x2', out_bound=b0s, planes=hip.use_planes(256))
    T0s, b1s = hip.irrep_gemm(X0s, None, 128, 256, nb, f16x2=layer.wsplit2, x_bound=b0s, next_bound=nb_t, x_planes=hip.use_planes(256))
    hws, _ = hip.ft_nonlin_packed(nb, 256, T0s, layer.bias, bn_t, b1s, g_map=gmap, Lout=net.LIVE_PAD, Lvalid=45, raw_g=0)
    assert torch.equal(hws, hw[sub]) and torch.equal(b1s[:nb], b1[sub])
    assert torch.equal(hip.group_conv_packed(hws, res._b_in.plan(), b1s, gb, lds_order=net._trunk_lds_order()), m_pk[sub])
