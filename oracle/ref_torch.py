"""ORACLE -- test infrastructure only.  The reference's hot path restated with the torch CPU operators the reference itself calls
(conv2d on the gathered 13-stencil tensor, batch_norm(eval), einsum, the chunked broadcast-difference nearest-neighbour search), so that it
runs on ALL host cores (torch.set_num_threads): the CPU baseline SURVEY 8d defines ("the build's CPU restatement, validated against the
reference, on the GPU box's host cores with torch.set_num_threads(all)").  Only tests/ and bench.py's cpu_baseline leg may import this
module; the product (roreg_amd/) never does.

Parity status: PINNED -- tests/test_oracle_golden.py::test_torch_restatement_matches_reference checks every function against the vectors
tools/gen_golden.py produced by running the actual reference (GF / ET / Des2R / nearest neighbours / mutual matches), at the tolerances
of the numpy oracle (oracle/ref_numpy.py).  Citations are reference file:line.

State dicts are {name: torch tensor} with the reference's key names."""
import numpy as np
import torch
import torch.nn.functional as F

G, K = 60, 13


def _gather13(x, nei):
    """data_process (network/group_feat.py:20-24, ops.py:45-51, eqv_trans.py:103-109): [B,C,60] -> [B,C,60,13]."""
    B, C, _ = x.shape
    return x[:, :, nei].reshape(B, C, G, K)


def _comb(x, sd, prefix, nei):
    """BatchNorm2d(eval) -> ReLU -> Conv2d(C, O, (1,13)) on the gathered tensor (network/ops.py:11-20)."""
    h = F.batch_norm(_gather13(x, nei), sd[prefix + '.0.running_mean'], sd[prefix + '.0.running_var'], sd[prefix + '.0.weight'],
                     sd[prefix + '.0.bias'], False, 0.0, 1e-5)
    return F.conv2d(F.relu(h), sd[prefix + '.2.weight'], sd[prefix + '.2.bias'])[..., 0]


def _residual(x, sd, prefix, nei):
    """Residual_Comb_Conv (network/ops.py:22-64)."""
    h = _comb(_comb(x, sd, prefix + '.comb_layer_in', nei), sd, prefix + '.comb_layer_out', nei)
    sc = _comb(x, sd, prefix + '.short_cut_layer', nei) if (prefix + '.short_cut_layer.2.weight') in sd else x
    return h + sc


def gf_forward(x, sd, Nei):
    """GF_test.forward (network/group_feat.py:26-45,80-87): x [B,32,60] float32 tensor -> {'eqv','inv'}."""
    nei = torch.as_tensor(np.asarray(Nei).reshape(-1), dtype=torch.long)
    p = 'PartI_net.'
    with torch.no_grad():
        t0 = F.conv2d(_gather13(x, nei), sd[p + 'Conv_in.0.weight'], sd[p + 'Conv_in.0.bias'])[..., 0]
        t1 = _residual(t0, sd, p + 'SO3_Conv_layers.0', nei)
        eqv = _comb(t1, sd, p + 'Conv_out.comb_layer', nei) + x
        inv = eqv.mean(-1)
        eqv = eqv / torch.clamp_min(torch.norm(eqv, dim=1, keepdim=True), 1e-4)
        inv = inv / torch.clamp_min(torch.norm(inv, dim=1, keepdim=True), 1e-4)
    return {'eqv': eqv, 'inv': inv}


def et_forward(batch, sd, Nei, P):
    """ET_test.forward (network/eqv_trans.py:119-138): the 1x1 head evaluated at every g, column g = 0 kept, as written."""
    nei = torch.as_tensor(np.asarray(Nei).reshape(-1), dtype=torch.long)
    Pt = torch.as_tensor(np.asarray(P), dtype=torch.long)
    with torch.no_grad():
        idx = Pt[batch['pre_idx'].long()][:, None, :].expand(-1, 32, -1)
        x = torch.cat([torch.gather(batch['before_eqv0'], 2, idx), batch['before_eqv1'], torch.gather(batch['after_eqv0'], 2, idx),
                       batch['after_eqv1']], 1)
        h = _residual(_comb(x, sd, 'Conv_init.comb_layer', nei), sd, 'PartII_SO3_Conv_layers.0', nei)[..., None]      # [B,256,60,1]
        fc = 'PartII_To_R_FC.'
        z = F.conv2d(h, sd[fc + '0.weight'], sd[fc + '0.bias'])
        z = F.relu(F.batch_norm(z, sd[fc + '1.running_mean'], sd[fc + '1.running_var'], sd[fc + '1.weight'], sd[fc + '1.bias'], False, 0.0, 1e-5))
        z = F.conv2d(z, sd[fc + '3.weight'], sd[fc + '3.bias'])
        z = F.relu(F.batch_norm(z, sd[fc + '4.running_mean'], sd[fc + '4.running_var'], sd[fc + '4.weight'], sd[fc + '4.bias'], False, 0.0, 1e-5))
        q = F.conv2d(z, sd[fc + '6.weight'], sd[fc + '6.bias'])[:, :, 0, 0]
        return q / torch.norm(q, dim=1, keepdim=True)


def nn_search(target, source, chunk=500):
    """knn_module.KNN(1) (utils/knn_search.py:17-24,41-44,138-162): for every source row the nearest target row,
    d = sqrt(sum (s - t)^2 + 1e-7) by explicit differences in chunks of 500 -> (d [m], idx [m])."""
    ds, ids = [], []
    for s in range(0, source.shape[0], chunk):
        D = torch.sqrt(torch.sum((source[s:s + chunk, None, :] - target[None, :, :]) ** 2, 2) + 1e-7)
        d, i = D.min(dim=1)
        ds.append(d); ids.append(i)
    return torch.cat(ds), torch.cat(ids)


def mutual_match(eqv0, eqv1, sample0, sample1):
    """mutual.run for one pair (test/matcher.py:67-107): eqv tensors [N,32,60], sample index arrays -> matches [M,2] int64 (numpy)."""
    def inv(e):                                                   # matcher.py:69-72 is numpy
        f = np.mean(e.numpy(), axis=-1)
        return torch.from_numpy(f / (np.sqrt(np.sum(np.square(f), axis=1, keepdims=True)) + 1e-5))
    f0 = inv(eqv0)[torch.as_tensor(sample0)]; f1 = inv(eqv1)[torch.as_tensor(sample1)]
    nn01 = nn_search(f1, f0)[1].numpy(); nn10 = nn_search(f0, f1)[1].numpy()
    i = np.arange(nn01.shape[0])
    keep = nn10[nn01] == i
    return np.stack([np.asarray(sample0)[i[keep]], np.asarray(sample1)[nn01[keep]]], 1).astype(np.int64)


def des2r(d1, d2, P):
    """Batch_Des2R_torch (test/estimator.py:85-89): permuted copy [B,32,60,60], einsum, argmax."""
    Pt = torch.as_tensor(np.asarray(P).reshape(-1), dtype=torch.long)
    B = d1.shape[0]
    cor = torch.einsum('bfag,bfg->ba', d1[:, :, Pt].reshape(B, 32, G, G), d2)
    return torch.argmax(cor, dim=1)


def overlap_cal(k0, k1, T, scores, ird):
    """yohoo_ransac.overlap_cal (test/estimator.py:377-382): numpy in the reference, numpy here."""
    p = k1 @ T[:, :3].T + T[:, 3:].T
    inl = np.where(np.sum(np.square(k0 - p), axis=-1) < ird * ird)[0]
    return np.sum(scores[inl]) / scores.shape[0]
