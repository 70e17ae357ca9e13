"""ORACLE -- test infrastructure only (see oracle/ref_numpy.py header).  numpy restatement of the rotation-coherence
matcher Match_ot (network/rot_coh_match.py:8-390).  Parity status: PINNED against tests/golden/match_ot.npz (the
reference run with its shipped RM checkpoint), tests/test_oracle_golden.py::test_match_ot_matches_reference.

Layout: every per-point tensor is position-major here ([points, channels] or [points, k, channels]); the reference is
channel-major ([1,C,points,k]).  `taps` collects intermediates for the kernel tests.
Top-k: the reference takes the first k of a full descending argsort (rot_coh_match.py:34-45; ties unspecified); here
ties go to the lower index.
"""
import numpy as np

f32 = np.float32
G = 60


def conv1x1(x, sd, name):
    """x [...,Cin] -> [...,Cout] with the Conv2d(1x1) weights `name`.weight [Cout,Cin,1,1], `name`.bias."""
    W = sd[name + '.weight'][:, :, 0, 0].astype(f32)
    return (x @ W.T + sd[name + '.bias'].astype(f32)).astype(f32)


def instance_norm(h):
    """InstanceNorm2d(affine=False, eps=1e-5): per channel over all positions (biased variance).  h [L,C]."""
    h64 = h.astype(np.float64)
    mu = h64.mean(0, keepdims=True)
    var = h64.var(0, keepdims=True)
    return ((h64 - mu) / np.sqrt(var + 1e-5)).astype(f32)


def mlp_2layer(x, sd, name):
    """rot_coh_match.py:14-32 / Contextnorm :63-81.  x [..., Cin] (all leading dims are positions)."""
    shp = x.shape[:-1]
    xf = x.reshape(-1, x.shape[-1])
    h = conv1x1(xf, sd, name + '.net.0')
    h = np.maximum(instance_norm(h), 0)
    y = conv1x1(h, sd, name + '.net.3')
    if (name + '.res.weight') in sd:
        y = y + conv1x1(xf, sd, name + '.res')
    return y.reshape(shp + (y.shape[-1],)).astype(f32)


def topk_rows(score, k):
    """indices of the k largest entries per row, descending, lower index first on ties."""
    order = np.argsort(-score, axis=1, kind='stable')
    return order[:, :k]


def mha(q, key, value, sd, name):
    """MultiHeadedAttention(4,32) (rot_coh_match.py:84-119).  q [m,32], key/value [m,k,32] -> [m,32].
    `.view(b, 8, 4, -1)` maps channel c to (d = c//4, head = c%4)."""
    m, k, _ = key.shape
    qp = conv1x1(q, sd, name + '.proj.0').reshape(m, 8, 4)
    kp = conv1x1(key, sd, name + '.proj.1').reshape(m, k, 8, 4)
    vp = conv1x1(value, sd, name + '.proj.2').reshape(m, k, 8, 4)
    scores = np.einsum('pdh,pjdh->phj', qp, kp) / f32(8 ** .5)
    scores = scores - scores.max(-1, keepdims=True)
    e = np.exp(scores)
    prob = e / e.sum(-1, keepdims=True)
    x = np.einsum('phj,pjdh->pdh', prob, vp).reshape(m, 32)
    return conv1x1(x.astype(f32), sd, name + '.merge')


def r_indicator(src_eqv, nn_eqv, P, s2t):
    """rot_coh_match.py:154-163.  src_eqv [m,32,60] (the block's own cloud), nn_eqv [m,32,60] (eqv feature of each
    point's best match in the other cloud) -> [m,60].
      s2t : R[p,h] = sum_f sum_g src[p,f,P[g,h]] * nn[p,f,g]
      t2s : R[p,h] = sum_f sum_g nn[p,f,P[g,h]] * src[p,f,g]"""
    a, b = (src_eqv, nn_eqv) if s2t else (nn_eqv, src_eqv)
    m = a.shape[0]
    out = np.zeros((m, G), f32)
    for h in range(G):
        out[:, h] = np.einsum('pfg,pfg->p', a[:, :, P[:, h]], b)
    return out


def cross_block(source, target, source_eqv, target_eqv, featinv, sd, name, k, s2t, P, taps=None):
    score = source @ target.T
    knn = topk_rows(score, k)
    nn = knn[:, 0]
    knn_fea = target[knn]                                            # [m,k,32]
    att = mha(source, knn_fea, knn_fea, sd, name + '.cross_attn')
    feat = mlp_2layer(np.concatenate([featinv, source, att], 1), sd, name + '.merge')
    R = r_indicator(source_eqv, target_eqv[nn], P, s2t)
    if taps is not None:
        taps[name] = dict(knn=knn, att=att, feat=feat, R=R)
    return feat, R


def l2n(x):
    return (x / np.sqrt((x * x).sum(-1, keepdims=True))).astype(f32)


def self_block(feat, coor, R_ind, featinv, sd, name, k, taps=None):
    score = feat @ feat.T
    knn = topk_rows(score, k)
    knn_fea = feat[knn]                                              # [m,k,32]
    knn_coor = coor[knn] - coor[:, None, :]                          # [m,k,3]
    pos = mlp_2layer(knn_coor, sd, name + '.pos_en')                 # [m,k,32]
    ctx = np.concatenate([R_ind, np.broadcast_to(R_ind.max(0, keepdims=True), R_ind.shape)], 1)    # [m,120]
    conf = mlp_2layer(ctx, sd, name + '.ambiguity')                  # [m,32]
    pos = l2n(pos); knn_fea = l2n(knn_fea); conf = l2n(conf)
    val_in = np.concatenate([pos, knn_fea, np.broadcast_to(conf[:, None, :], pos.shape)], -1)        # [m,k,96]
    value = mlp_2layer(val_in, sd, name + '.val_en')
    att = mha(feat, knn_fea, value, sd, name + '.self_attn')
    out = mlp_2layer(np.concatenate([featinv, feat, att], 1), sd, name + '.merge')
    if taps is not None:
        taps[name] = dict(knn=knn, pos=pos, conf=conf, value=value, att=att, out=out)
    return out


def log_sinkhorn(score, alpha, iters):
    """sinkhorn_ot.log_optimal_transport (rot_coh_match.py:285-314).  score [m,n] -> Z [(m+1),(n+1)]."""
    m, n = score.shape
    Z0 = np.full((m + 1, n + 1), alpha, f32)
    Z0[:m, :n] = score
    norm = f32(-np.log(f32(m + n)))
    log_mu = np.concatenate([np.full(m, norm, f32), [f32(np.log(f32(n))) + norm]]).astype(f32)
    log_nu = np.concatenate([np.full(n, norm, f32), [f32(np.log(f32(m))) + norm]]).astype(f32)
    u = np.zeros(m + 1, f32); v = np.zeros(n + 1, f32)

    def lse(a, axis):
        mx = a.max(axis, keepdims=True)
        return (mx + np.log(np.exp(a - mx).sum(axis, keepdims=True))).squeeze(axis).astype(f32)
    for _ in range(iters):
        u = log_mu - lse(Z0 + v[None, :], 1)
        v = log_nu - lse(Z0 + u[:, None], 0)
    return (Z0 + u[:, None] + v[None, :] - norm).astype(f32)


def readout(Z):
    """rot_coh_match.py:369-379 -> matches0 [m], matches1 [n], mscores0, mscores1."""
    S = Z[:-1, :-1]
    i0 = S.argmax(1); i1 = S.argmax(0)
    m0 = np.arange(S.shape[0]) == i1[i0]
    m1 = np.arange(S.shape[1]) == i0[i1]
    ms0 = np.where(m0, np.exp(S.max(1)), 0).astype(f32)
    ms1 = np.where(m1, ms0[i1], 0).astype(f32)
    v1 = m1 & m0[i1]
    return np.where(m0, i0, -1), np.where(v1, i1, -1), ms0, ms1


def match_ot_forward(batch, sd, P, iters=100, taps=None):
    """Match_ot.forward (rot_coh_match.py:339-390).  batch: feats0 [1,m,32,60], feats1 [1,n,32,60], keys0 [1,m,3], keys1 [1,n,3]."""
    src_eqv = batch['feats0'][0].astype(f32); tgt_eqv = batch['feats1'][0].astype(f32)
    src_coor = (batch['keys0'][0] / f32(0.025)).astype(f32); tgt_coor = (batch['keys1'][0] / f32(0.025)).astype(f32)
    src_inv = src_eqv.mean(-1).astype(f32); tgt_inv = tgt_eqv.mean(-1).astype(f32)
    source, target = src_inv, tgt_inv
    sources, targets = [], []
    for bi, k in enumerate([16, 8]):
        p = f'Graph.merge_blocks.{bi}'
        s2t, R_s = cross_block(source, target, src_eqv, tgt_eqv, src_inv, sd, p + '.cross_graph_s2t', k, True, P, taps)
        eh_s = self_block(s2t, src_coor, R_s, src_inv, sd, p + '.self_graph_s', k, taps)
        t2s, R_t = cross_block(target, source, tgt_eqv, src_eqv, tgt_inv, sd, p + '.cross_graph_t2s', k, False, P, taps)
        eh_t = self_block(t2s, tgt_coor, R_t, tgt_inv, sd, p + '.self_graph_t', k, taps)
        source, target = eh_s, eh_t
        sources.append(source); targets.append(target)
    s_fin = mlp_2layer(np.concatenate([src_inv, sources[-1]], 1), sd, 'final_mlp')
    t_fin = mlp_2layer(np.concatenate([tgt_inv, targets[-1]], 1), sd, 'final_mlp')
    score = (s_fin @ t_fin.T).astype(f32)
    Z = log_sinkhorn(score, f32(sd['ot_layer.bin_score']), iters)
    m0, m1, ms0, ms1 = readout(Z)
    # scores_other (training-only supervision, rot_coh_match.py:355-358)
    so = np.stack([sources[i] @ targets[i].T for i in range(2)], -1)                      # [m,n,2]

    def softmax(a, axis):
        e = np.exp(a - a.max(axis, keepdims=True))
        return e / e.sum(axis, keepdims=True)
    scores_other = (softmax(so, 0) * softmax(so, 1)).astype(f32)
    return {'scores': Z[None], 'scores_other': scores_other[None], 'matches0': m0[None], 'matches1': m1[None],
            'matching_scores0': ms0[None], 'matching_scores1': ms1[None], 'source_final': s_fin.T[None, :, :, None],
            'target_final': t_fin.T[None, :, :, None]}
