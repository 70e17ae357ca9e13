"""ORACLE -- test infrastructure only.  CPU (numpy) restatement of the reference's algorithms for the
per-pair registration hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (roreg_amd/) never does.

Parity status: PINNED -- every function here is checked in tests/test_oracle_golden.py against vectors
produced by running the actual reference in the build container (tools/gen_golden.py ->
tests/golden/*.npz).  Citations are reference file:line (relative to the reference checkout).

Conventions fixed here (the HIP kernels follow the same operation order where the contract is
bit-exactness; floating-point network outputs are compared with a stated tolerance instead):
  * nearest-neighbour distance: acc=0; for f in 0..F-1: acc = acc + (s_f - t_f)*(s_f - t_f)   (fp32, no FMA),
    d = sqrt(acc + 1e-7); argmin takes the first minimum.
  * Des2R correlation: s_f = sum over g (in order) of d1[f,P[a,g]]*d2[f,g]; cor[a] = sum over f (in order)
    of s_f   (fp32, no FMA); argmax takes the first maximum.
  * RANSAC point test (fp64, no FMA): p_i = ((k1x*R_i0 + k1y*R_i1) + k1z*R_i2) + t_i;
    d2 = ((dx*dx + dy*dy) + dz*dz);  inlier <=> d2 < ird*ird.
"""
import numpy as np

G, K = 60, 13
f32 = np.float32


# ====================================================================================================
# group convolution and the networks built from it
# ====================================================================================================
def bn_relu(x, sd, prefix, eps=1e-5):
    """BatchNorm2d(eval) -> ReLU on [B,C,60]  (network/ops.py:14-16,27-28,32-33)."""
    g = sd[prefix + '.weight'].astype(f32); b = sd[prefix + '.bias'].astype(f32)
    m = sd[prefix + '.running_mean'].astype(f32); v = sd[prefix + '.running_var'].astype(f32)
    scale = g / np.sqrt(v + f32(eps))
    shift = b - m * scale
    return np.maximum(x * scale[None, :, None] + shift[None, :, None], f32(0)).astype(f32)


def group_conv(x, W, bias, Nei):
    """out[b,o,g] = bias[o] + sum_c sum_k W[o,c,0,k] * x[b,c,Nei[g,k]]
    (network/group_feat.py:20-24 data_process + Conv2d(C,O,(1,13)); network/ops.py:45-51)."""
    B, C, _ = x.shape
    O = W.shape[0]
    xg = x[:, :, Nei.reshape(-1)].reshape(B, C, G, K)                 # gather
    A = np.ascontiguousarray(xg.transpose(0, 2, 1, 3)).reshape(B * G, C * K)
    Wm = W.reshape(O, C * K)
    out = A @ Wm.T + bias[None, :]
    return np.ascontiguousarray(out.reshape(B, G, O).transpose(0, 2, 1)).astype(f32)


def comb_conv(x, sd, prefix, Nei):
    """Comb_Conv / the three branches of Residual_Comb_Conv: BN -> ReLU -> group conv (ops.py:11-20)."""
    return group_conv(bn_relu(x, sd, prefix + '.0'), sd[prefix + '.2.weight'], sd[prefix + '.2.bias'], Nei)


def residual_comb_conv(x, sd, prefix, Nei):
    """network/ops.py:22-64."""
    h = comb_conv(x, sd, prefix + '.comb_layer_in', Nei)
    h = comb_conv(h, sd, prefix + '.comb_layer_out', Nei)
    if (prefix + '.short_cut_layer.2.weight') in sd:
        sc = comb_conv(x, sd, prefix + '.short_cut_layer', Nei)
    else:
        sc = x
    return h + sc


def gf_forward(x, sd, Nei, taps=False):
    """GF_test.forward -> {'eqv','inv'}  (network/group_feat.py:26-45,80-87)."""
    p = 'PartI_net.'
    t0 = group_conv(x, sd[p + 'Conv_in.0.weight'], sd[p + 'Conv_in.0.bias'], Nei)
    t1 = residual_comb_conv(t0, sd, p + 'SO3_Conv_layers.0', Nei)
    y = comb_conv(t1, sd, p + 'Conv_out.comb_layer', Nei)
    eqv = y + x
    inv = eqv.mean(-1)
    eqv = eqv / np.maximum(np.sqrt((eqv * eqv).sum(1, keepdims=True)), f32(1e-4))
    inv = inv / np.maximum(np.sqrt((inv * inv).sum(1, keepdims=True)), f32(1e-4))
    out = {'eqv': eqv.astype(f32), 'inv': inv.astype(f32)}
    if taps:
        out.update(conv_in=t0, res=t1)
    return out


def rd_encoder(x, sd, Nei):
    return residual_comb_conv(x, sd, 'eqv_encoder.0', Nei)


def rd_scores_from_encoding(f, P):
    """normalise over 16 ch (no eps), c[a] = sum_f sum_b f[f,P[a,b]] f[f,b], unbiased std over a
    (network/rot_detect.py:47-52)."""
    f = f / np.sqrt((f * f).sum(1, keepdims=True))
    B, C, _ = f.shape
    fp = f[:, :, P.reshape(-1)].reshape(B, C, G, G)
    c = np.einsum('mfab,mfb->ma', fp, f)
    return c.std(axis=1, ddof=1).astype(f32)


def rd_forward(x, sd, Nei, P):
    """detector_eqv_test.forward -> raw scores [B]  (network/rot_detect.py:43-55)."""
    return rd_scores_from_encoding(rd_encoder(x, sd, Nei), P)


def det_rank_scores(scores):
    """test/detector.py:45-46: replace scores by rank/N (argsort = numpy default quicksort)."""
    s = scores.copy()
    a = np.argsort(s)
    s[a] = np.arange(s.shape[0]) / s.shape[0]
    return s


def et_forward(batch, sd, Nei, P):
    """ET_test.forward -> quaternion [B,4]  (network/eqv_trans.py:119-138).
    The 1x1 head is evaluated at every g and column g=0 is kept, exactly as written (the product's
    dead-output pruning must reproduce this)."""
    b0 = batch['before_eqv0']; b1 = batch['before_eqv1']; a0 = batch['after_eqv0']; a1 = batch['after_eqv1']
    pre = np.asarray(batch['pre_idx']).astype(np.int64)
    B = b0.shape[0]
    idx = P[pre]                                             # [B,60]
    b0p = np.take_along_axis(b0, idx[:, None, :].repeat(32, 1), axis=2)
    a0p = np.take_along_axis(a0, idx[:, None, :].repeat(32, 1), axis=2)
    x = np.concatenate([b0p, b1, a0p, a1], 1).astype(f32)    # [B,128,60]
    h = comb_conv(x, sd, 'Conv_init.comb_layer', Nei)
    h = residual_comb_conv(h, sd, 'PartII_SO3_Conv_layers.0', Nei)        # [B,256,60]

    def conv1x1(z, w, b):
        return np.einsum('oc,bcg->bog', w[:, :, 0, 0], z) + b[None, :, None]

    def bn_relu_named(z, pre_):
        g = sd[pre_ + '.weight']; bb = sd[pre_ + '.bias']; m = sd[pre_ + '.running_mean']; v = sd[pre_ + '.running_var']
        sc = g / np.sqrt(v + f32(1e-5)); sh = bb - m * sc
        return np.maximum(z * sc[None, :, None] + sh[None, :, None], 0)
    z = conv1x1(h, sd['PartII_To_R_FC.0.weight'], sd['PartII_To_R_FC.0.bias'])
    z = bn_relu_named(z, 'PartII_To_R_FC.1')
    z = conv1x1(z, sd['PartII_To_R_FC.3.weight'], sd['PartII_To_R_FC.3.bias'])
    z = bn_relu_named(z, 'PartII_To_R_FC.4')
    z = conv1x1(z, sd['PartII_To_R_FC.6.weight'], sd['PartII_To_R_FC.6.bias'])
    q = z[:, :, 0]
    q = q / np.sqrt((q * q).sum(1))[:, None]
    return q.astype(f32)


# ====================================================================================================
# nearest neighbours, NMS, mutual matching
# ====================================================================================================
def pdist(S, T, dist_type='L2'):
    """modified_knn_matcher.pdist (knn_search.py:17-24) with the fixed accumulation order of this oracle.  S [m,f], T [n,f] -> [m,n]."""
    S = S.astype(f32); T = T.astype(f32)
    acc = np.zeros((S.shape[0], T.shape[0]), f32)
    for f in range(S.shape[1]):
        d = S[:, f][:, None] - T[:, f][None, :]
        acc = acc + d * d
    if dist_type == 'SquareL2':
        return acc
    if dist_type != 'L2':
        raise NotImplementedError('Not implemented')
    return np.sqrt(acc + f32(1e-7))


def pdist_l2(S, T):
    return pdist(S, T, 'L2')


def knn(target, source, k=1, chunk=500, dist_type='L2'):
    """knn_module.KNN(k)(target_F[1,f,n], source_F[1,f,m]) on [n,f],[m,f] arrays (knn_search.py:138-162) = find_nn_gpu / find_knn_gpu
    (:26-103) on (source, target).  Returns (d, idx): k==1 -> [m],[m];  k>=2 -> [m,k],[m,k] ordered by increasing distance."""
    m = source.shape[0]
    ds, ids = [], []
    for s in range(0, m, chunk):
        D = pdist(source[s:s + chunk], target, dist_type)
        if k < 2:
            i = D.argmin(1)
            ds.append(D[np.arange(D.shape[0]), i]); ids.append(i)
        else:
            i = np.argsort(D, axis=1, kind='stable')[:, :k]       # topk(-d): ascending d, first index on ties
            ds.append(np.take_along_axis(D, i, 1)); ids.append(i)
    return np.concatenate(ds, 0), np.concatenate(ids, 0).astype(np.int64)


def find_corr(F0, F1, mutual=True):
    """modified_knn_matcher.find_corr without subsampling (knn_search.py:105-136): 'SquareL2' (mutual) nearest neighbours."""
    _, nn01 = knn(F1, F0, 1, dist_type='SquareL2')
    if not mutual:
        return np.arange(F0.shape[0]), nn01
    _, nn10 = knn(F0, F1, 1, dist_type='SquareL2')
    keep = [i for i in range(len(nn01)) if nn10[nn01[i]] == i]
    return np.array(keep, np.int64), nn01[keep]


def nms_sample(keys, scores, num, k=5):
    """NMS_sample.sample (test/matcher.py:18-42)."""
    if keys.shape[0] < num:
        return np.arange(keys.shape[0])
    _, argmin = knn(keys.astype(f32), keys.astype(f32), k=k)
    nei_max = scores[argmin.reshape(-1)].reshape(-1, k).max(-1)
    sam = np.where(scores >= nei_max)[0]
    if sam.shape[0] > num:
        ss = scores[sam]; ss = ss / np.sum(ss)
        sam = sam[np.argsort(ss)[-num:]]
    if sam.shape[0] < num:
        left = num - sam.shape[0]
        il = np.where(scores < nei_max)[0]
        li = il[np.argsort(scores[il])[-left:]]
        sam = np.concatenate([sam, li], 0)
    return sam


def inv_descriptor(eqv):
    """test/matcher.py:69-72: mean over g (numpy f32) then / (norm + 1e-5)."""
    f = np.mean(eqv, axis=-1).astype(f32)
    return f / (np.sqrt(np.sum(np.square(f), axis=1, keepdims=True)) + 1e-5)


def mutual_check(nn01, nn10):
    """matcher.py:98-105: keep i with nn10[nn01[i]] == i, increasing i."""
    i = np.arange(nn01.shape[0])
    keep = nn10[nn01] == i
    return np.stack([i[keep], nn01[keep]], 1).astype(np.int64)


def mutual_match(eqv0, eqv1, sample0, sample1):
    """mutual.run core for one pair (matcher.py:67-107) given the sample index arrays."""
    f0 = inv_descriptor(eqv0)[sample0]; f1 = inv_descriptor(eqv1)[sample1]
    _, nn01 = knn(f1, f0, 1)      # KNN(feats1, feats0): nearest in 1 for each of 0
    _, nn10 = knn(f0, f1, 1)
    m = mutual_check(nn01, nn10)
    m[:, 0] = sample0[m[:, 0]]; m[:, 1] = sample1[m[:, 1]]
    return m


# ====================================================================================================
# estimator
# ====================================================================================================
def des2r_cor(d1, d2, P):
    """cor[b,a] = sum_f sum_g d1[b,f,P[a,g]] d2[b,f,g] in the oracle's fixed order
    (test/estimator.py:85-89)."""
    B = d1.shape[0]
    d1 = d1.astype(f32); d2 = d2.astype(f32)
    s = np.zeros((B, 32, G), f32)                      # [b,f,a]
    for g in range(G):
        s = s + d1[:, :, P[:, g]] * d2[:, :, g][:, :, None]
    cor = np.zeros((B, G), f32)
    for f in range(32):
        cor = cor + s[:, f, :]
    return cor


def des2r(d1, d2, P):
    return des2r_cor(d1, d2, P).argmax(1).astype(np.int64)


def matrix_from_quaternion(q):
    """utils/r_eval.py:90-106 (w,x,y,z); arithmetic in the dtype of q (float32 from the network),
    stored into a float64 matrix."""
    w, x, y, z = q[0], q[1], q[2], q[3]
    m = np.eye(3)
    m[0, 0] = 1 - 2 * y * y - 2 * z * z; m[0, 1] = 2 * x * y - 2 * z * w; m[0, 2] = 2 * x * z + 2 * y * w
    m[1, 0] = 2 * x * y + 2 * z * w; m[1, 1] = 1 - 2 * x * x - 2 * z * z; m[1, 2] = 2 * y * z - 2 * x * w
    m[2, 0] = 2 * x * z - 2 * y * w; m[2, 1] = 2 * y * z + 2 * x * w; m[2, 2] = 1 - 2 * x * x - 2 * y * y
    return m


def rt_pre(q, anchor, Rgroup_f32, keys0, keys1):
    """test/estimator.py:353-365: R = R_res(q) @ Rgroup[a] (float64 @ float32 -> float64),
    t = key0 - key1 @ R.T ; returns [M,3,4] float64."""
    M = q.shape[0]
    out = np.zeros((M, 3, 4))
    for i in range(M):
        R = matrix_from_quaternion(q[i]) @ Rgroup_f32[int(anchor[i])]
        t = keys0[i] - keys1[i] @ R.T
        out[i, :, :3] = R; out[i, :, 3] = t
    return out


def transform_points_3x4(p, T):
    """utils/utils.py:42-43 with this oracle's fixed operation order."""
    x = ((p[:, 0] * T[0, 0] + p[:, 1] * T[0, 1]) + p[:, 2] * T[0, 2]) + T[0, 3]
    y = ((p[:, 0] * T[1, 0] + p[:, 1] * T[1, 1]) + p[:, 2] * T[1, 2]) + T[1, 3]
    z = ((p[:, 0] * T[2, 0] + p[:, 1] * T[2, 1]) + p[:, 2] * T[2, 2]) + T[2, 3]
    return np.stack([x, y, z], 1)


def inlier_mask(k0, k1, T, dist):
    p = transform_points_3x4(k1, T)
    d = k0 - p
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    return d2 < dist * dist


def overlap_cal(k0, k1, T, scores, ird):
    """yohoo_ransac.overlap_cal (estimator.py:377-382): sum(scores[inliers]) / M."""
    inl = np.where(inlier_mask(k0, k1, T, ird))[0]
    return np.sum(scores[inl]) / scores.shape[0]


def np_sum_f32_model(a):
    """What np.sum does to a contiguous float32 vector, written out (numpy/_core/src/umath/loops_utils.h.src `pairwise_sum`, the reduction
    loop's 8192-element buffer): the order the HIP kernels rebuild for float32 match scores (csrc/ransac.hip).  Pure-Python loops, small
    cases only; tests compare it with np.sum itself."""
    f = np.float32

    def leaf(x):
        n = len(x)
        if n < 8:
            r = f(0)
            for v in x:
                r = f(r + v)
            return r
        r = [x[j] for j in range(8)]
        nb = n - n % 8
        for i in range(8, nb, 8):
            for j in range(8):
                r[j] = f(r[j] + x[i + j])
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        for i in range(nb, n):
            res = f(res + x[i])
        return res

    def pairwise(x):
        n = len(x)
        if n <= 128:
            return leaf(x)
        n2 = 8 * (n // 16)
        return f(pairwise(x[:n2]) + pairwise(x[n2:]))

    a = [f(v) for v in a]
    res = f(0)
    for i in range(0, max(len(a), 1), 8192):
        res = f(res + pairwise(a[i:i + 8192]))
    return res


def refine_trans(k0, k1, T, scores, dist):
    """refiner.Refine_trans (estimator.py:28-72)."""
    T = np.asarray(T, np.float64)[:3]
    inl = np.where(inlier_mask(k0, k1, T, dist))[0]
    s = scores[inl]; a = k0[inl]; b = k1[inl]
    s = s / np.sum(s)
    c0 = np.sum(a * s[:, None], 0); c1 = np.sum(b * s[:, None], 0)
    A = a - c0[None]; Bm = b - c1[None]
    H = (A * s[:, None]).T @ Bm                      # afterrot^T diag(w) beforerot
    U, _, VT = np.linalg.svd(H)
    R = U @ VT
    t = c0 - c1 @ R.T
    Tn = np.eye(4); Tn[:3, :3] = R; Tn[:3, 3] = t
    return Tn


def select_top_matches(scores, match_n):
    """estimator.py:415-421 / evaluator.py:58-64."""
    num = max(scores.shape[0] * match_n, 10) if match_n < 0.999 else match_n
    return np.argsort(scores)[-int(num):]


def yohoo_ransac(k0, k1, scores, Trans, ird, max_iter, RM, match_n, rng_shuffle):
    """yohoo_ransac.ransac core for one pair (estimator.py:410-439).  rng_shuffle(index) must shuffle in
    place like np.random.shuffle (the reference uses the global numpy RNG)."""
    if RM:
        Trans = Trans[select_top_matches(scores, match_n)]
    index = np.arange(Trans.shape[0])
    rng_shuffle(index)
    Tr = Trans[index[0:max_iter]]
    best, bestT, recall = 0, 0, 0
    ovs = np.zeros(Tr.shape[0])
    for i in range(Tr.shape[0]):
        ov = overlap_cal(k0, k1, Tr[i], scores, ird)
        ovs[i] = ov
        if ov > best:
            best, bestT, recall = ov, Tr[i], i
    T1 = refine_trans(k0, k1, bestT, scores, ird * 2.0)
    T2 = refine_trans(k0, k1, T1, scores, ird)
    return T2, recall, ovs


# ---- yohoc (estimator.py:113-242) -------------------------------------------------------------------
def dr_statistic(idx):
    stat = {i: [] for i in range(60)}
    for t in range(idx.shape[0]):
        stat[int(idx[t])].append(t)
    prob = []
    for i in range(60):
        if len(stat[i]) < 2:
            prob.append(0)
        else:
            n = float(len(stat[i])) / 100.0
            prob.append(n * (n - 0.01) * (n - 0.02))
    prob = np.array(prob)
    if np.sum(prob) == 0:
        return None, np.zeros(60)
    return stat, prob / np.sum(prob)


def three_pps_to_tran(k0, k1):
    c0 = np.mean(k0, 0, keepdims=True); c1 = np.mean(k1, 0, keepdims=True)
    m = (k1 - c1).T @ (k0 - c0)
    U, S, VT = np.linalg.svd(m)
    R = VT.T @ U.T
    off = c0 - (c1 @ R.T)
    return np.concatenate([R, off.T], 1)


def yohoc_ransac(k0_all, k1_all, scores, dr_index, ird, max_iter, RM, match_n, rng=np.random):
    """yohoc_ransac.ransac_once core (estimator.py:163-242); `rng` provides choice() like np.random."""
    sample_index = np.arange(k0_all.shape[0])
    if RM:
        sample_index = select_top_matches(scores, match_n)
    k0 = k0_all[sample_index]; k1 = k1_all[sample_index]
    stat, prob = dr_statistic(dr_index[sample_index])
    if np.sum(prob) < 1e-5:
        return None, 50000
    it, recall, best, bestT, execs = 0, 0, 0, np.ones(4), 0
    while it < max_iter:
        if execs > 50000:
            break
        execs += 1
        ri = rng.choice(range(60), p=prob)
        if len(stat[ri]) < 2:
            continue
        it += 1
        ids = rng.choice(np.array(stat[ri]), 3)
        T = three_pps_to_tran(k0[ids], k1[ids])
        ov = overlap_cal(k0_all, k1_all, T, scores, ird)
        if ov > best:
            best, bestT, recall = ov, T, it
    T1 = refine_trans(k0_all, k1_all, bestT, scores, ird * 2.0)
    T2 = refine_trans(k0_all, k1_all, T1, scores, ird)
    return T2, recall


# ====================================================================================================
# metrics (test/evaluator.py:50-101, utils/r_eval.py)
# ====================================================================================================
def quaternion_from_matrix(M):
    """utils/r_eval.py:5-88, isprecise=False branch."""
    M = np.asarray(M, np.float64)
    m00, m01, m02 = M[0, 0], M[0, 1], M[0, 2]
    m10, m11, m12 = M[1, 0], M[1, 1], M[1, 2]
    m20, m21, m22 = M[2, 0], M[2, 1], M[2, 2]
    Kq = np.array([[m00 - m11 - m22, 0.0, 0.0, 0.0],
                   [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
                   [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
                   [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22]])
    Kq /= 3.0
    w, V = np.linalg.eigh(Kq)
    q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        q = -q
    return q


def compute_R_diff(R_gt, R):
    """utils/r_eval.py:108-115 (degrees)."""
    eps = 1e-15
    q_gt = quaternion_from_matrix(R_gt); q = quaternion_from_matrix(R)
    q = q / (np.linalg.norm(q) + eps); q_gt = q_gt / (np.linalg.norm(q_gt) + eps)
    loss_q = np.maximum(eps, (1.0 - np.sum(q * q_gt) ** 2))
    return np.rad2deg(np.abs(np.arccos(1 - 2 * loss_q)))


def pair_inlier_ratio(k0, k1, corr, gt, tau_2):
    a = k0[corr[:, 0]]; b = k1[corr[:, 1]]
    b = b @ gt[:, :3].T + gt[:, 3:].T
    return np.mean(np.sqrt(np.sum(np.square(a - b), -1)) < tau_2)


def pre_log_text(pairs, n_pc, transforms):
    """R_pre_log (estimator.py:14-26)."""
    out = []
    for (a, b), T in zip(pairs, transforms):
        out.append(f'{int(a)}\t{int(b)}\t{n_pc}\n')
        for r in range(3):
            out.append(f'{T[r][0]}\t{T[r][1]}\t{T[r][2]}\t{T[r][3]}\n')
        out.append(f'{0.0}\t{0.0}\t{0.0}\t{1.0}\n')
    return ''.join(out)
