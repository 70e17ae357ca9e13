"""ctypes bindings of the rotation-coherence matcher's kernels (csrc/rm.hip, ot_flash.hip, linear_mfma.hip, corr_mfma.hip): part of the
`roreg_amd.hip` namespace (hip.py re-exports everything here; callers keep writing `hip.sinkhorn_batch(...)`).  Module state that callers
set on `hip` (hip.WORK) is read through the module, never copied."""
import ctypes
import os

import numpy as np
import torch

from . import hip as _core
from .hip import HipError, _check, _ptr, _stream, ensure_des2r, ensure_tables, lib, upload

__all__ = ['Cat3Rows', 'OT_COOP', 'OT_RECOMPUTE', 'Segments', '_rm_op', 'concat_rows', 'context_with_colmax', 'group_corr', 'knn_attention', 'knn_coor', 'l2_normalize_rows', 'linear', 'matrix_core_layers', 'mean_over_group', 'round5_chain_layers', 'mlp_instnorm', 'sinkhorn', 'sinkhorn_batch', 'sinkhorn_early_exit', 'sinkhorn_iteration_stats', 'topk_dot', 'value_input', 'vector_pipe_layers']


def group_corr(perm_feats, bcast_feats, perm_rows=None, bcast_rows=None, transpose=False, want_idx=False, perm_coefs=None, bcast_coefs=None):
    """cor [M,60] (and optionally the first argmax) of the generalised 60x60 group cross-correlation.  With perm_coefs / bcast_coefs =
    feat_coefs(perm_feats / bcast_feats): evaluated in the irrep domain (sum_d d^3 = 244 multiply-adds per channel instead of 3600; the
    values agree with the literal float32 evaluation to its own rounding level, ~1e-6 of |d1||d2| -- for use as a FEATURE, as the matcher's
    R_indicator is; the literal kernel remains the one whose arg-max is the contract)."""
    ensure_tables()
    M = int(perm_rows.shape[0]) if perm_rows is not None else (int(bcast_rows.shape[0]) if bcast_rows is not None else int(perm_feats.shape[0]))
    if perm_coefs is not None and not want_idx:
        ensure_des2r()
        cor = torch.empty((M, 60), dtype=torch.float32, device=perm_coefs.device)
        _check(lib().roreg_group_corr_irrep(_ptr(perm_coefs, torch.float32), _ptr(perm_rows, torch.int64), _ptr(bcast_coefs, torch.float32),
                                            _ptr(bcast_rows, torch.int64), M, 1 if transpose else 0, _ptr(cor), _stream()), 'roreg_group_corr_irrep')
        return cor
    cor = torch.empty((M, 60), dtype=torch.float32, device=perm_feats.device)
    if (MATRIX_CORE_LAYERS or CORR_MFMA) and not want_idx:      # one 60 x 32 x 60 float32 MFMA product per point + coset sums (csrc/corr_mfma.hip)
        _check(lib().roreg_group_corr_mfma(_ptr(perm_feats, torch.float32), _ptr(perm_rows, torch.int64), _ptr(bcast_feats, torch.float32),
                                           _ptr(bcast_rows, torch.int64), M, 1 if transpose else 0, _ptr(cor), _stream()), 'roreg_group_corr_mfma')
        return cor
    idx = torch.empty(M, dtype=torch.int64, device=perm_feats.device) if want_idx else None
    _check(lib().roreg_group_corr(_ptr(perm_feats, torch.float32), _ptr(perm_rows, torch.int64), _ptr(bcast_feats, torch.float32),
                                  _ptr(bcast_rows, torch.int64), M, 1 if transpose else 0, _ptr(idx), _ptr(cor), _stream()), 'roreg_group_corr')
    return (cor, idx) if want_idx else cor


class Segments:
    """Row offsets of several pairs' point lists concatenated into one tensor (the matcher's per-pair operations stay inside their
    pair; see include/roreg_hip.h).  lengths: points per pair."""

    def __init__(self, lengths):
        lengths = np.asarray(lengths, np.int64)
        self.host = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
        self.dev = upload(self.host)
        self.n = int(lengths.shape[0]); self.max = int(lengths.max()); self.min = int(lengths.min()); self.total = int(self.host[-1])


def topk_dot(A, B, k, want_val=False, segA=None, segB=None):
    """k best rows of B per row of A; with segments, inside the row's pair (indices are global rows of B)."""
    m, n = A.shape[0], B.shape[0]
    idx = torch.empty((m, k), dtype=torch.int64, device=A.device)
    val = torch.empty((m, k), dtype=torch.float32, device=A.device) if want_val else None
    wsn = lib().roreg_topk_dot_workspace_size(m, n, k)
    ws = torch.empty(wsn, dtype=torch.float32, device=A.device)
    if segA is not None:
        if segB.min < k:
            raise HipError(f'topk_dot: a pair has fewer than k={k} targets')
        seg = (_ptr(segA.dev, torch.int32), _ptr(segB.dev, torch.int32), segA.n, segA.max, segB.max)
    else:
        seg = (None, None, 1, m, n)
    _check(lib().roreg_topk_dot(_ptr(A, torch.float32), m, _ptr(B, torch.float32), n, k, _ptr(idx), _ptr(val), _ptr(ws), wsn, *seg, _stream()),
           'roreg_topk_dot')
    if _core.WORK is not None:
        pairs = float(m) * n if segA is None else float(np.sum(np.diff(segA.host).astype(np.float64) * np.diff(segB.host)))
        _core.WORK['topk_flop'] = _core.WORK.get('topk_flop', 0.0) + 2.0 * A.shape[1] * pairs
    return (idx, val) if want_val else idx


# The matcher's 1x1 layers and R_indicator: False = one float32 fmaf chain per (row, output) / the literal gathered correlation on the vector
# pipe (Match_ot.forward(): the arithmetic that keeps the log-couplings within 1e-4 of the reference's at keynum 2500 -- a different rounding,
# equally accurate, flips a top-k neighbour on that fixture); True = fp16 hi + lo MFMAs (csrc/linear_mfma.hip) and one float32 MFMA product per
# point (csrc/corr_mfma.hip): the stacked matcher, which returns matches and scores only.  Set by matrix_core_layers().
MATRIX_CORE_LAYERS = False
# ROREG_CORR_MFMA=1: R_indicator alone through csrc/corr_mfma.hip (in forward() and the stacked path alike); measurement switch, see NOTES.md round 5
CORR_MFMA = os.environ.get('ROREG_CORR_MFMA', '0') == '1'


class matrix_core_layers:
    """`with hip.matrix_core_layers():` -- linear() / mlp_instnorm() / group_corr() inside the block run on the matrix cores."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global MATRIX_CORE_LAYERS
        self.prev = MATRIX_CORE_LAYERS
        MATRIX_CORE_LAYERS = self.on
        return self

    def __exit__(self, *exc):
        global MATRIX_CORE_LAYERS
        MATRIX_CORE_LAYERS = self.prev
        return False


class vector_pipe_layers:
    """`with hip.vector_pipe_layers():` -- linear() / mlp_instnorm() evaluate their fmaf chains on the vector pipe instead of the matrix cores
    (roreg_linear_path; same bits -- tests and A/B measurements)."""

    def __enter__(self):
        self.prev = lib().roreg_linear_path(1)

    def __exit__(self, *exc):
        lib().roreg_linear_path(self.prev)
        return False


class round5_chain_layers:
    """`with hip.round5_chain_layers():` -- the matrix-core fmaf chains through round 5's kernels (one launch per convolution, no software
    pipeline, roreg_instnorm_stats for the statistics): roreg_linear_path(2); tests and A/B measurements."""

    def __enter__(self):
        self.prev = lib().roreg_linear_path(2)

    def __exit__(self, *exc):
        lib().roreg_linear_path(self.prev)
        return False


class Cat3Rows:
    """The value MLP's input rows [m * k, 96] = [pos[r] | table[idx[r]] | conf[r // k]] WITHOUT the tensor (value_input): linear() hands the three
    sources to roreg_linear_cat3, whose row staging assembles them (same chains on the same values: bitwise linear(materialise(), ...));
    on the other kernel paths (vector pipe, fp16 hi/lo layers) the rows are built once and kept."""

    def __init__(self, pos, table, conf, idx):
        self.pos, self.table, self.conf, self.idx = pos, table, conf, idx
        self.m, self.k = idx.shape
        self.shape = (self.m * self.k, 96)
        self.device = pos.device
        self._rows = None

    def materialise(self):
        if self._rows is None:
            self._rows = _rm_op(5, self.pos, torch.empty((self.m * self.k, 96), dtype=torch.float32, device=self.device), b=self.table, c=self.conf,
                                idx=self.idx, L=self.m, k=self.k)
        return self._rows


def linear(x, W, b):
    """x [L,Cin] -> [L,Cout]; W [Cout,Cin], b [Cout] device float32."""
    if isinstance(x, Cat3Rows):
        Cout = W.shape[0]
        if MATRIX_CORE_LAYERS or Cout not in (64, 32) or lib().roreg_linear_path(-1) == 1:
            x = x.materialise()
        else:
            y = torch.empty((x.shape[0], Cout), dtype=torch.float32, device=x.device)
            _check(lib().roreg_linear_cat3(_ptr(x.pos, torch.float32), _ptr(x.table, torch.float32), _ptr(x.idx, torch.int64), _ptr(x.conf, torch.float32),
                                           x.m, x.k, _ptr(W, torch.float32), _ptr(b, torch.float32), Cout, _ptr(y), _stream()), 'roreg_linear_cat3')
            return y
    L, Cin = x.shape
    Cout = W.shape[0]
    y = torch.empty((L, Cout), dtype=torch.float32, device=x.device)
    fn = lib().roreg_linear_mfma if MATRIX_CORE_LAYERS else lib().roreg_linear
    _check(fn(_ptr(x, torch.float32), L, Cin, _ptr(W, torch.float32), _ptr(b, torch.float32), Cout, _ptr(y), _stream()), 'roreg_linear')
    return y


def mlp_instnorm(x, W1, b1, W2, b2, Wr, br, eps=1e-5, seg=None):
    """mlp_2layer / Contextnorm: conv -> InstanceNorm -> ReLU -> conv, plus the residual conv.  x [L,Cin] -> [L,32].
    seg (Segments of the points; L = mult * seg.total rows): the InstanceNorm statistics are per pair.
    Default path: roreg_mlp_head (first conv + residual conv in one launch, the statistics of h from its tiles' channel sums) and
    roreg_mlp_tail; other paths / shapes: the two convs, roreg_instnorm_stats, the tail."""
    L = x.shape[0]
    C = W1.shape[0]
    n_seg = seg.n if seg is not None else 1
    mult = L // seg.total if seg is not None else 1
    if seg is not None and mult * seg.total != L:
        raise HipError('mlp_instnorm: rows are not a multiple of the segmented points')
    sg = (_ptr(seg.dev, torch.int32) if seg is not None else None, n_seg, mult)
    dev = x.device
    stats = torch.empty(n_seg * 2 * C, dtype=torch.float32, device=dev)
    h = y = None
    if not MATRIX_CORE_LAYERS and Wr.shape[0] == 32:
        cat = isinstance(x, Cat3Rows)
        h = torch.empty((L, C), dtype=torch.float32, device=dev); y = torch.empty((L, 32), dtype=torch.float32, device=dev)
        ws = torch.empty(lib().roreg_mlp_head_workspace(L, n_seg, C), dtype=torch.float64, device=dev)
        if cat:
            args = (None, _ptr(x.pos, torch.float32), _ptr(x.table, torch.float32), _ptr(x.idx, torch.int64), _ptr(x.conf, torch.float32), x.m, x.k, L, 96)
        else:
            args = (_ptr(x, torch.float32), None, None, None, None, 0, 0, L, x.shape[1])
        rc = lib().roreg_mlp_head(*args, _ptr(W1, torch.float32), _ptr(b1, torch.float32), C, _ptr(Wr, torch.float32), _ptr(br, torch.float32), _ptr(h), _ptr(y),
                                  *sg, float(eps), _ptr(stats), _ptr(ws), _stream())
        if rc == 3:                                             # a shape or path the fused head does not serve
            h = y = None
        else:
            _check(rc, 'roreg_mlp_head')
    if h is None:
        h = linear(x, W1, b1)
        ws = torch.empty(n_seg * 2 * C * 256, dtype=torch.float64, device=dev)
        _check(lib().roreg_instnorm_stats(_ptr(h), L, C, float(eps), _ptr(stats), _ptr(ws), *sg, _stream()), 'roreg_instnorm_stats')
        y = linear(x, Wr, br)
    tail = lib().roreg_mlp_tail_mfma if MATRIX_CORE_LAYERS else lib().roreg_mlp_tail
    _check(tail(_ptr(h), L, C, _ptr(stats), _ptr(W2, torch.float32), _ptr(b2, torch.float32), _ptr(y), *sg, _stream()), 'roreg_mlp_tail')
    return y


def knn_attention(qp, kp, vp, idx, k, k_is_table, v_is_table):
    m = qp.shape[0]
    x = torch.empty((m, 32), dtype=torch.float32, device=qp.device)
    _check(lib().roreg_knn_attention(_ptr(qp, torch.float32), _ptr(kp, torch.float32), _ptr(vp, torch.float32), _ptr(idx, torch.int64),
                                     1 if k_is_table else 0, 1 if v_is_table else 0, m, k, _ptr(x), _stream()), 'roreg_knn_attention')
    return x


def _rm_op(op, a, out, b=None, c=None, idx=None, L=0, k=0, C=0, ws=None):
    _check(lib().roreg_rm_elementwise(op, _ptr(a, torch.float32), _ptr(b, torch.float32), _ptr(c, torch.float32), _ptr(idx, torch.int64),
                                      L, k, C, _ptr(out, torch.float32), _ptr(ws, torch.float32), _stream()), 'roreg_rm_elementwise')
    return out


def l2_normalize_rows(x):
    L, C = x.shape
    return _rm_op(0, x, torch.empty_like(x), L=L, C=C)


def context_with_colmax(R, seg=None):
    """[R | max over the pair's points of R, broadcast]  (rot_coh_match.py:201)  [m,60] -> [m,120]."""
    m = R.shape[0]
    n_seg = seg.n if seg is not None else 1
    ws = torch.empty(n_seg * 257 * 60, dtype=torch.float32, device=R.device)
    ctx = torch.empty((m, 120), dtype=torch.float32, device=R.device)
    _check(lib().roreg_context_colmax(_ptr(R, torch.float32), m, _ptr(seg.dev, torch.int32) if seg is not None else None, n_seg,
                                      seg.max if seg is not None else m, _ptr(ctx), _ptr(ws), _stream()), 'roreg_context_colmax')
    return ctx


def knn_coor(coor, idx):
    m, k = idx.shape
    return _rm_op(3, coor, torch.empty((m * k, 3), dtype=torch.float32, device=coor.device), idx=idx, L=m, k=k)


def value_input(pos_n, fea_n_table, conf_n, idx, materialise=False):
    """-> the [m * k, 96] rows as a Cat3Rows (consumed by linear() / mlp_instnorm() without being built), or the tensor itself."""
    rows = Cat3Rows(pos_n, fea_n_table, conf_n, idx)
    return rows.materialise() if materialise else rows


def concat_rows(a, b, c=None):
    L, C = a.shape
    Cc = 0 if c is None else c.shape[1]
    return _rm_op(6, a, torch.empty((L, 2 * C + Cc), dtype=torch.float32, device=a.device), b=b, c=c, L=L, k=Cc, C=C)


def mean_over_group(eqv):
    m = eqv.shape[0]
    return _rm_op(7, eqv, torch.empty((m, 32), dtype=torch.float32, device=eqv.device), L=m)


def sinkhorn(src_final, tgt_final, alpha, iters):
    m, n = src_final.shape[0], tgt_final.shape[0]
    dev = src_final.device
    Z = torch.empty((m + 1, n + 1), dtype=torch.float32, device=dev)
    m0 = torch.empty(m, dtype=torch.int64, device=dev); m1 = torch.empty(n, dtype=torch.int64, device=dev)
    s0 = torch.empty(m, dtype=torch.float32, device=dev); s1 = torch.empty(n, dtype=torch.float32, device=dev)
    wsn = lib().roreg_sinkhorn_workspace_size(m, n)
    ws = torch.empty(wsn, dtype=torch.float32, device=dev)
    _check(lib().roreg_sinkhorn(_ptr(src_final, torch.float32), m, _ptr(tgt_final, torch.float32), n, float(alpha), int(iters), _ptr(Z),
                                _ptr(m0), _ptr(m1), _ptr(s0), _ptr(s1), _ptr(ws), wsn, _stream()), 'roreg_sinkhorn')
    return Z, m0, m1, s0, s1


# Sinkhorn iterations of the stacked matcher path: 1 = recompute the scores on the matrix cores in every pass (csrc/ot_flash.hip, default);
# ROREG_OT_RECOMPUTE=0 = read the materialised coupling matrix once per iteration (rounds 1-3; A/B switch)
OT_RECOMPUTE = os.environ.get('ROREG_OT_RECOMPUTE', '1') == '1'
# target clouds of 2560 ... 5119 points (keynum 5000): 0 = two recomputations per iteration (default: measured faster), 1 = one, two cooperating
# workgroups per strip (csrc/ot_flash.hip)
OT_COOP = os.environ.get('ROREG_OT_COOP', '0') == '1'


class sinkhorn_early_exit:
    """`with hip.sinkhorn_early_exit(False): ...` -- every pair runs all `iters` Sinkhorn iterations, like the reference's loop
    (network/rot_coh_match.py:289-292); the default (True) stops a pair at the fixed point of the float32 iteration: once an iteration's largest
    step is within 2 .. 4 float32 units in the last place, or within 8 such units and not smaller than the previous iteration's (include/roreg_hip.h,
    roreg_sinkhorn_early_exit).  The library-wide switch is restored on exit."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.before = lib().roreg_sinkhorn_early_exit(1 if self.on else 0)
        return self

    def __exit__(self, *exc):
        lib().roreg_sinkhorn_early_exit(self.before)
        return False


def sinkhorn_iteration_stats(reset=True):
    """-> (iterations run, pairs) summed over the recomputed-iteration Sinkhorn calls since the last reset (synchronises the stream)."""
    it = ctypes.c_longlong(0); pr = ctypes.c_longlong(0)
    _check(lib().roreg_sinkhorn_iteration_stats(ctypes.byref(it), ctypes.byref(pr), 1 if reset else 0, _stream()), 'roreg_sinkhorn_iteration_stats')
    return int(it.value), int(pr.value)


def sinkhorn_batch(src_final, tgt_final, seg_src, seg_tgt, alpha, iters, recompute=None, want_Z=False):
    """Sinkhorn + mutual read-out of several pairs (descriptors concatenated by seg_src / seg_tgt) ->
    (matches0 [sum m] local indices or -1, matches1 [sum n], mscores0, mscores1).  recompute: None = OT_RECOMPUTE (and OT_COOP);
    False = the materialised matrix; True = scores recomputed on the matrix cores; 'coop' = the same with cooperating workgroups for target
    clouds of 2560 ... 5119 points.  want_Z (one pair only): the log-coupling matrix [(m+1),(n+1)] is returned in front."""
    if recompute is None:
        recompute = ('coop' if OT_COOP else True) if OT_RECOMPUTE else False
    mode = 2 if recompute == 'coop' else (1 if recompute else 0)
    dev = src_final.device
    tm, tn = seg_src.total, seg_tgt.total
    m0 = torch.empty(tm, dtype=torch.int64, device=dev); m1 = torch.empty(tn, dtype=torch.int64, device=dev)
    s0 = torch.empty(tm, dtype=torch.float32, device=dev); s1 = torch.empty(tn, dtype=torch.float32, device=dev)
    consts = np.empty(4 * seg_src.n, np.float32)
    _check(lib().roreg_sinkhorn_batch_consts(seg_src.host.ctypes.data, seg_tgt.host.ctypes.data, seg_src.n, consts.ctypes.data), 'roreg_sinkhorn_batch_consts')
    cdev = upload(consts)
    wsn = lib().roreg_sinkhorn_batch3_workspace_size(seg_src.n, seg_src.max, seg_tgt.max, tm, tn, mode, 1 if want_Z else 0)
    ws = torch.empty(wsn, dtype=torch.float32, device=dev)
    Z = torch.empty((seg_src.max + 1, seg_tgt.max + 1), dtype=torch.float32, device=dev) if want_Z else None
    _check(lib().roreg_sinkhorn_batch3(_ptr(src_final, torch.float32), _ptr(tgt_final, torch.float32), _ptr(seg_src.dev, torch.int32),
                                       _ptr(seg_tgt.dev, torch.int32), seg_src.host.ctypes.data, seg_tgt.host.ctypes.data, _ptr(cdev), seg_src.n,
                                       float(alpha), int(iters), _ptr(m0), _ptr(m1), _ptr(s0), _ptr(s1), _ptr(ws), wsn, mode,
                                       _ptr(Z) if want_Z else None, _stream()),
           'roreg_sinkhorn_batch3')
    if _core.WORK is not None:
        cells = float(np.sum((np.diff(seg_src.host).astype(np.float64) + 1) * (np.diff(seg_tgt.host) + 1)))
        _core.WORK['sinkhorn_bytes'] = _core.WORK.get('sinkhorn_bytes', 0.0) + 4.0 * cells * int(iters)
        _core.WORK['sinkhorn_cells'] = _core.WORK.get('sinkhorn_cells', 0.0) + cells * int(iters)
        _core.WORK['sinkhorn_pairs'] = _core.WORK.get('sinkhorn_pairs', 0) + seg_src.n
        _core.WORK['sinkhorn_recompute'] = bool(recompute)
    return (Z, m0, m1, s0, s1) if want_Z else (m0, m1, s0, s1)
