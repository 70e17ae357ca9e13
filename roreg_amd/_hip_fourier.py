"""ctypes bindings of the group-Fourier (irrep-domain) path of the group convolutions and of the split-operand dense / conv layers
(csrc/fourier.hip, group_conv.hip): part of the `roreg_amd.hip` namespace (hip.py re-exports everything here).  Module state that callers
set on `hip` (hip.PROFILE) is read through the module, never copied."""
import os
from ctypes import c_void_p

import numpy as np
import torch

from . import hip as _core
from .hip import HipError, _check, _feat, _ptr, _stream, lib

__all__ = ['DenseSplitLayer', 'IRREP_DIMS', 'IRREP_OFFSETS', '_bf16_split3', '_keypoint_of_columns', '_ptr_array', '_res_ptr', '_tile_cache', 'bf16_split3_pack', 'bound_exp', 'coef_pitch', 'coef_size', 'coef_views', 'dense_split', 'ensure_fourier', 'f16_scale_exp', 'f16_split2_pack', 'ft_nonlin', 'ft_nonlin_packed', 'gemm_persistent', 'group_conv_split_pack', 'irrep_gemm', 'next_bound', 'next_bound_spatial', 'pack_coefs_f16x2', 'row_bound', 'unpack_coefs_f16x2', 'words_to_planes']


_fourier_ready = False
IRREP_DIMS = (1, 3, 3, 4, 5)
IRREP_OFFSETS = (0, 1, 10, 19, 35, 60)


def ensure_fourier():
    global _fourier_ready
    if not _fourier_ready:
        from .fourier import group_fourier
        F = np.ascontiguousarray(group_fourier().F, np.float32)
        _check(lib().roreg_set_fourier_tables(F.ctypes.data), 'roreg_set_fourier_tables')
        _fourier_ready = True


def _ptr_array(views):
    arr = (c_void_p * 5)(*[c_void_p(v.data_ptr()) for v in views])
    return arr


def coef_pitch(B):
    """Keypoint pitch of the coefficient buffers: B rounded up to the 32-keypoint tile of ft_nonlin (pad keypoints hold zeros)."""
    return (int(B) + 31) // 32 * 32


def coef_size(C, B):
    return 60 * C * coef_pitch(B)


def coef_views(buf, C, B):
    """Five per-irrep GEMM operands [d*C, d*Bp] of a flat coefficient buffer of 60*C*Bp floats (Bp = coef_pitch(B))."""
    Bp = coef_pitch(B)
    return [buf[IRREP_OFFSETS[r] * C * Bp:IRREP_OFFSETS[r + 1] * C * Bp].view(IRREP_DIMS[r] * C, IRREP_DIMS[r] * Bp) for r in range(5)]


_tile_cache = {}


def irrep_gemm(X_buf, Wpacks, C, O, B, split=None, add=None, f16x2=None, x_bound=None, next_bound=None, x_planes=False):
    """coefficients [60*C*Bp] -> [60*O*Bp] through the five per-irrep GEMMs (Bp = coef_pitch(B): the GEMMs run on the padded width).
    split: the five 3xbf16-split weight tensors (f32-accurate GEMM on the bf16 matrix cores) or None for the f32-input MFMA kernel.
    add: optional coefficient buffer [60*O*Bp] summed onto the result in the epilogue (residual short cut in the irrep domain).
    f16x2 = (five fp16x2 weight tensors, w_exp): X_buf holds the fp16 hi/lo words ft_nonlin(split='f16x2', out_bound=x_bound) wrote under the
    per-keypoint bound x_bound [Bp]; next_bound = (u [O], v [O]) (NextBound of the following nonlinearity) additionally returns the
    per-keypoint bound [Bp] of the NEXT transform's coefficients: -> (out, bound).
    x_planes: X_buf is in the half-block layout (ft_nonlin(..., planes=True): per 32-column block 32 fp16 hi values, then 32 lo values, columns in the order 0, 16, 1, 17, ...): the activations reach LDS by LDS-DMA; O % 256 == 0 only.  True: the kernel hip.MFMA16 selects
    (default: 16x16x32 MFMAs); the integers 1 / 2 name the 32x32x16 / 16x16x32 kernel (tests)."""
    Bp = coef_pitch(B)
    if X_buf.numel() != 60 * C * Bp or (add is not None and add.numel() != 60 * O * Bp):
        raise HipError(f'irrep_gemm: coefficient buffers must hold 60*C*{Bp} floats (B={B} padded to the 32-keypoint pitch)')
    out = torch.empty(60 * O * Bp, dtype=torch.float32, device=X_buf.device)
    tile_m = 256 if (f16x2 is not None and O % 256 == 0 and not os.environ.get('ROREG_TILE_M128')) else 128          # 256-row tiles = 8-wave workgroups (fp16 x 2 kernel)
    key = (O, Bp, tile_m)
    t = _tile_cache.get(key)
    if t is None:
        n = lib().roreg_irrep_gemm_tiles_m(O, Bp, tile_m, None)
        host = np.empty((n, 3), np.int32)
        lib().roreg_irrep_gemm_tiles_m(O, Bp, tile_m, host.ctypes.data)
        t = torch.from_numpy(host).cuda()
        _tile_cache[key] = t
    xv = coef_views(X_buf, C, B); ov = coef_views(out, O, B)
    av = _ptr_array(coef_views(add, O, B)) if add is not None else None
    if _core.PROFILE is not None:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    bound_out = None
    if f16x2 is not None:
        wl, w_exp = f16x2
        if x_bound is None or x_bound.numel() != Bp:
            raise HipError(f'irrep_gemm: the fp16 x 2 GEMM needs x_bound with one value per (padded) keypoint ({Bp})')
        nu = nv = None
        if next_bound is not None:
            nu, nv = next_bound
            if nu.numel() != O or nv.numel() != O:
                raise HipError('irrep_gemm: next_bound must hold one (u, v) per output channel')
            bound_out = torch.zeros(Bp, dtype=torch.float32, device=X_buf.device)
        _check(lib().roreg_irrep_gemm_f16x2(_ptr_array(xv), _ptr_array(ov), av, _ptr_array(wl), _ptr(x_bound, torch.float32), int(w_exp),
                                            _ptr(nu, torch.float32), _ptr(nv, torch.float32), _ptr(bound_out), C, O, Bp,
                                            _ptr(t, torch.int32), int(t.shape[0]), tile_m, (int(x_planes) if x_planes in (1, 2) and x_planes is not True else (2 if _core.MFMA16 else 1)) if x_planes else 0, _stream()), 'roreg_irrep_gemm_f16x2')
    elif split is not None:
        _check(lib().roreg_irrep_gemm_split(_ptr_array(xv), _ptr_array(ov), av, _ptr_array(split), C, O, Bp, _ptr(t, torch.int32), int(t.shape[0]),
                                            _stream()), 'roreg_irrep_gemm_split')
    else:
        _check(lib().roreg_irrep_gemm(_ptr_array(xv), _ptr_array(ov), av, _ptr_array(Wpacks), C, O, Bp, _ptr(t, torch.int32), int(t.shape[0]), _stream()),
               'roreg_irrep_gemm')
    if _core.PROFILE is not None:
        e1.record(); _core.PROFILE.append((('irrep_gemm_f16x2' if f16x2 is not None else 'irrep_gemm_split' if split is not None else 'irrep_gemm', Bp, C, O), e0, e1))
    return (out, bound_out) if next_bound is not None else out


class gemm_persistent:
    """`with hip.gemm_persistent(True | False):` -- how the 16x16x32 LDS-DMA GEMM is launched inside the block (roreg_gemm_persistent): persistent
    workgroups that claim tiles, or one workgroup per tile.  Same bits either way -- tests and A/B measurements."""

    def __init__(self, on=True):
        self.on = int(on)                                     # 0 / False: one workgroup per tile; 1 / True: persistent; 2: half tiles, two workgroups per CU

    def __enter__(self):
        self.prev = lib().roreg_gemm_persistent(self.on)

    def __exit__(self, *exc):
        lib().roreg_gemm_persistent(self.prev)
        return False


def row_bound(x, bn=None):
    """x [B,C,60] group-domain tensor -> float32 [Bp]: sqrt(60) * max_{c,g} |act(x[b])| (act = ReLU(BN(.)) with bn = (scale, shift), else the
    identity), a bound on every coefficient of FT(act(x[b])); 0 for the pad keypoints.  The x_bound of a layer fed from the group domain."""
    B, C = int(x.shape[0]), int(x.shape[1])
    out = torch.empty(coef_pitch(B), dtype=torch.float32, device=x.device)
    scale, shift = bn if bn is not None else (None, None)
    xp, bf = _feat(x)
    _check(lib().roreg_row_bound(xp, bf, _ptr(scale), _ptr(shift), _ptr(out), B, C, _stream()), 'roreg_row_bound')
    return out


def _keypoint_of_columns(d, Bp):
    """keypoint index of every GEMM column of an irrep of dimension d (columns are blocked by 32 keypoints)."""
    n = torch.arange(d * Bp, device='cuda')
    return ((n // 32) // d) * 32 + (n % 32)


def bound_exp(bound):
    """e with bound * 2^e < 2^14, as the kernels derive it (tensor in, int32 tensor out)."""
    _, ex = torch.frexp(bound.float())
    e = torch.where((bound > 0) & torch.isfinite(bound), 14 - ex, torch.zeros_like(ex))
    return e.clamp(-100, 100)


def pack_coefs_f16x2(X_buf, C, B, bound=None):
    """float32 coefficient buffer -> (fp16 hi/lo words in the layout roreg_ft_nonlin(split=2) writes, per-keypoint bound [Bp]).
    Test / tooling helper (torch ops): the product path gets its split operands from ft_nonlin directly."""
    Bp = coef_pitch(B)
    views = coef_views(X_buf, C, B)
    if bound is None:
        bound = torch.zeros(Bp, dtype=torch.float32, device=X_buf.device)
        for r, v in enumerate(views):
            kp = _keypoint_of_columns(IRREP_DIMS[r], Bp)
            bound.scatter_reduce_(0, kp, v.abs().amax(0), 'amax')
    e = bound_exp(bound)
    out = torch.empty_like(X_buf)
    for r, (v, o) in enumerate(zip(views, coef_views(out, C, B))):
        kp = _keypoint_of_columns(IRREP_DIMS[r], Bp)
        y = torch.ldexp(v, e[kp][None, :])
        hi = y.half()
        lo = (y - hi.float()).half()
        w = (hi.view(torch.int16).to(torch.int32) & 0xffff) | (lo.view(torch.int16).to(torch.int32) << 16)
        o.copy_(w.view(torch.float32))
    return out, bound


def words_to_planes(X_words, C, B):
    """The fp16 hi/lo word layout -> the half-block layout of ft_nonlin(planes=True) (test helper, torch ops): every 32-column block of a row
    holds its 32 hi values (16 bit each, in the column order 0, 16, 1, 17, ...: word w pairs columns w and w + 16), then its 32 lo values in
    the same order, in the 128 bytes the 32 words occupied."""
    w = X_words.contiguous().view(torch.int32).view(-1, 2, 16)                        # [block][column // 16][column % 16]
    hi = (w & 0xffff).to(torch.int16).transpose(1, 2); lo = (w >> 16).to(torch.int16).transpose(1, 2)         # [block][w][pair member]
    return torch.cat([hi.reshape(-1, 32), lo.reshape(-1, 32)], 1).contiguous().view(torch.float32).view(-1)


def unpack_coefs_f16x2(X_words, bound, C, B):
    """Inverse of the split: fp16 hi/lo words + per-keypoint bound -> float32 coefficients (hi + lo) * 2^-e.  Test helper."""
    Bp = coef_pitch(B)
    e = bound_exp(bound)
    out = torch.empty_like(X_words)
    for r, (v, o) in enumerate(zip(coef_views(X_words, C, B), coef_views(out, C, B))):
        kp = _keypoint_of_columns(IRREP_DIMS[r], Bp)
        w = v.view(torch.int32)
        hi = (w & 0xffff).to(torch.int16).view(torch.float16).float()
        lo = (w >> 16).to(torch.int16).view(torch.float16).float()
        o.copy_(torch.ldexp(hi + lo, -e[kp][None, :]))
    return out


def next_bound_spatial(bn, bias):
    """(u, v) device float32 [O] for a GEMM epilogue whose output goes back to the GROUP domain through ReLU(scale_o (IFT(T)_o + bias_o) + shift_o)
    (roreg_ft_nonlin_packed): |IFT(T)(g)| <= sqrt(60) max_q |T_q|, so |value| <= max_{o,q} (sqrt(60) |scale_o| |T_oq| + |scale_o| |bias_o| + |shift_o|);
    the 2^-9 margin covers the float32 rounding of the transform."""
    scale, shift = bn
    sc = scale.detach().double().abs().cpu(); sh = shift.detach().double().abs().cpu()
    b = bias.detach().double().cpu()
    k = 1.0 + 2.0 ** -9
    u = (np.sqrt(60.0) * k) * sc
    v = k * (sc * b.abs() + sh) + 1e-30
    return u.float().cuda().contiguous(), v.float().cuda().contiguous()


def ft_nonlin_packed(B, C, coef_in, bias, bn, out_bound, g_map=None, Lout=60, Lvalid=60, raw_g=None):
    """IFT + bias + BatchNorm + ReLU -> (words int32 [B,C,Lout] = fp16 hi | lo << 16 under the block scale of out_bound, raw [B,C] float32 = the
    pre-activation value of group element raw_g or None): the operand of hip.group_conv_packed."""
    ensure_fourier()
    dev = coef_in.device
    if coef_in.numel() != coef_size(C, B) or out_bound.numel() != coef_pitch(B):
        raise HipError('ft_nonlin_packed: coef_in / out_bound sizes')
    words = torch.empty((B, C, Lout if g_map is not None else 60), dtype=torch.int32, device=dev)
    raw = torch.empty((B, C), dtype=torch.float32, device=dev) if raw_g is not None else None
    scale, shift = bn if bn is not None else (None, None)
    _check(lib().roreg_ft_nonlin_packed(_ptr(coef_in, torch.float32), _ptr(bias), _ptr(scale), _ptr(shift), _ptr(words), _ptr(g_map, torch.int32),
                                        int(Lout), int(Lvalid), B, C, _ptr(out_bound, torch.float32), _ptr(raw), int(raw_g or 0), _stream()),
           'roreg_ft_nonlin_packed')
    return words, raw


def next_bound(bn, bias, bias2=None):
    """(u, v) device float32 [O] of the bound a GEMM epilogue propagates to the next transform (include/roreg_hip.h, roreg_irrep_gemm_f16x2):
    the next nonlinearity is x = ReLU(scale_o (IFT(T)_o + bias_o) + shift_o); with an orthonormal 60 x 60 transform
    |FT(x)_q| <= sqrt(60) max_g |x(g)| and |IFT(T)(g)| <= sqrt(60) max_q |T_q|, so  |FT(x)| <= max_{o,q} (60 |scale_o| |T_oq| + sqrt(60) (|scale_o| |bias_o| + |shift_o|)).
    A 2^-9 margin covers the float32 rounding of the transforms themselves."""
    scale, shift = bn
    sc = scale.detach().double().abs().cpu(); sh = shift.detach().double().abs().cpu()
    b = bias.detach().double().cpu() + (bias2.detach().double().cpu() if bias2 is not None else 0.0)
    k = 1.0 + 2.0 ** -9
    u = (60.0 * k) * sc
    v = (np.sqrt(60.0) * k) * (sc * b.abs() + sh) + 1e-30
    return u.float().cuda().contiguous(), v.float().cuda().contiguous()


def ft_nonlin(B, C, coef_in=None, x_spatial=None, bias=None, bias2=None, bn=None, resid_spatial=None, spatial_out=False,
              g_map=None, Lout=60, Lvalid=60, split=False, out_bound=None, want_rowmax=False, planes=False):
    """split='f16x2' with coefficient output: out_bound [Bp] (row_bound() or the producing GEMM's propagated bound) is required and the
    result holds fp16 hi/lo words for irrep_gemm(f16x2=...) instead of floats.  want_rowmax (group-domain output): also return the
    per-keypoint max |out[b]| [B] (the block scale of the fp16 x 2 convolution that follows)."""
    ensure_fourier()
    dev = (coef_in if coef_in is not None else x_spatial).device
    if spatial_out:
        out = torch.empty((B, C, Lout if g_map is not None else 60), dtype=torch.float32, device=dev); xout = None; osp = _ptr(out)
    else:
        out = torch.empty(coef_size(C, B), dtype=torch.float32, device=dev); xout = _ptr(out); osp = None
    scale, shift = bn if bn is not None else (None, None)
    if want_rowmax and not spatial_out:
        raise HipError('ft_nonlin: want_rowmax goes with spatial_out')
    amax = torch.zeros(B, dtype=torch.float32, device=dev) if want_rowmax else None
    if coef_in is not None and coef_in.numel() != coef_size(C, B):
        raise HipError(f'ft_nonlin: coef_in must hold 60*C*{coef_pitch(B)} floats')
    if out_bound is not None and out_bound.numel() != coef_pitch(B):
        raise HipError(f'ft_nonlin: out_bound must hold one value per (padded) keypoint ({coef_pitch(B)})')
    xs, bf_x = _feat(x_spatial); rs, bf_r = _feat(resid_spatial)
    if x_spatial is not None and resid_spatial is not None and bf_x != bf_r:
        raise HipError('ft_nonlin: x_spatial and resid_spatial must share one dtype')
    _check(lib().roreg_ft_nonlin(_ptr(coef_in, torch.float32), xs, _ptr(bias),
                                 _ptr(bias2), _ptr(scale), _ptr(shift), rs, xout, osp,
                                 _ptr(g_map, torch.int32), int(Lout), int(Lvalid), B, C, 2 if split == 'f16x2' else (1 if split else 0),
                                 _ptr(out_bound, torch.float32), _ptr(amax), bf_x or bf_r, 1 if planes else 0, _stream()), 'roreg_ft_nonlin')
    return (out, amax) if want_rowmax else out


def _bf16_split3(x):
    """float32 array -> three uint16 arrays of bf16 bits: round-to-nearest-even pieces, each taken from the exact float32 remainder."""
    out = []
    rem = np.ascontiguousarray(x, np.float32)
    for _ in range(3):
        u = rem.view(np.uint32)
        r = ((u >> 16) & 1) + 0x7fff
        hi = ((u + r) >> 16).astype(np.uint16)
        out.append(hi)
        rem = rem - (hi.astype(np.uint32) << 16).view(np.float32)
    return out


def f16_scale_exp(absmax):
    """e with absmax * 2^e in [2^13, 2^14): the block-scaling exponent of the fp16 x 2 operand split."""
    return 14 - int(np.frexp(float(absmax))[1]) if absmax > 0 else 0


def f16_split2_pack(Wm, w_exp):
    """Wm float32 [Mpad, K] (K % 16 == 0) -> int16 device tensor [2][K/16][2][Mpad][8] of fp16 bits: hi = fp16(w * 2^w_exp),
    lo = fp16(w * 2^w_exp - hi), in the fragment order of irrep_gemm_split_kernel<NP=2>."""
    Ws = np.ldexp(np.ascontiguousarray(Wm, np.float32), w_exp).astype(np.float32)
    Mpad, K = Ws.shape
    hi = Ws.astype(np.float16)
    lo = (Ws - hi.astype(np.float32)).astype(np.float16)
    out = np.empty((2, K // 16, 2, Mpad, 8), np.uint16)
    for sp, part in enumerate((hi, lo)):
        out[sp] = part.view(np.uint16).reshape(Mpad, K // 16, 2, 8).transpose(1, 2, 0, 3)
    return torch.from_numpy(out.view(np.int16)).cuda()


def bf16_split3_pack(Wm):
    """Wm float32 [Mpad, K] (K % 16 == 0) -> int16 device tensor [3][K/16][2][Mpad][8]: the three bf16 pieces of every weight in the
    fragment order of irrep_gemm_split_kernel."""
    Wm = np.ascontiguousarray(Wm, np.float32)
    Mpad, K = Wm.shape
    out = np.empty((3, K // 16, 2, Mpad, 8), np.uint16)
    for sp, bits in enumerate(_bf16_split3(Wm)):
        out[sp] = bits.reshape(Mpad, K // 16, 2, 8).transpose(1, 2, 0, 3)
    return torch.from_numpy(out.view(np.int16)).cuda()


class DenseSplitLayer:
    """out[b][o] = bias[o] + sum_k W[o][k] * act_k(x[b][k]) (+ residual) in kernel-ready form: W float32 [O,K] (K % 16 == 0) as the
    three bf16 planes [3][K/16][2][round_up(O,256)][8]; act = BatchNorm(eval)+ReLU given as per-k scale/shift, or None."""

    def __init__(self, W, bias, scale=None, shift=None):
        Wn = np.ascontiguousarray(W, np.float32)
        self.O, self.K = Wn.shape
        Opad = (self.O + 255) // 256 * 256
        Wp = np.zeros((Opad, self.K), np.float32); Wp[:self.O] = Wn
        out = np.empty((3, self.K // 16, 2, Opad, 8), np.uint16)
        for sp, bits in enumerate(_bf16_split3(Wp)):
            out[sp] = bits.reshape(Opad, self.K // 16, 2, 8).transpose(1, 2, 0, 3)
        self.ws = torch.from_numpy(out.view(np.int16)).cuda()
        # fp16 x 2 planes (hi, lo) under the power-of-two scale 2^w_exp
        self.w_exp = f16_scale_exp(float(np.abs(Wp).max()))
        Ws = np.ldexp(Wp, self.w_exp).astype(np.float32)
        hi = Ws.astype(np.float16); lo = (Ws - hi.astype(np.float32)).astype(np.float16)
        out2 = np.empty((2, self.K // 16, 2, Opad, 8), np.uint16)
        for sp, part in enumerate((hi, lo)):
            out2[sp] = part.view(np.uint16).reshape(Opad, self.K // 16, 2, 8).transpose(1, 2, 0, 3)
        self.ws2 = torch.from_numpy(out2.view(np.int16)).cuda()
        self.act_smax = float(np.abs(scale).max()) if scale is not None else 1.0
        self.act_tmax = float(np.abs(shift).max()) if shift is not None else 0.0
        self.bias = torch.from_numpy(np.ascontiguousarray(bias, np.float32)).cuda()
        self.scale = torch.from_numpy(np.ascontiguousarray(scale, np.float32)).cuda() if scale is not None else None
        self.shift = torch.from_numpy(np.ascontiguousarray(shift, np.float32)).cuda() if shift is not None else None


def _res_ptr(t):
    """data pointer of a residual operand that may be a strided VIEW (its first element is what the kernel indexes from)"""
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda:
        raise HipError('dense_split: residual must be a float32 device tensor')
    return c_void_p(t.data_ptr())


def dense_split(x, layer, residual=None, in_rowmax=None, want_rowmax=False, residual_stride=1):
    """x [B, K] float32 (device, contiguous) -> [B, O].  residual: element (b, o) at residual.flat[(b*O + o) * residual_stride] (pass a
    [B, O, L] tensor's column c as residual=t[:, :, c:] -- a view, no copy -- with residual_stride=L).  in_rowmax (device float32 [B], the tracked max |x[b]| per row): use the fp16 x 2
    kernel (per-row block scale); want_rowmax: also return the tracked per-row max |out[b]| (device float32 [B]) for the next layer."""
    B, K = x.shape
    if K != layer.K:
        raise HipError(f'dense_split: K mismatch ({K} vs {layer.K})')
    out = torch.empty((B, layer.O), dtype=torch.float32, device=x.device)
    if in_rowmax is not None:
        if in_rowmax.numel() != B:
            raise HipError(f'dense_split: in_rowmax must hold one value per row ({B}), got {in_rowmax.numel()}')
        amax = torch.zeros(B, dtype=torch.float32, device=x.device) if want_rowmax else None
        _check(lib().roreg_dense_f16x2(_ptr(x, torch.float32), _ptr(layer.ws2), layer.w_exp, _ptr(layer.bias), _ptr(layer.scale), _ptr(layer.shift),
                                       layer.act_smax, layer.act_tmax, _ptr(in_rowmax, torch.float32), _res_ptr(residual), int(residual_stride), _ptr(out),
                                       _ptr(amax), B, K, layer.O, _stream()), 'roreg_dense_f16x2')
        return (out, amax) if want_rowmax else out
    _check(lib().roreg_dense_split(_ptr(x, torch.float32), _ptr(layer.ws), _ptr(layer.bias), _ptr(layer.scale), _ptr(layer.shift),
                                   _res_ptr(residual), int(residual_stride), _ptr(out), B, K, layer.O, _stream()), 'roreg_dense_split')
    return out


def group_conv_split_pack(W):
    """W [Cout,Cin,1,KS] / [Cout,Cin,KS] float32 -> int16 device tensor [3][KS][Cin/16][2][Cout][8] (group_conv_split_kernel's
    fragment order: piece p of W[o, 16*(c/16) + 8*h + e, k])."""
    Wn = W.detach().to('cpu', torch.float32).contiguous().numpy()
    Cout, Cin = Wn.shape[0], Wn.shape[1]
    KS = int(np.prod(Wn.shape[2:]))
    Wn = Wn.reshape(Cout, Cin // 16, 2, 8, KS)
    out = np.empty((3, KS, Cin // 16, 2, Cout, 8), np.uint16)
    for sp, bits in enumerate(_bf16_split3(Wn)):
        out[sp] = bits.transpose(4, 1, 2, 0, 3)
    return torch.from_numpy(out.view(np.int16)).cuda()
