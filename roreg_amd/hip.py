"""ctypes binding of libroreg_hip.so (C-ABI: include/roreg_hip.h).

PyTorch is plumbing only here: it owns device memory and the HIP stream; every call below hands raw device
pointers + the current stream handle to the C-ABI.  There is NO fallback: if the library is missing or a call
fails, an exception is raised (the product path must never silently run on the CPU).
"""
import ctypes
import os
import threading
from ctypes import c_int, c_void_p, c_size_t, c_double, c_float, c_char_p

import numpy as np
import torch

from .group import tables

# ROREG_HIP_LIB: another build of the same library (A/B measurements of kernel variants, tools/gemm_ab.sh); default: the in-tree build
_LIB_PATH = os.environ.get('ROREG_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libroreg_hip.so')

# How the irrep-domain GEMMs and transforms feed the matrix cores:
#   'f16x2' (default): every f32 operand as hi + lo fp16 with power-of-two block scaling (22 significant bits), products hi.hi + hi.lo +
#                      lo.hi, f32 accumulate: measured error <= the f32-input kernel's, 5.3x fewer matrix-core cycles than 'f32';
#   'bf16x3'         : every f32 operand as three bf16 pieces (24 bits), six cross products, f32 accumulate; 2.67x fewer cycles;
#   'f32'            : f32-input MFMA (bitwise an fmaf chain).
# Environment override for drop-in runs of Test.py: ROREG_GEMM=f32 | bf16x3 | f16x2.
GEMM_MODES = ('f16x2', 'bf16x3', 'f32')
GEMM_MODE = os.environ.get('ROREG_GEMM', 'f16x2')
if GEMM_MODE == 'split':
    GEMM_MODE = 'bf16x3'
if GEMM_MODE not in GEMM_MODES:
    raise ValueError(f"ROREG_GEMM must be one of {GEMM_MODES}, got {GEMM_MODE!r}")
# The fp16 x 2 GEMMs with 256-row tiles and a transform as producer (the extractor's two big layers, ET's Conv_init) take their activations in
# half-block layout by LDS-DMA (csrc/fourier.hip irrep_gemm_xdma_kernel); ROREG_GEMM_XDMA=0: the word layout + register staging (A/B switch,
# bitwise the same results)
XDMA = os.environ.get('ROREG_GEMM_XDMA', '1') == '1'
# ... and with v_mfma_f32_16x16x32_f16 (K = 32 per step; ROREG_GEMM_MFMA16=0: the 32x32x16 LDS-DMA kernel, bitwise the register-staged one)
MFMA16 = os.environ.get('ROREG_GEMM_MFMA16', '1') == '1'


def use_planes(O):
    """Whether the fp16 x 2 GEMM with O output channels takes its activations in the half-block layout (LDS-DMA kernel): decided HERE for
    both ft_nonlin(planes=) and irrep_gemm(x_planes=) -- the kernel needs the 256-row tile (O % 256 == 0, not ROREG_TILE_M128)."""
    return bool(XDMA and O % 256 == 0 and not os.environ.get('ROREG_TILE_M128'))
_lib = None
_tables_uploaded = False

# the C-ABI as data: version, ctypes prototypes and the numpy layouts of the task structs (roreg_amd/_abi.py)
from ._abi import ABI_VERSION, PROTOTYPES, _P, _LT_TASK, _RANSAC_TASK, _MATCH_TASK, _GATHER_TASK


class HipError(RuntimeError):
    pass


def lib():
    """Load libroreg_hip.so (once).  Raises if it has not been built: run `python -c "import __graft_entry__ as g; g.build()"`."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HipError(f'{_LIB_PATH} is missing -- build it with `make -C roreg_amd/csrc` '
                           f'(or __graft_entry__.build()); there is no CPU fallback')
        L = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = L.roreg_abi_version()
        if got != ABI_VERSION:                 # a stale or foreign build: its entry points take other argument lists
            raise HipError(f'{_LIB_PATH} reports C-ABI version {got}, this binding was written against {ABI_VERSION} '
                           f'(include/roreg_hip.h ROREG_ABI_VERSION): rebuild with `make -C roreg_amd/csrc`')
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise HipError(f'{what} failed ({rc}): {lib().roreg_last_error().decode()}')


def ensure_tables():
    """Upload the icosahedral tables to the device (once per process)."""
    global _tables_uploaded
    if not _tables_uploaded:
        T = tables()
        P = np.ascontiguousarray(T.P, np.int32)
        Nei = np.ascontiguousarray(T.Nei, np.int32)
        R = np.ascontiguousarray(T.R, np.float64)
        _check(lib().roreg_set_group_tables(P.ctypes.data, Nei.ctypes.data, R.ctypes.data), 'roreg_set_group_tables')
        _tables_uploaded = True


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class PinnedPool:
    """Pinned host buffers kept across scenes.  A pinned allocation while kernels are in flight costs 5-90 ms on this stack (the launching thread
    sat in `torch.empty(pin_memory=True)` / `.pin_memory()` for ~100 ms of a 600 ms scene, tools/probe/dropin_timeline.py --sample), so the
    file loader, the stage-file writer and the staging ring take their buffers here and hand them back when the copy AND its consumer are done
    (the caller's responsibility: release() only after the event behind the buffer's last copy has completed)."""

    def __init__(self, max_bytes=None):
        self.lock = threading.Lock()
        self.free = {}                                           # size class -> [uint8 pinned tensors]
        self.held = 0
        self.max_bytes = int(os.environ.get('ROREG_PINNED_POOL_GB', 12)) << 30 if max_bytes is None else max_bytes

    @staticmethod
    def size_class(nbytes):
        if nbytes <= 4096:
            return 4096
        step = 1 << max(int(nbytes - 1).bit_length() - 4, 0)     # eight classes per octave: <= 12.5 % unused, and sizes that drift from scene to
        return -(-nbytes // step) * step                         # scene (match lists, hypothesis tables) land in a class the pool already holds

    def acquire(self, nbytes):
        c = self.size_class(max(int(nbytes), 1))
        with self.lock:
            got = self.free.get(c)
            if got:
                self.held -= c
                return got.pop()
        return torch.empty(c, dtype=torch.uint8, pin_memory=True)

    def release(self, buf):
        c = int(buf.shape[0])
        with self.lock:
            if self.held + c <= self.max_bytes:
                self.free.setdefault(c, []).append(buf)
                self.held += c


pinned_pool = PinnedPool()
_named_streams = {}


def named_stream(name, priority=0):
    """One persistent side stream per purpose ('loader', 'writer', ...).  torch's device allocator caches blocks per stream: a NEW stream per scene
    for the loader's uploads would leave each scene's 2.3 GB of input blocks cached for a stream that never comes back."""
    st = _named_streams.get(name)
    if st is None:
        st = _named_streams[name] = torch.cuda.Stream(priority=priority)
    return st


class _StagingRing:
    """Host -> device uploads that do not block the host: a ring of persistent pinned buffers, each guarded by an event recorded behind
    its last copy.  (A pageable `.cuda()` waits for everything queued on the stream before it copies -- in the middle of a scene that is
    a full synchronisation after which the host prepares the next launches with the GPU idle; `pin_memory()` per call costs more than
    that.)  A slot is reused only after its event has completed."""

    def __init__(self, slots=64):          # (a slot is reused only when its copy has run: with 8 the host stalled in upload() behind ~300 ms of queued kernels)
        self.bufs = [None] * slots
        self.events = [None] * slots
        self.next = 0

    def upload(self, array):
        a = np.ascontiguousarray(array)
        nbytes = a.nbytes
        if nbytes == 0:
            return torch.from_numpy(a).cuda()
        k = self.next
        self.next = (k + 1) % len(self.bufs)
        if self.events[k] is not None:
            self.events[k].synchronize()
        if self.bufs[k] is None or self.bufs[k].shape[0] < nbytes or self.bufs[k].shape[0] > 4 * PinnedPool.size_class(nbytes):
            # the slot's buffer goes back to the pool and one of this upload's size class comes out of it: the scene's few large arrays (36 MB of
            # sampled rows) and its many small ones rotate through the slots, and no slot pins a new buffer for a size the process has seen
            if self.bufs[k] is not None:
                pinned_pool.release(self.bufs[k])
            self.bufs[k] = pinned_pool.acquire(nbytes)
        host = self.bufs[k][:nbytes]
        host.numpy()[:] = a.reshape(-1).view(np.uint8)
        dev = host.cuda(non_blocking=True).view(torch.from_numpy(a[:0].reshape(-1)).dtype).reshape(a.shape)
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        self.events[k].record()
        return dev


_staging = threading.local()          # one ring per host thread (a thread drives one stream)


def upload(array):
    """numpy array -> device tensor (same shape and dtype) through the staging ring: does not wait for the kernels queued on the stream."""
    ring = getattr(_staging, 'ring', None)
    if ring is None:
        ring = _staging.ring = _StagingRing()
    return ring.upload(array)


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError('expected a device tensor (the HIP path has no host fallback)')
    if not t.is_contiguous():
        raise HipError('expected a contiguous tensor')
    if dtype is not None and t.dtype != dtype:
        raise HipError(f'expected dtype {dtype}, got {t.dtype}')
    return c_void_p(t.data_ptr())


def _feat(t):
    """Pointer + 'is bfloat16' flag of a group-feature tensor [*,32,60]: float32, or bfloat16 (BASELINE config 5: descriptors stored and
    streamed as bf16, every consumer accumulates in float32)."""
    if t is None:
        return None, 0
    if t.dtype == torch.bfloat16:
        return _ptr(t, torch.bfloat16), 1
    return _ptr(t, torch.float32), 0


# ----------------------------------------------------------------------------------------------------
# group convolution
# ----------------------------------------------------------------------------------------------------
def pack_conv_weights(W):
    """W: float32 [Cout,Cin,1,KS] or [Cout,Cin,KS] (host or device) -> packed float32 device tensor."""
    Wn = W.detach().to('cpu', torch.float32).contiguous().numpy()
    Cout, Cin = Wn.shape[0], Wn.shape[1]
    KS = int(np.prod(Wn.shape[2:]))
    Wn = np.ascontiguousarray(Wn.reshape(Cout, Cin, KS))
    n = lib().roreg_group_conv_packed_size(Cin, Cout, KS)
    out = np.empty(n, np.float32)
    _check(lib().roreg_group_conv_pack_weights(Wn.ctypes.data, Cin, Cout, KS, out.ctypes.data), 'roreg_group_conv_pack_weights')
    return torch.from_numpy(out).cuda()


class ConvLayer:
    """One BN(eval)+ReLU+group-conv layer in kernel-ready form (weights packed, BN folded to scale/shift)."""

    def __init__(self, weight, bias, bn=None, eps=1e-5, device='cuda'):
        self.Cout, self.Cin = int(weight.shape[0]), int(weight.shape[1])
        self.KS = int(np.prod(weight.shape[2:]))
        self.wpack = pack_conv_weights(weight)
        self._weight = weight.detach().to('cpu', torch.float32).clone()
        self._wsplit = None
        self.bias = bias.detach().to(device, torch.float32).contiguous()
        if bn is not None:
            g, b, m, v = [t.detach().to('cpu', torch.float32) for t in bn]
            scale = g / torch.sqrt(v + eps)
            shift = b - m * scale
            self.scale = scale.to(device).contiguous()
            self.shift = shift.to(device).contiguous()
        else:
            self.scale = self.shift = None


_gather_cache = {}


def _conv_wsplit(layer):
    if layer._wsplit is None:
        layer._wsplit = group_conv_split_pack(layer._weight)
    return layer._wsplit


def _conv_wsplit2(layer):
    """(fp16 x 2 planes [2][KS][Cin/16][2][Cout][8], w_exp, act_smax, act_tmax) of a ConvLayer, built on first use."""
    if getattr(layer, '_wsplit2', None) is None:
        Wn = layer._weight.numpy()
        Cout, Cin = Wn.shape[0], Wn.shape[1]
        KS = int(np.prod(Wn.shape[2:]))
        w_exp = f16_scale_exp(float(np.abs(Wn).max()))
        Ws = np.ldexp(Wn.reshape(Cout, Cin // 16, 2, 8, KS).astype(np.float32), w_exp).astype(np.float32)
        hi = Ws.astype(np.float16); lo = (Ws - hi.astype(np.float32)).astype(np.float16)
        out = np.empty((2, KS, Cin // 16, 2, Cout, 8), np.uint16)
        for sp, part in enumerate((hi, lo)):
            out[sp] = part.view(np.uint16).transpose(4, 1, 2, 0, 3)
        smax = float(layer.scale.abs().max()) if layer.scale is not None else 1.0
        tmax = float(layer.shift.abs().max()) if layer.shift is not None else 0.0
        layer._wsplit2 = (torch.from_numpy(out.view(np.int16)).cuda(), w_exp, smax, tmax)
    return layer._wsplit2

# bench.py sets this to a list to collect (shape tag, start event, end event) per group-conv launch; the
# events are recorded on the stream the kernel is launched on (torch's current stream).
PROFILE = None


# bench.py sets this to a dict to collect the algorithmic work of the profiled rotation-coherence-matcher kernels:
# 'sinkhorn_bytes' (iterations x bytes of every pair's coupling matrix, ONE pass each), 'topk_flop' (2 x 32 x m x n per searched pair)
WORK = None

PROFILE_SLOTS = {'mm_tile': 0, 'ransac_score': 1, 'des2r': 2, 'ft_nonlin': 3, 'sinkhorn': 4, 'topk_dot': 5}


def profile_enable(on=True):
    """Library-side kernel timing (HIP events on the launch stream around selected launches; include/roreg_hip.h)."""
    _check(lib().roreg_profile_enable(1 if on else 0), 'roreg_profile_enable')


def profile_read(name):
    """-> (total ms, number of timed brackets) of slot `name` (PROFILE_SLOTS) since profile_enable()."""
    ms = c_double(0.0); n = c_int(0)
    _check(lib().roreg_profile_read(PROFILE_SLOTS[name], ctypes.byref(ms), ctypes.byref(n)), 'roreg_profile_read')
    return ms.value, n.value


def gather_table(key, array):
    """int32 device copy of a [Lout,KS] gather table, cached by key."""
    t = _gather_cache.get(key)
    if t is None:
        t = torch.from_numpy(np.ascontiguousarray(array, np.int32)).cuda()
        _gather_cache[key] = t
    return t


def full_gather():
    return gather_table('nei60', tables().Nei)


_lds_order_ok = set()


def _check_lds_order(order_t, stride, gather, Lin):
    """The contract of roreg_group_conv_{split,f16x2}'s lds_order, enforced once per (table, gather) pair on the host: every input column the
    gather table reads has a slot in [0, stride), and no two of them share one (the kernel uses order[gather[i]] as an LDS address
    unchecked: a -1, a slot >= stride or a duplicate would alias or overrun LDS silently)."""
    key = (order_t.data_ptr(), int(stride), gather.data_ptr())
    if key in _lds_order_ok:
        return
    if order_t.numel() != Lin or not 1 <= int(stride) <= 64:
        raise HipError(f'group_conv: lds_order must hold one slot per input column ({Lin}) and a stride in 1..64')
    o = order_t.cpu().numpy(); cols = np.unique(gather.cpu().numpy())
    if cols.min() < 0 or cols.max() >= Lin:
        raise HipError('group_conv: the gather table reads a column outside the input')
    slots = o[cols]
    if slots.min() < 0 or slots.max() >= stride or np.unique(slots).size != slots.size:
        raise HipError(f'group_conv: lds_order must give the {cols.size} gathered columns distinct slots in [0, {stride})')
    _lds_order_ok.add(key)


def group_conv(x, layer, gather=None, Lout=None, residual=None, out=None, split=False, in_rowmax=None, want_rowmax=False, lds_order=None):
    """x [B,Cin,Lin] f32 -> [B,Cout,Lout] f32.  split: use the 3 x bf16 split kernel (f32-accurate) where its shape constraints hold;
    with in_rowmax (device float32 [B], tracked max |x[b]| per row) the fp16 x 2 kernel, whose block scale is per row (a row's result does
    not depend on the other rows of the batch); it can also return the tracked per-row max |out[b]| (want_rowmax).
    lds_order: optional (device int32 [Lin] slot table, stride) -- the split kernels' LDS slot order of the input columns (an execution hint
    against bank conflicts of the gathered reads, tools/lds_perm_search.py; results do not depend on it)."""
    ensure_tables()
    B, Cin, Lin = x.shape
    assert Cin == layer.Cin, (Cin, layer.Cin)
    if gather is None:
        gather = full_gather() if layer.KS == 13 else gather_table(('id', Lin), np.arange(Lin)[:, None])
    Lout = int(gather.shape[0]) if Lout is None else Lout
    if out is None:
        out = torch.empty((B, layer.Cout, Lout), dtype=torch.float32, device=x.device)
    split_ok = residual is None and layer.KS == 13 and Cin % 16 == 0 and layer.Cout % 256 == 0 and Lin <= 64 and Lout <= 64 and B > 0
    # ... and the split kernels' precomputed staging plan (csrc/group_conv.hip, launch_conv_split: RAW_ITERS = 9, CV_ITEMS = 5 trips of 256
    # threads over the tile's keypoints x columns) must cover the tile: e.g. Lin = 60 with Lout = 13 (11 keypoints per tile) does not fit and
    # takes the f32 kernel below instead of an error
    nkp = min((128 - 1) // max(Lout, 1) + 2, max(B, 1))
    split_ok = split_ok and nkp * 2 * Lin <= 5 * 256 and nkp * 4 * Lin <= 9 * 256
    order_t, order_s = (None, 0) if lds_order is None else lds_order
    if order_t is not None:
        _check_lds_order(order_t, order_s, gather, Lin)
    if in_rowmax is not None:
        if not split_ok:
            raise HipError('group_conv: the fp16 x 2 kernel does not support this shape')
        if in_rowmax.numel() != B:
            raise HipError(f'group_conv: in_rowmax must hold one value per row ({B}), got {in_rowmax.numel()}')
        w2, w_exp, smax, tmax = _conv_wsplit2(layer)
        amax = torch.zeros(B, dtype=torch.float32, device=x.device) if want_rowmax else None
        _check(lib().roreg_group_conv_f16x2(_ptr(x, torch.float32), _ptr(w2), w_exp, _ptr(layer.bias), _ptr(layer.scale), _ptr(layer.shift), smax, tmax,
                                            _ptr(in_rowmax, torch.float32), _ptr(out, torch.float32), _ptr(amax), _ptr(gather, torch.int32),
                                            _ptr(order_t, torch.int32), order_s, B, Cin, layer.Cout, Lin, Lout, layer.KS, _stream()), 'roreg_group_conv_f16x2')
        return (out, amax) if want_rowmax else out
    if split and split_ok:
        _check(lib().roreg_group_conv_split(_ptr(x, torch.float32), _ptr(_conv_wsplit(layer)), _ptr(layer.bias), _ptr(layer.scale), _ptr(layer.shift),
                                            _ptr(out, torch.float32), _ptr(gather, torch.int32), _ptr(order_t, torch.int32), order_s, B, Cin, layer.Cout, Lin, Lout,
                                            layer.KS, _stream()), 'roreg_group_conv_split')
        return out
    ws_n = lib().roreg_group_conv_workspace_size(B, Cin, layer.Cout, Lin, Lout, layer.KS)
    ws = torch.empty(ws_n, dtype=torch.float32, device=x.device) if ws_n else None
    if PROFILE is not None:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(lib().roreg_group_conv(_ptr(x, torch.float32), _ptr(layer.wpack), _ptr(layer.bias), _ptr(layer.scale), _ptr(layer.shift),
                                  _ptr(residual, torch.float32), _ptr(out, torch.float32), _ptr(gather, torch.int32),
                                  B, Cin, layer.Cout, Lin, Lout, layer.KS, _ptr(ws), ws_n, _stream()), 'roreg_group_conv')
    if PROFILE is not None:
        e1.record()
        PROFILE.append(((B, Cin, layer.Cout, Lout, layer.KS), e0, e1))
    return out


def group_conv_packed(x_words, layer, in_bound, gather, want_rowmax=False, lds_order=None):
    """The fp16 x 2 stencil convolution on PACKED input words (ft_nonlin_packed: BatchNorm, ReLU and the hi / lo split already applied under the
    block scale of in_bound [>= B]): x_words int32 [B,Cin,Lin] -> out f32 [B,Cout,Lout] (+ tracked max |out[b]|).  layer: a ConvLayer; only its
    weights and bias are used (its BatchNorm belongs to the producer)."""
    ensure_tables()
    B, Cin, Lin = x_words.shape
    assert Cin == layer.Cin and x_words.dtype == torch.int32, (Cin, layer.Cin, x_words.dtype)
    Lout = int(gather.shape[0])
    if not (layer.KS == 13 and Cin % 16 == 0 and layer.Cout % 256 == 0 and Lin <= 64 and Lout <= 64 and B > 0):
        raise HipError('group_conv_packed: the fp16 x 2 kernel does not support this shape')
    if in_bound.numel() < B:
        raise HipError(f'group_conv_packed: in_bound must hold one value per row ({B}), got {in_bound.numel()}')
    order_t, order_s = (None, 0) if lds_order is None else lds_order
    if order_t is not None:
        _check_lds_order(order_t, order_s, gather, Lin)
    w2, w_exp, _, _ = _conv_wsplit2(layer)
    out = torch.empty((B, layer.Cout, Lout), dtype=torch.float32, device=x_words.device)
    amax = torch.zeros(B, dtype=torch.float32, device=x_words.device) if want_rowmax else None
    _check(lib().roreg_group_conv_f16x2_packed(_ptr(x_words, torch.int32), _ptr(w2), w_exp, _ptr(layer.bias), _ptr(in_bound, torch.float32), _ptr(out),
                                               _ptr(amax), _ptr(gather, torch.int32), _ptr(order_t, torch.int32), order_s, B, Cin, layer.Cout, Lin, Lout,
                                               layer.KS, _stream()), 'roreg_group_conv_f16x2_packed')
    return (out, amax) if want_rowmax else out


def gf_finalize(eqv_raw, want_inv=True, out_dtype=torch.float32):
    """out_dtype=torch.bfloat16: the descriptors are stored in bfloat16 (BASELINE config 5); `inv` is computed from the float32 values."""
    B = eqv_raw.shape[0]
    eqv = torch.empty(eqv_raw.shape, dtype=out_dtype, device=eqv_raw.device)
    inv = torch.empty((B, 32), dtype=torch.float32, device=eqv_raw.device) if want_inv else None
    ep, bf = _feat(eqv)
    _check(lib().roreg_gf_finalize(_ptr(eqv_raw, torch.float32), ep, bf, _ptr(inv), B, _stream()), 'roreg_gf_finalize')
    return eqv, inv


def det_score(enc):
    ensure_tables()
    B = enc.shape[0]
    assert enc.shape[1:] == (16, 60)
    out = torch.empty(B, dtype=torch.float32, device=enc.device)
    _check(lib().roreg_det_score(_ptr(enc, torch.float32), _ptr(out), B, _stream()), 'roreg_det_score')
    return out


def inv_descriptor(eqv):
    N = eqv.shape[0]
    assert eqv.shape[1:] == (32, 60)
    out = torch.empty((N, 32), dtype=torch.float32, device=eqv.device)
    ep, bf = _feat(eqv)
    _check(lib().roreg_inv_descriptor(ep, bf, _ptr(out), N, _stream()), 'roreg_inv_descriptor')
    return out


def nn_search(src, tgt, src_rows=None, tgt_rows=None, want_dist=False, squared=False):
    """nearest target for every source.  src [*,F], tgt [*,F] f32; optional int64 row lists.  squared: dist_type 'SquareL2'."""
    F = src.shape[1]
    m = int(src_rows.shape[0]) if src_rows is not None else int(src.shape[0])
    n = int(tgt_rows.shape[0]) if tgt_rows is not None else int(tgt.shape[0])
    idx = torch.empty(m, dtype=torch.int64, device=src.device)
    dist = torch.empty(m, dtype=torch.float32, device=src.device) if want_dist else None
    scratch = torch.empty(max(m, 1), dtype=torch.int64, device=src.device)
    _check(lib().roreg_nn_search_ex(_ptr(src, torch.float32), _ptr(src_rows, torch.int64), m, _ptr(tgt, torch.float32),
                                    _ptr(tgt_rows, torch.int64), n, F, 1 if squared else 0, _ptr(idx), _ptr(dist), _ptr(scratch), _stream()), 'roreg_nn_search')
    return (idx, dist) if want_dist else idx


def knn_search(src, tgt, k, want_dist=False, squared=False):
    m, F = src.shape
    n = tgt.shape[0]
    idx = torch.empty((m, k), dtype=torch.int64, device=src.device)
    dist = torch.empty((m, k), dtype=torch.float32, device=src.device) if want_dist else None
    ws_n = lib().roreg_knn_search_workspace(m, n)
    ws = torch.empty(ws_n // 4, dtype=torch.int32, device=src.device) if ws_n else None
    _check(lib().roreg_knn_search_ex(_ptr(src, torch.float32), m, _ptr(tgt, torch.float32), n, F, k, 1 if squared else 0, _ptr(idx), _ptr(dist),
                                     _ptr(ws), ws_n, _stream()), 'roreg_knn_search')
    return (idx, dist) if want_dist else idx


def pdist(A, B, squared=False):
    """[m,F] x [n,F] f32 -> the [m,n] distance matrix of modified_knn_matcher.pdist (utils/knn_search.py:17-24)."""
    m, F = A.shape
    n = B.shape[0]
    if B.shape[1] != F:
        raise HipError(f'pdist: feature widths differ ({F} vs {B.shape[1]})')
    out = torch.empty((m, n), dtype=torch.float32, device=A.device)
    _check(lib().roreg_pdist(_ptr(A, torch.float32), m, _ptr(B, torch.float32), n, F, 1 if squared else 0, _ptr(out), _stream()), 'roreg_pdist')
    return out


def knn_search_seg(pts, seg, k):
    """k nearest neighbours of every point among the points of its own segment (pts [total,F] stacked clouds, seg: Segments) ->
    int64 [total,k] indices local to the segment."""
    total, F = pts.shape
    idx = torch.empty((total, k), dtype=torch.int64, device=pts.device)
    if seg.min < k:
        raise HipError(f'knn_search_seg: a segment has fewer than k={k} points')
    ws_n = lib().roreg_knn_search_seg_workspace(total, seg.n, seg.max, seg.max)
    ws = torch.empty(ws_n // 4, dtype=torch.int32, device=pts.device)
    sp = _ptr(seg.dev, torch.int32)
    _check(lib().roreg_knn_search_seg(_ptr(pts, torch.float32), _ptr(pts, torch.float32), sp, sp, seg.n, total, seg.max, seg.max, F, k, _ptr(idx),
                                      _ptr(ws), ws_n, _stream()), 'roreg_knn_search_seg')
    return idx


def mutual_matches(nn01, nn10, sample0=None, sample1=None):
    """-> (matches int64 [m,2] buffer, count int32[1]); rows [0,count) are valid, in increasing source order."""
    m = nn01.shape[0]
    out = torch.empty((max(m, 1), 2), dtype=torch.int64, device=nn01.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=nn01.device)
    _check(lib().roreg_mutual_matches(_ptr(nn01, torch.int64), _ptr(nn10, torch.int64), m, int(nn10.shape[0]), _ptr(sample0, torch.int64),
                                      _ptr(sample1, torch.int64), _ptr(out), _ptr(cnt), _stream()), 'roreg_mutual_matches')
    return out, cnt




class LtBatch:
    """Task table of the batched local-transform stage.  tasks: [(before0, before1, after0, after1 [*,32,60] f32 (or all bf16), keys0, keys1
    [*,3] f64, matches [M,2] int64, sel int64 [n] or None[, coef0, coef1 = feat_coefs(after0 / after1)])] device tensors; task p owns
    output rows [off[p], off[p]+n[p]).  With the coefficient tensors Des2R runs through the irrep-domain bound + exact re-check."""

    def __init__(self, tasks):
        self.n_tasks = len(tasks)
        table = np.zeros(self.n_tasks, _LT_TASK)
        off = 0
        self.offsets = []
        n_coef = 0; n_bf16 = 0
        # a scene's pairs share its clouds: every distinct tensor is validated once per batch (449 kitchen pairs touch 60 clouds), the
        # table is filled column by column -- this constructor runs with the GPU idle behind the match-count download
        seen = {}

        def feat(t):                                           # -> (device pointer, is bfloat16)
            r = seen.get(id(t))
            if r is None:
                r = seen[id(t)] = (t.data_ptr(), _feat(t)[1])
            return r

        def vptr(t, dtype):
            r = seen.get(id(t))
            if r is None:
                _ptr(t, dtype)
                r = seen[id(t)] = (t.data_ptr(), False)
            return r[0]

        cols = [[] for _ in range(12)]
        for task in tasks:
            b0, b1, a0, a1, k0, k1, m, sel = task[:8]
            c0, c1 = (task[8], task[9]) if len(task) > 8 else (None, None)
            f = [feat(t) for t in (b0, b1, a0, a1)]
            if not (f[0][1] == f[1][1] == f[2][1] == f[3][1]):
                raise HipError('LtBatch: the four feature tensors of a task must share one dtype')
            n_bf16 += f[0][1]
            n = int(sel.shape[0]) if sel is not None else int(m.shape[0])
            if c0 is not None:
                n_coef += 1
            row = (f[0][0], f[1][0], f[2][0], f[3][0], vptr(k0, torch.float64), vptr(k1, torch.float64), vptr(m, torch.int64) if n else 0,
                   vptr(sel, torch.int64) if sel is not None else 0, n, off,
                   vptr(c0, torch.float32) if c0 is not None else 0, vptr(c1, torch.float32) if c1 is not None else 0)
            for col, v in zip(cols, row):
                col.append(v)
            self.offsets.append((off, n))
            off += n
        for name, col in zip(('before0', 'before1', 'after0', 'after1', 'keys0', 'keys1', 'matches', 'sel', 'n', 'off', 'coef0', 'coef1'), cols):
            table[name] = col
        if n_coef not in (0, self.n_tasks) or n_bf16 not in (0, self.n_tasks):
            raise HipError('LtBatch: either every task carries coefficient tensors / bfloat16 features or none does')
        self.flags = (1 if n_coef else 0) | (2 if n_bf16 else 0)
        if self.flags & 1:
            ensure_des2r()
        self.total = off
        self.max_n = int(table['n'].max()) if self.n_tasks else 0
        self.keep = tasks                              # the table holds raw pointers: keep the tensors alive
        self.table = upload(table.view(np.uint8).reshape(self.n_tasks, _LT_TASK.itemsize)) if self.n_tasks else None

    def prepare(self, rows_alloc=None, bound_bn=None):
        """Des2R + ET input assembly -> (dr int64 [total], x [rows_alloc,128,60] f32; rows beyond total are zero).
        bound_bn = (scale, shift) of Conv_init's BatchNorm: also returns the per-row bound row_bound(x, bn) [coef_pitch(rows_alloc)], computed
        while x is assembled instead of in a pass of its own."""
        ensure_tables()
        rows_alloc = self.total if rows_alloc is None else rows_alloc
        dr = torch.empty(self.total, dtype=torch.int64, device='cuda')
        x = torch.empty((rows_alloc, 128, 60), dtype=torch.float32, device='cuda')
        if rows_alloc > self.total:
            x[self.total:].zero_()
        bound = torch.zeros(coef_pitch(rows_alloc), dtype=torch.float32, device='cuda') if bound_bn is not None else None
        sc, sh = bound_bn if bound_bn is not None else (None, None)
        if self.total:
            _check(lib().roreg_lt_prepare_batch(_ptr(self.table), self.n_tasks, self.max_n, self.flags, _ptr(dr), _ptr(x), _ptr(sc, torch.float32),
                                                _ptr(sh, torch.float32), _ptr(bound), _stream()), 'roreg_lt_prepare_batch')
        return (dr, x, bound) if bound_bn is not None else (dr, x)

    def des2r(self):
        """Des2R alone -> dr int64 [total] (the YOHO-C estimator's DR_index, test/estimator.py:85-111)."""
        ensure_tables()
        dr = torch.empty(self.total, dtype=torch.int64, device='cuda')
        if self.total:
            _check(lib().roreg_lt_prepare_batch(_ptr(self.table), self.n_tasks, self.max_n, self.flags, _ptr(dr), None, None, None, None, _stream()),
                   'roreg_lt_prepare_batch')
        return dr

    def finish(self, q_all, dr):
        """un-normalised quaternions [>=total,4] f32 + dr -> Trans [total,3,4] f64."""
        T = torch.empty((self.total, 3, 4), dtype=torch.float64, device='cuda')
        if self.total:
            _check(lib().roreg_lt_finish_batch(_ptr(self.table), self.n_tasks, self.max_n, _ptr(q_all, torch.float32), _ptr(dr, torch.int64), _ptr(T),
                                               _stream()), 'roreg_lt_finish_batch')
        return T


def yohoc_draw(prob, bin_size, max_iter, max_tries=50000, rng=None):
    """The reference's YOHO-C sampling loop (test/estimator.py:220-230) replayed on a legacy numpy generator -- rng=None: the process-GLOBAL
    one, as the reference uses it; or a `np.random.RandomState` of the caller's (the engine's per-pair streams): same calls, same
    order, same final generator state as `np.random.choice(range(60), p=prob)` + `np.random.choice(members, 3)` per try, without the
    per-call Python overhead.  prob f64 [60], bin_size [60] -> (bins int32 [H], picks int64 [H,3] positions inside the bin)."""
    rng = np.random if rng is None else rng
    prob = np.ascontiguousarray(prob, np.float64)
    cdf = prob.cumsum()
    cdf /= cdf[-1]                                                  # numpy's RandomState.choice does exactly this
    sizes = np.ascontiguousarray(bin_size, np.int32)
    bins = np.empty(max(max_iter, 1), np.int32); picks = np.empty((max(max_iter, 1), 3), np.int64)
    n_hyp = ctypes.c_int32(0); used = ctypes.c_longlong(0)
    state = rng.get_state()
    n_words = 16 * max_iter + 64
    while True:
        rng.set_state(state)
        words = rng.randint(0, 2 ** 32, size=n_words, dtype=np.uint32)       # raw generator words (full-range uint32 draws)
        rc = lib().roreg_yohoc_draw(words.ctypes.data, n_words, cdf.ctypes.data, sizes.ctypes.data, int(max_iter), int(max_tries),
                                    bins.ctypes.data, picks.ctypes.data, ctypes.byref(n_hyp), ctypes.byref(used))
        if rc == 3:
            n_words *= 4
            continue
        _check(rc, 'roreg_yohoc_draw')
        break
    rng.set_state(state)
    if used.value:
        rng.randint(0, 2 ** 32, size=used.value, dtype=np.uint32)            # advance the generator by what the loop consumed
    return bins[:n_hyp.value].copy(), picks[:n_hyp.value].copy()




_NPY_HEADERS = {}


def npy_header(dtype, shape):
    """The bytes np.save puts in front of a C-ordered array of this dtype and shape (numpy's own header writer, version 1.0; cached)."""
    key = (np.dtype(dtype).str, tuple(int(x) for x in shape))
    h = _NPY_HEADERS.get(key)
    if h is None:
        import io
        f = io.BytesIO()
        np.lib.format.write_array_header_1_0(f, {'descr': np.lib.format.dtype_to_descr(np.dtype(dtype)), 'fortran_order': False, 'shape': key[1]})
        h = f.getvalue()
        if len(_NPY_HEADERS) < 65536:
            _NPY_HEADERS[key] = h
    return h


def write_npy_files(paths, arrays, n_threads=4):
    """np.save(path, a) for many C-contiguous host arrays in ONE call that releases the interpreter lock (roreg_write_files): byte for byte
    np.save's files (the same header bytes, the same data)."""
    n = len(paths)
    if n == 0:
        return
    arrs = [np.ascontiguousarray(a) for a in arrays]
    heads = [npy_header(a.dtype, a.shape) for a in arrs]
    c_paths = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    c_heads = (ctypes.c_char_p * n)(*heads)
    c_hlen = (ctypes.c_int32 * n)(*[len(h) for h in heads])
    c_data = (ctypes.c_void_p * n)(*[a.ctypes.data if a.size else None for a in arrs])
    c_len = (ctypes.c_int64 * n)(*[a.nbytes for a in arrs])
    _check(lib().roreg_write_files(c_paths, c_heads, c_hlen, c_data, c_len, n, int(n_threads)), 'roreg_write_files')


def write_files(paths, blobs, n_threads=4):
    """file q = blobs[q] (bytes-like), many files in ONE call that releases the interpreter lock (roreg_write_files with empty headers)."""
    n = len(paths)
    if n == 0:
        return
    views = [np.frombuffer(b, np.uint8) for b in blobs]
    empty = ctypes.c_char_p(b'')
    c_paths = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    c_heads = (ctypes.c_char_p * n)(*([empty.value] * n))
    c_hlen = (ctypes.c_int32 * n)(*([0] * n))
    c_data = (ctypes.c_void_p * n)(*[v.ctypes.data if v.size else None for v in views])
    c_len = (ctypes.c_int64 * n)(*[v.nbytes for v in views])
    _check(lib().roreg_write_files(c_paths, c_heads, c_hlen, c_data, c_len, n, int(n_threads)), 'roreg_write_files')


def yohoc_draw_many(seeds, anchors_list, max_iter, max_tries=50000, n_threads=None):
    """yohoc_draws() of many pairs in one threaded host call, pair p from np.random.RandomState(seeds[p]) (roreg_yohoc_draw_many):
    anchors_list[p] = the coarse rotations of the correspondences pair p draws from.  -> [rows int64 [H_p, 3] or None (the reference gives up),
    give-up matrices f64 [n_pairs, 4, 4] (rng.rand(4, 4) of the pairs that gave up)]."""
    n = len(anchors_list)
    seeds = np.ascontiguousarray(np.asarray(seeds, np.int64) % (2 ** 32), np.uint32)
    offs = np.zeros(n + 1, np.int64)
    offs[1:] = np.cumsum([len(a) for a in anchors_list])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(a, np.int64) for a in anchors_list]) if n else np.zeros(0, np.int64))
    rows = np.empty((n, max(max_iter, 1), 3), np.int64); n_hyp = np.empty(n, np.int32); giveup = np.zeros((n, 4, 4), np.float64)
    if n:
        nt = n_threads if n_threads is not None else max(1, min(32, (os.cpu_count() or 2) // 2, n // 8 + 1))
        _check(lib().roreg_yohoc_draw_many(c_void_p(seeds.ctypes.data), n, c_void_p(flat.ctypes.data), c_void_p(offs.ctypes.data), int(max_iter), int(max_tries),
                                           c_void_p(rows.ctypes.data), c_void_p(n_hyp.ctypes.data), c_void_p(giveup.ctypes.data), int(nt)), 'roreg_yohoc_draw_many')
    return [None if n_hyp[p] < 0 else rows[p, :n_hyp[p]] for p in range(n)], giveup


def ransac_batch(tasks, ird, w_f32=False, keep=False):
    """tasks: [(keys0 [*,3] f64, keys1 [*,3] f64, matches [M,2] int64, w [M] f64 or None, Trans [*,3,4] f64, hyp_rows int64 [H] or None)]
    (device tensors).  One-shot RANSAC + the two refinements of every task in five launches ->
    (best int32 [n], T1 [n,4,4], stats1 [n,16], T2 [n,4,4], stats2 [n,16]) device tensors.
    w_f32: the tasks' weights are float32 scores (widened to f64 for the upload): numpy's float32 reductions, see ransac_score.
    keep: also return the call's context (task table, gathered keypoints) for refine_batch()."""
    n = len(tasks)
    dev = tasks[0][0].device if n else torch.device('cuda')
    best = torch.empty(n, dtype=torch.int32, device=dev)
    T1 = torch.empty((n, 4, 4), dtype=torch.float64, device=dev); T2 = torch.empty_like(T1)
    st1 = torch.empty((n, 16), dtype=torch.float64, device=dev); st2 = torch.empty_like(st1)
    if n == 0:
        return (best, T1, st1, T2, st2, None) if keep else (best, T1, st1, T2, st2)
    table = np.zeros(n, _RANSAC_TASK)
    koff = 0
    for i, (k0, k1, m, w, Tr, hr) in enumerate(tasks):
        _ptr(k0, torch.float64); _ptr(k1, torch.float64); _ptr(m, torch.int64); _ptr(Tr, torch.float64)
        if w is not None:
            _ptr(w, torch.float64)
        if hr is not None:
            _ptr(hr, torch.int64)
        M = int(m.shape[0]); H = int(hr.shape[0]) if hr is not None else int(Tr.shape[0])
        table[i] = (k0.data_ptr(), k1.data_ptr(), m.data_ptr() if M else 0, w.data_ptr() if w is not None else 0, Tr.data_ptr() if H else 0,
                    hr.data_ptr() if hr is not None else 0, M, H, koff)
        koff += M
    max_M = int(table['M'].max()); max_H = int(table['H'].max())
    tdev = upload(table.view(np.uint8).reshape(n, _RANSAC_TASK.itemsize))
    ws_n = lib().roreg_ransac_batch_workspace(n, koff, max_H)
    ws = torch.empty(max(ws_n, 8) // 8, dtype=torch.float64, device=dev)
    _check(lib().roreg_ransac_batch(_ptr(tdev), n, koff, max(max_M, 1), max_H, float(ird), int(bool(w_f32)), _ptr(best), _ptr(T1), _ptr(st1), _ptr(T2), _ptr(st2),
                                    _ptr(ws), ws_n, _stream()), 'roreg_ransac_batch')
    if keep:
        return best, T1, st1, T2, st2, (tdev, ws, koff, bool(w_f32), tasks)
    return best, T1, st1, T2, st2


def refine_batch(ctx, sel, T_in, dist):
    """One more refinement at `dist` of the tasks `sel` (host ints) of ransac_batch(keep=True)'s context, from T_in [len(sel),4,4] f64 (host or
    device) -> (T [len(sel),4,4], stats [len(sel),16]) device tensors, one launch."""
    tdev, ws, total_M, w_f32, _tasks = ctx
    n = len(sel)
    dev = ws.device
    T = torch.empty((n, 4, 4), dtype=torch.float64, device=dev); st = torch.empty((n, 16), dtype=torch.float64, device=dev)
    if n == 0:
        return T, st
    sel_dev = upload(np.ascontiguousarray(sel, np.int32))
    Tin = T_in if torch.is_tensor(T_in) else upload(np.ascontiguousarray(T_in, np.float64))
    _check(lib().roreg_refine_batch(_ptr(tdev), _ptr(sel_dev, torch.int32), n, total_M, _ptr(Tin, torch.float64), float(dist), int(w_f32), _ptr(T), _ptr(st),
                                    _ptr(ws), _stream()), 'roreg_refine_batch')
    return T, st




def mutual_match_batch(tasks):
    """tasks: [(desc0 [*,32] f32, desc1 [*,32] f32, rows0 int64 [m0] or None, rows1 int64 [m1] or None)] (device tensors).
    The mutual matcher of every task in three launches -> (match buffer int64 [n_tasks, pitch, 2], counts int32 [n_tasks]);
    rows [0,count) of task p are its mutual pairs (rows0 value, rows1 value) in increasing source order."""
    n = len(tasks)
    table = np.zeros(n, _MATCH_TASK)
    keep = []
    for i, (d0, d1, r0, r1) in enumerate(tasks):
        if d0.shape[1] != 32 or d1.shape[1] != 32:
            raise HipError('mutual_match_batch: descriptors must be [*,32] float32')
        for t in (d0, d1):
            _ptr(t, torch.float32)
        table[i] = (d0.data_ptr(), d1.data_ptr(), r0.data_ptr() if r0 is not None else 0, r1.data_ptr() if r1 is not None else 0,
                    r0.shape[0] if r0 is not None else d0.shape[0], r1.shape[0] if r1 is not None else d1.shape[0])
        for t in (r0, r1):
            if t is not None:
                _ptr(t, torch.int64)
    dev = tasks[0][0].device if n else torch.device('cuda')
    max_m = int(max([0] + [max(int(t['m0']), int(t['m1'])) for t in table]))
    pitch = (max_m + 1) & ~1
    out = torch.empty((n, max(pitch, 1), 2), dtype=torch.int64, device=dev)
    cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)[:n]
    if n == 0:
        return out, cnt
    tdev = upload(table.view(np.uint8).reshape(n, _MATCH_TASK.itemsize))
    ws_n = lib().roreg_mutual_match_batch_workspace(n, max_m)
    ws = torch.empty(max(ws_n, 8) // 8, dtype=torch.int64, device=dev)
    _check(lib().roreg_mutual_match_batch(_ptr(tdev), n, max_m, _ptr(out), _ptr(cnt), _ptr(ws), ws_n, _stream()), 'roreg_mutual_match_batch')
    return out, cnt


_des2r_ready = False


def des2r_tables(transpose=False):
    """(ia, ib uint8 [60,5], cnt uint8 [60], NT float32 [60 (q), 60 (a)]) of the irrep-domain group correlation, derived from the
    multiplication table.  transpose=False (Des2R, test/estimator.py:85-89): x -> x[P[a,.]] acts on the coefficient matrices as
    X(rho) -> rho(a)^T X(rho), so
        cor[a] = sum_g x1[P[a,g]] x2[g] = sum_rho <rho(a)^T X1, X2> = sum_q NT[q][a] C[q],   C[(rho,i,j)] = sum_k X2[(rho,i,k)] X1[(rho,j,k)].
    transpose=True (the matcher's R_indicator, network/rot_coh_match.py:154-163): x -> x[P[.,a]] acts as X -> X rho(a)^T, so
        cor[a] = sum_g x1[P[g,a]] x2[g] = sum_q NT[q][a] C[q],   C[(rho,k,j)] = sum_i X1[(rho,i,k)] X2[(rho,i,j)],   NT[(rho,k,j)][a] = rho(a)[j][k].
    (Coefficient index of (rho,r,c) = offset_rho + r*d + c; ia indexes the broadcast side X2, ib the permuted side X1.)  The identity is
    asserted here in float64 on random data."""
    from .fourier import group_fourier, DIMS
    gf = group_fourier(); T = tables()
    ia = np.zeros((60, 5), np.uint8); ib = np.zeros((60, 5), np.uint8); cnt = np.zeros(60, np.uint8)
    NT = np.zeros((60, 60), np.float64)
    for q, (ri, i, j) in enumerate(gf.index):
        d = DIMS[ri]; off = int(gf.offsets[ri])
        cnt[q] = d
        for k in range(5):
            kk = min(k, d - 1)
            if not transpose:
                ia[q, k] = off + i * d + kk; ib[q, k] = off + j * d + kk
            else:                                              # q = (rho, k = i, j): sum over the row index kk
                ia[q, k] = off + kk * d + j; ib[q, k] = off + kk * d + i
        NT[q, :] = gf.rho[ri][:, j, i]
    rng = np.random.default_rng(0)
    d1 = rng.standard_normal((4, 60)); d2 = rng.standard_normal((4, 60))
    perm = (lambda a: T.P[:, a]) if transpose else (lambda a: T.P[a])
    want = np.array([(d1[:, perm(a)] * d2).sum() for a in range(60)])
    X1 = d1 @ gf.F.T; X2 = d2 @ gf.F.T
    C = np.array([sum((X2[f, ia[q, :cnt[q]]] * X1[f, ib[q, :cnt[q]]]).sum() for f in range(4)) for q in range(60)])
    assert np.abs(C @ NT - want).max() < 1e-11, 'irrep-domain correlation identity violated'
    return ia, ib, cnt, np.ascontiguousarray(NT, np.float32)


def ensure_des2r():
    global _des2r_ready
    if not _des2r_ready:
        ensure_tables(); ensure_fourier()
        for mode in (0, 1):
            ia, ib, cnt, NT = des2r_tables(transpose=bool(mode))
            _check(lib().roreg_set_des2r_tables(mode, ia.ctypes.data, ib.ctypes.data, cnt.ctypes.data, NT.ctypes.data), 'roreg_set_des2r_tables')
        _des2r_ready = True


def feat_coefs(x):
    """Group-Fourier coefficients of a group-domain tensor x [B,C,60] (float32 or bfloat16) in per-keypoint layout -> float32 [B,C,60]
    (the operand of the irrep-domain Des2R; computed once per cloud)."""
    ensure_fourier()
    B, C = int(x.shape[0]), int(x.shape[1])
    out = torch.empty((B, C, 60), dtype=torch.float32, device=x.device)
    xp, bf = _feat(x)
    _check(lib().roreg_feat_coefs(xp, bf, _ptr(out), B, C, {'f16x2': 2, 'bf16x3': 1, 'f32': 0}[GEMM_MODE], _stream()), 'roreg_feat_coefs')
    return out


def des2r_recheck_count(reset=True):
    """Correspondences that took the exact (literal) path of the irrep-domain Des2R since the last reset (synchronises)."""
    ensure_des2r()
    n = ctypes.c_int32(0)
    _check(lib().roreg_des2r_recheck_count(1 if reset else 0, ctypes.byref(n)), 'roreg_des2r_recheck_count')
    return n.value


def des2r(feats1, feats0, rows1=None, rows0=None, want_cor=False, coefs1=None, coefs0=None):
    """First arg-max of the 60 local-rotation correlations per correspondence (test/estimator.py:85-89).  With coefs1 / coefs0 =
    feat_coefs(feats1 / feats0): the irrep-domain bound + exact re-check of near ties (same index, ~10x fewer operations)."""
    ensure_tables()
    M = int(rows1.shape[0]) if rows1 is not None else int(feats1.shape[0])
    if coefs1 is not None and not want_cor:
        ensure_des2r()
        idx = torch.empty(M, dtype=torch.int64, device=feats1.device)
        p1, bf1 = _feat(feats1); p0, bf0 = _feat(feats0)
        if bf1 != bf0:
            raise HipError('des2r: feats1 and feats0 must share one dtype')
        _check(lib().roreg_des2r_irrep(_ptr(coefs1, torch.float32), _ptr(rows1, torch.int64), _ptr(coefs0, torch.float32), _ptr(rows0, torch.int64),
                                       p1, p0, bf1, M, _ptr(idx), _stream()), 'roreg_des2r_irrep')
        return idx
    idx = torch.empty(M, dtype=torch.int64, device=feats1.device)
    cor = torch.empty((M, 60), dtype=torch.float32, device=feats1.device) if want_cor else None
    _check(lib().roreg_des2r(_ptr(feats1, torch.float32), _ptr(rows1, torch.int64), _ptr(feats0, torch.float32), _ptr(rows0, torch.int64),
                             M, _ptr(idx), _ptr(cor), _stream()), 'roreg_des2r')
    return (idx, cor) if want_cor else idx


def et_gather(before0, before1, after0, after1, pre_idx, rows0=None, rows1=None, out=None):
    ensure_tables()
    M = pre_idx.shape[0]
    x = out if out is not None else torch.empty((M, 128, 60), dtype=torch.float32, device=before0.device)
    assert x.shape == (M, 128, 60)
    ptrs = [_feat(t) for t in (before0, before1, after0, after1)]
    if len({f for _, f in ptrs}) != 1:
        raise HipError('et_gather: the four feature tensors must share one dtype')
    _check(lib().roreg_et_gather(ptrs[0][0], ptrs[1][0], ptrs[2][0], ptrs[3][0], ptrs[0][1], _ptr(rows0, torch.int64), _ptr(rows1, torch.int64),
                                 _ptr(pre_idx, torch.int64), M, _ptr(x), _stream()), 'roreg_et_gather')
    return x


def quat_to_trans(q, anchor, keys0, keys1, rows0=None, rows1=None, want_quat=False):
    ensure_tables()
    M = q.shape[0]
    T = torch.empty((M, 3, 4), dtype=torch.float64, device=q.device)
    qn = torch.empty((M, 4), dtype=torch.float32, device=q.device) if want_quat else None
    _check(lib().roreg_quat_to_trans(_ptr(q, torch.float32), _ptr(anchor, torch.int64), _ptr(keys0, torch.float64), _ptr(rows0, torch.int64),
                                     _ptr(keys1, torch.float64), _ptr(rows1, torch.int64), M, _ptr(T), _ptr(qn), _stream()), 'roreg_quat_to_trans')
    return (T, qn) if want_quat else T


def ransac_score(k0, k1, w, Trans, ird, hyp_rows=None, want_mask=False, w_f32=False):
    """w_f32: `w` holds the rotation-coherence matcher's float32 scores (widened): overlaps are numpy's float32 pairwise sums / float32 M,
    as test/estimator.py:381 computes them on a float32 score array."""
    M = k0.shape[0]
    H = int(hyp_rows.shape[0]) if hyp_rows is not None else int(Trans.shape[0])
    ov = torch.empty(max(H, 1), dtype=torch.float64, device=k0.device)
    best = torch.empty(1, dtype=torch.int32, device=k0.device)
    mask = torch.empty((H, M), dtype=torch.uint8, device=k0.device) if want_mask else None
    _check(lib().roreg_ransac_score(_ptr(k0, torch.float64), _ptr(k1, torch.float64), _ptr(w, torch.float64), int(bool(w_f32)), M, _ptr(Trans, torch.float64),
                                    _ptr(hyp_rows, torch.int64), H, float(ird), _ptr(ov), _ptr(best), _ptr(mask), _stream()), 'roreg_ransac_score')
    return ov[:H], best, mask


def refine(k0, k1, w, dist, T_in=None, Trans=None, hyp_rows=None, best=None, want_stats=False, w_f32=False):
    """-> T [4,4] f64 (device)  [, stats f64[16] = H(9), c0(3), c1(3), sum w].  w_f32: float32 scores (see ransac_score): the weights are
    normalised in float32 like the reference's scores / np.sum(scores) on a float32 array."""
    M = k0.shape[0]
    out = torch.empty((4, 4), dtype=torch.float64, device=k0.device)
    stats = torch.empty(16, dtype=torch.float64, device=k0.device) if want_stats else None
    _check(lib().roreg_refine(_ptr(k0, torch.float64), _ptr(k1, torch.float64), _ptr(w, torch.float64), int(bool(w_f32)), M, _ptr(T_in, torch.float64), 4,
                              _ptr(Trans, torch.float64), _ptr(hyp_rows, torch.int64), _ptr(best, torch.int32), float(dist), _ptr(out),
                              _ptr(stats), _stream()), 'roreg_refine')
    return (out, stats) if want_stats else out


def kabsch_from_stats(stats):
    """Host evaluation of the refinement's closing step from the device-reduced statistics, with the same
    LAPACK call as the reference (np.linalg.svd; test/estimator.py:39-43,48-51) -> [4,4] float64."""
    s = stats if isinstance(stats, np.ndarray) else stats.cpu().numpy()
    H = s[0:9].reshape(3, 3); c0 = s[9:12]; c1 = s[12:15]
    U, _, VT = np.linalg.svd(H)
    R = U @ VT
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = c0 - c1 @ R.T
    return T


def stats_rank_deficient(stats, tol=1e-10):
    s = stats if isinstance(stats, np.ndarray) else stats.cpu().numpy()
    H = s[0:9].reshape(3, 3)
    if not np.isfinite(H).all():
        return False
    sv = np.linalg.svd(H, compute_uv=False)
    # rank <= 2 (three inliers, coplanar inliers, ...): the third singular pair is a null pair and whether U V^T is a rotation or a
    # reflection is the SVD routine's sign convention -- the reference's value is LAPACK's
    return bool(sv[2] <= tol * max(sv[0], 1e-300))


def stats_rank_deficient_many(stats, tol=1e-10):
    """stats_rank_deficient for a stack of refinement statistics [..., >= 9] in ONE batched SVD -> bool array [...] (the per-pair loop cost
    ~8 us x 2 x pairs of host time with the GPU idle).  Singular values do not depend on the batching (LAPACK gesdd per matrix either way)."""
    s = np.asarray(stats)
    H = s[..., 0:9].reshape(s.shape[:-1] + (3, 3))
    finite = np.isfinite(H).all(axis=(-1, -2))
    sv = np.linalg.svd(np.where(finite[..., None, None], H, 0.0), compute_uv=False)
    return finite & (sv[..., 2] <= tol * np.maximum(sv[..., 0], 1e-300))


def mt_shuffle_prefix(seeds, sizes, take, n_threads=None):
    """Host function: per job j `np.random.seed(seeds[j])`, then for each list of sizes[j] `idx = np.arange(n); np.random.shuffle(idx);
    idx[:take]` (numpy's legacy MT19937 stream, replayed in C over host threads).  seeds [J] (any integers; taken modulo 2^32), sizes [J, L]
    -> int64 [J, L, take] with -1 beyond a list's length."""
    seeds = np.ascontiguousarray(np.asarray(seeds, np.int64) % (2 ** 32), np.uint32)
    sizes = np.ascontiguousarray(sizes, np.int32).reshape(seeds.shape[0], -1)
    out = np.empty((seeds.shape[0], sizes.shape[1], int(take)), np.int64)
    if seeds.shape[0]:
        nt = n_threads if n_threads is not None else max(1, min(32, (os.cpu_count() or 2) // 2, seeds.shape[0] // 8 + 1))
        _check(lib().roreg_mt_shuffle_prefix(c_void_p(seeds.ctypes.data), seeds.shape[0], c_void_p(sizes.ctypes.data), sizes.shape[1], int(take),
                                             c_void_p(out.ctypes.data), int(nt)), 'roreg_mt_shuffle_prefix')
    return out




def global_stream_shuffle_prefix(sizes, take):
    """Host function: `for n in sizes: idx = np.arange(n); np.random.shuffle(idx); idx[:take]` on the PROCESS-GLOBAL numpy generator (the stream an
    unseeded Test.py consumes), replayed in C (roreg_mt_stream_shuffle_prefix): the generator is left exactly where numpy would have left it.
    sizes [L] -> int64 [L, take] with -1 beyond a list's length.  None when the global generator is not numpy's legacy MT19937."""
    sizes = np.ascontiguousarray(sizes, np.int32).reshape(-1)
    st = np.random.get_state()
    if st[0] != 'MT19937':
        return None
    key = np.ascontiguousarray(st[1], np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    out = np.empty((sizes.shape[0], int(take)), np.int64)
    if sizes.shape[0]:
        _check(lib().roreg_mt_stream_shuffle_prefix(c_void_p(key.ctypes.data), ctypes.byref(pos), c_void_p(sizes.ctypes.data), sizes.shape[0], int(take),
                                                    c_void_p(out.ctypes.data)), 'roreg_mt_stream_shuffle_prefix')
        np.random.set_state(('MT19937', key, int(pos.value), st[3], st[4]))
    return out


def gather_rows_batch(tasks):
    """tasks [(src [*, ...] contiguous device tensor, rows int64 device [n], dst [n, ...] contiguous device tensor of src's dtype and row
    shape)]: dst[i] = src[rows[i]] for every task, ONE launch (all tasks must have the same row size in bytes, a multiple of 8)."""
    if not tasks:
        return
    row_bytes = int(tasks[0][0][0].numel() * tasks[0][0].element_size())
    table = np.zeros(len(tasks), _GATHER_TASK)
    for i, (src, rows, dst) in enumerate(tasks):
        _ptr(src); _ptr(rows, torch.int64); _ptr(dst)
        if src.dtype != dst.dtype or int(src[0].numel() * src.element_size()) != row_bytes or dst.shape[0] != rows.shape[0]:
            raise HipError('gather_rows_batch: tasks must share the row size, and dst must hold one row per index')
        table[i] = (src.data_ptr(), rows.data_ptr(), dst.data_ptr(), int(rows.shape[0]), 0)
    tdev = upload(table.view(np.uint8).reshape(len(tasks), _GATHER_TASK.itemsize))
    _check(lib().roreg_gather_rows_batch(_ptr(tdev), len(tasks), int(table['n'].max()), row_bytes, _stream()), 'roreg_gather_rows_batch')


def gather_rows_f64(src, rows):
    M, width = rows.shape[0], src.shape[1]
    out = torch.empty((M, width), dtype=torch.float64, device=src.device)
    _check(lib().roreg_gather_rows_f64(_ptr(src, torch.float64), _ptr(rows, torch.int64), M, width, _ptr(out), _stream()), 'roreg_gather_rows_f64')
    return out


# ----------------------------------------------------------------------------------------------------
# the rest of the namespace: the rotation-coherence matcher's bindings and the group-Fourier / split-operand layer bindings live in
# modules of their own (they import the helpers above, so they come last); `hip.<name>` stays the one way to call them
# ----------------------------------------------------------------------------------------------------
from ._hip_matcher import *      # noqa: E402,F401,F403
from ._hip_fourier import *      # noqa: E402,F401,F403
from ._hip_matcher import _rm_op                                                                   # noqa: E402,F401  (underscore names are not star-exported by default ...)
from ._hip_fourier import _bf16_split3, _keypoint_of_columns, _ptr_array, _res_ptr, _tile_cache     # noqa: E402,F401  (... __all__ lists them; kept explicit for readers)
