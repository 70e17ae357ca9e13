"""Group-conv building blocks: same class names, sub-module names and state_dict keys as network/ops.py:11-64,
so the reference's checkpoints load unchanged.  The torch sub-modules (BatchNorm2d / Conv2d) are parameter
containers only: forward() packs them once into kernel layout (hip.ConvLayer) and runs the HIP group
convolution on [B,C,L] activations -- the [B,C,60,13] gathered tensor of the reference is never built."""
import torch
import torch.nn as nn

from .. import hip


def _version_key(*mods):
    key = []
    for m in mods:
        for t in list(m.parameters()) + list(m.buffers()):
            key.append((t.data_ptr(), t._version))
    return tuple(key)


class _PlannedConv(nn.Module):
    """Base: caches a hip.ConvLayer built from (optional BatchNorm2d, Conv2d); rebuilt when parameters change."""

    def _seq(self):
        raise NotImplementedError

    def plan(self):
        seq = self._seq()
        conv = seq[-1]
        bn = seq[0] if isinstance(seq[0], nn.BatchNorm2d) else None
        key = _version_key(seq)
        if getattr(self, '_plan_key', None) != key:
            bnp = (bn.weight, bn.bias, bn.running_mean, bn.running_var) if bn is not None else None
            self._plan = hip.ConvLayer(conv.weight, conv.bias, bnp, eps=bn.eps if bn is not None else 1e-5)
            self._plan_key = key
        return self._plan

    def forward(self, x, gather=None, residual=None, split=False, in_rowmax=None, want_rowmax=False, lds_order=None):
        return hip.group_conv(x, self.plan(), gather=gather, residual=residual, split=split, in_rowmax=in_rowmax, want_rowmax=want_rowmax,
                              lds_order=lds_order)


class Comb_Conv(_PlannedConv):
    """BN(eval) -> ReLU -> Conv2d(in,out,(1,13))  (network/ops.py:11-20)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.comb_layer = nn.Sequential(nn.BatchNorm2d(in_dim), nn.ReLU(), nn.Conv2d(in_dim, out_dim, (1, 13), 1))

    def _seq(self):
        return self.comb_layer


class _Branch(_PlannedConv):
    def __init__(self, seq):
        super().__init__()
        object.__setattr__(self, '_s', seq)       # not registered: the owner registers it under the reference's name

    def _seq(self):
        return self._s


class Residual_Comb_Conv(nn.Module):
    """comb_layer_in -> comb_layer_out (+ identity or conv short cut)  (network/ops.py:22-64)."""

    def __init__(self, in_dim, middle_dim, out_dim, Nei_in_SO3=None):
        super().__init__()
        self.comb_layer_in = nn.Sequential(nn.BatchNorm2d(in_dim), nn.ReLU(), nn.Conv2d(in_dim, middle_dim, (1, 13), 1))
        self.comb_layer_out = nn.Sequential(nn.BatchNorm2d(middle_dim), nn.ReLU(), nn.Conv2d(middle_dim, out_dim, (1, 13), 1))
        self.short_cut = in_dim != out_dim
        if self.short_cut:
            self.short_cut_layer = nn.Sequential(nn.BatchNorm2d(in_dim), nn.ReLU(), nn.Conv2d(in_dim, out_dim, (1, 13), 1))
        self._b_in = _Branch(self.comb_layer_in)
        self._b_out = _Branch(self.comb_layer_out)
        self._b_sc = _Branch(self.short_cut_layer) if self.short_cut else None

    # the helper branches hold no parameters of their own; keep them out of state_dict()/children()
    def __setattr__(self, name, value):
        if name in ('_b_in', '_b_out', '_b_sc'):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def forward(self, x):
        h = self._b_in(x)
        sc = self._b_sc(x) if self.short_cut else x
        return self._b_out(h, residual=sc)
