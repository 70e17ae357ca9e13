"""GF in the irrep domain: same parameters and network function as Group_feat_network (network/group_feat.py:7-45), evaluated
with roreg_irrep_gemm / roreg_ft_nonlin (3.2x fewer multiply-adds than the direct 13-stencil group conv).

Pipeline for x [B,32,60]:
   X0 = FT(x)                        T0 = GEMM(X0, Conv_in)
   X1 = FT(relu(bn1(IFT(T0)+b_in)))  T1 = GEMM(X1, comb_layer_in)
   X2 = FT(relu(bn2(IFT(T1)+b_1)))   T2 = GEMM(X2, comb_layer_out)
   X3 = FT(relu(bn3(IFT(T2+T0)+b_2+b_in)))          (identity short cut added in the Fourier domain)
   T3 = GEMM(X3, Conv_out)           eqv_raw = IFT(T3) + b_out + x
"""
import numpy as np
import torch

from .. import hip
from ..fourier import group_fourier, DIMS
from .ops import _version_key


def _fold_bn(bn):
    scale = bn.weight.detach().float().cpu() / torch.sqrt(bn.running_var.detach().float().cpu() + bn.eps)
    shift = bn.bias.detach().float().cpu() - bn.running_mean.detach().float().cpu() * scale
    return scale.cuda().contiguous(), shift.cuda().contiguous()


class _Layer:
    def __init__(self, conv):
        gf = group_fourier()
        W = conv.weight.detach().double().cpu().numpy()
        O, C = W.shape[0], W.shape[1]
        W = W.reshape(O, C, 13)
        self.C, self.O = C, O
        self.bias = conv.bias.detach().float().cuda().contiguous()
        self.wpack = []
        self.wsplit = []          # 3 x bf16 pieces for the f32-accurate GEMM on the bf16 matrix cores
        self.dense = []           # the five padded float32 matrices [round_up(d*O,128), d*C]
        for ri, Wh in enumerate(gf.transform_weights(W)):                 # [O,C,l,j]
            d = DIMS[ri]
            Wm = np.ascontiguousarray(Wh.transpose(3, 0, 2, 1)).reshape(d * O, d * C)     # rows (j,o), cols (l,c)
            Mpad = (d * O + 127) // 128 * 128
            Wp = np.zeros((Mpad, d * C), np.float32)
            Wp[:d * O] = Wm.astype(np.float32)
            self.dense.append(Wp)
            self.wpack.append(hip.pack_conv_weights(torch.from_numpy(Wp).reshape(Mpad, d * C, 1)))
            self.wsplit.append(hip.bf16_split3_pack(Wp) if (d * C) % 16 == 0 else None)
        # fp16 x 2 pieces with one power-of-two scale for the whole layer (the five GEMMs are one launch)
        w_exp = hip.f16_scale_exp(max(float(np.abs(Wp).max()) for Wp in self.dense))
        self.wsplit2 = ([hip.f16_split2_pack(Wp, w_exp) for Wp in self.dense], w_exp) if all(Wp.shape[1] % 16 == 0 for Wp in self.dense) else None


class FourierGF:
    def __init__(self, net):
        """net: Group_feat_network (parameter container)."""
        self.net = net
        self._key = None
        self.gemm = hip.GEMM_MODE       # matrix-core mode of the GEMMs: 'f16x2' | 'bf16x3' | 'f32' (hip.GEMM_MODE); transforms: bf16x3 unless 'f32'

    def _plan(self):
        key = _version_key(self.net)
        if self._key != key:
            res = self.net.SO3_Conv_layers[0]
            self.l_in = _Layer(self.net.Conv_in[0])
            self.l_1 = _Layer(res.comb_layer_in[2]); self.bn_1 = _fold_bn(res.comb_layer_in[0])
            self.l_2 = _Layer(res.comb_layer_out[2]); self.bn_2 = _fold_bn(res.comb_layer_out[0])
            self.l_out = _Layer(self.net.Conv_out.comb_layer[2]); self.bn_3 = _fold_bn(self.net.Conv_out.comb_layer[0])
            self._key = key

    def forward_raw(self, x):
        """x [B,32,60] device float32 -> eqv_raw = conv stack(x) + x  [B,32,60]."""
        self._plan()
        hip.ensure_fourier()
        B = x.shape[0]
        sp = {'f32': False, 'bf16x3': True, 'f16x2': 'f16x2'}[self.gemm]      # matrix-core mode of the transforms
        f16 = self.gemm == 'f16x2'                    # GEMMs with fp16 x 2 operands: every transform also tracks max |coefficient|

        def gemm(Xa, layer, C, O, add=None):
            X, amax = Xa
            if f16:
                return hip.irrep_gemm(X, layer.wpack, C, O, B, f16x2=layer.wsplit2, x_absmax=amax, add=add)
            return hip.irrep_gemm(X, layer.wpack, C, O, B, split=layer.wsplit if self.gemm == 'bf16x3' else None, add=add)

        def ft(C, **kw):
            r = hip.ft_nonlin(B, C, split=sp, want_absmax=f16, **kw)
            return r if f16 else (r, None)
        X0 = ft(32, x_spatial=x)
        T0 = gemm(X0, self.l_in, 32, 256)
        del X0
        X1 = ft(256, coef_in=T0, bias=self.l_in.bias, bn=self.bn_1)
        T1 = gemm(X1, self.l_1, 256, 512)
        del X1
        X2 = ft(512, coef_in=T1, bias=self.l_1.bias, bn=self.bn_2)
        del T1
        T2 = gemm(X2, self.l_2, 512, 256, add=T0)                            # + identity short cut
        del X2, T0
        X3 = ft(256, coef_in=T2, bias=self.l_2.bias, bias2=self.l_in.bias, bn=self.bn_3)
        del T2
        T3 = gemm(X3, self.l_out, 256, 32)
        out = hip.ft_nonlin(B, 32, coef_in=T3, bias=self.l_out.bias, resid_spatial=x, spatial_out=True, split=sp)
        return out


class FourierRD:
    """The detector's Residual_Comb_Conv(32, 64, 16) with its conv short cut (network/rot_detect.py:39, network/ops.py:22-64) in the
    irrep domain:
        Xa = FT(relu(bn_in(x)))    T1 = GEMM(Xa, comb_layer_in)         Xs = FT(relu(bn_sc(x)))   S = GEMM(Xs, short_cut_layer)
        X1 = FT(relu(bn_out(IFT(T1) + b_in)))                            T2 = GEMM(X1, comb_layer_out) + S
        enc = IFT(T2) + b_out + b_sc                                     [B,16,60]"""

    def __init__(self, block):
        self.block = block
        self._key = None
        self.gemm = hip.GEMM_MODE

    def _plan(self):
        key = _version_key(self.block)
        if self._key != key:
            b = self.block
            self.l_in = _Layer(b.comb_layer_in[2]); self.bn_in = _fold_bn(b.comb_layer_in[0])
            self.l_out = _Layer(b.comb_layer_out[2]); self.bn_out = _fold_bn(b.comb_layer_out[0])
            self.l_sc = _Layer(b.short_cut_layer[2]); self.bn_sc = _fold_bn(b.short_cut_layer[0])
            self._key = key

    def forward(self, x):
        self._plan()
        hip.ensure_fourier()
        B = x.shape[0]
        sp = {'f32': False, 'bf16x3': True, 'f16x2': 'f16x2'}[self.gemm]
        f16 = self.gemm == 'f16x2'

        def gemm(Xa, layer, add=None):
            X, amax = Xa
            if f16:
                return hip.irrep_gemm(X, layer.wpack, layer.C, layer.O, B, f16x2=layer.wsplit2, x_absmax=amax, add=add)
            return hip.irrep_gemm(X, layer.wpack, layer.C, layer.O, B, split=layer.wsplit if self.gemm == 'bf16x3' else None, add=add)

        def ft(C, **kw):
            r = hip.ft_nonlin(B, C, split=sp, want_absmax=f16, **kw)
            return r if f16 else (r, None)
        Xs = ft(32, x_spatial=x, bn=self.bn_sc)
        S = gemm(Xs, self.l_sc)
        del Xs
        Xa = ft(32, x_spatial=x, bn=self.bn_in)
        T1 = gemm(Xa, self.l_in)
        del Xa
        X1 = ft(64, coef_in=T1, bias=self.l_in.bias, bn=self.bn_out)
        del T1
        T2 = gemm(X1, self.l_out, add=S)
        del X1, S
        return hip.ft_nonlin(B, 16, coef_in=T2, bias=self.l_out.bias, bias2=self.l_sc.bias, spatial_out=True, split=sp)
