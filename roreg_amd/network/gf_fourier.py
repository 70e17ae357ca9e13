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
        for ri, Wh in enumerate(gf.transform_weights(W)):                 # [O,C,l,j]
            d = DIMS[ri]
            Wm = np.ascontiguousarray(Wh.transpose(3, 0, 2, 1)).reshape(d * O, d * C)     # rows (j,o), cols (l,c)
            Mpad = (d * O + 127) // 128 * 128
            Wp = np.zeros((Mpad, d * C), np.float32)
            Wp[:d * O] = Wm.astype(np.float32)
            self.wpack.append(hip.pack_conv_weights(torch.from_numpy(Wp).reshape(Mpad, d * C, 1)))
            self.wsplit.append(hip.bf16_split3_pack(Wp) if (d * C) % 16 == 0 else None)


class FourierGF:
    def __init__(self, net):
        """net: Group_feat_network (parameter container)."""
        self.net = net
        self._key = None
        self.split_bf16 = hip.GEMM_MODE == 'split'       # GEMMs / transforms as 3 x bf16 split products (f32-accurate), see hip.GEMM_MODE

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
        sp = self.split_bf16
        X0 = hip.ft_nonlin(B, 32, x_spatial=x, split=sp)
        T0 = hip.irrep_gemm(X0, self.l_in.wpack, 32, 256, B, split=self.l_in.wsplit if sp else None)
        X1 = hip.ft_nonlin(B, 256, coef_in=T0, bias=self.l_in.bias, bn=self.bn_1, split=sp)
        T1 = hip.irrep_gemm(X1, self.l_1.wpack, 256, 512, B, split=self.l_1.wsplit if sp else None)
        del X1
        X2 = hip.ft_nonlin(B, 512, coef_in=T1, bias=self.l_1.bias, bn=self.bn_2, split=sp)
        del T1
        T2 = hip.irrep_gemm(X2, self.l_2.wpack, 512, 256, B, split=self.l_2.wsplit if sp else None, add=T0)   # + identity short cut
        del X2, T0
        X3 = hip.ft_nonlin(B, 256, coef_in=T2, bias=self.l_2.bias, bias2=self.l_in.bias, bn=self.bn_3, split=sp)
        del T2
        T3 = hip.irrep_gemm(X3, self.l_out.wpack, 256, 32, B, split=self.l_out.wsplit if sp else None)
        out = hip.ft_nonlin(B, 32, coef_in=T3, bias=self.l_out.bias, resid_spatial=x, spatial_out=True, split=sp)
        return out
