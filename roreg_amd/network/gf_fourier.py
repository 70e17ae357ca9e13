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

    def _plan_bounds(self):
        """(u, v) tables of the bound each GEMM epilogue propagates to the transform that follows it (hip.next_bound)."""
        if getattr(self, '_nb_key', None) != self._key:
            self.nb_1 = hip.next_bound(self.bn_1, self.l_in.bias)
            self.nb_2 = hip.next_bound(self.bn_2, self.l_1.bias)
            self.nb_3 = hip.next_bound(self.bn_3, self.l_2.bias, self.l_in.bias)
            self._nb_key = self._key

    def forward_raw(self, x):
        """x [B,32,60] device float32 -> eqv_raw = conv stack(x) + x  [B,32,60]."""
        self._plan()
        hip.ensure_fourier()
        B = x.shape[0]
        sp = {'f32': False, 'bf16x3': True, 'f16x2': 'f16x2'}[self.gemm]      # matrix-core mode of the transforms
        if self.gemm == 'f16x2':
            # fp16 x 2 GEMMs: every coefficient tensor is written already split (fp16 hi/lo words) under a PER-KEYPOINT power-of-two
            # scale; the scale comes from a bound that exists before the tensor does -- from the group-domain input (row_bound) or
            # propagated by the previous GEMM's epilogue (next_bound) -- so a keypoint's result depends on that keypoint alone.
            self._plan_bounds()
            b0 = hip.row_bound(x)
            X0 = hip.ft_nonlin(B, 32, x_spatial=x, split=sp, out_bound=b0)
            T0, b1 = hip.irrep_gemm(X0, None, 32, 256, B, f16x2=self.l_in.wsplit2, x_bound=b0, next_bound=self.nb_1)
            del X0
            # the two big layers' operands in half-block (hi | lo per 32 columns) layout: their activations reach LDS by LDS-DMA (hip.XDMA; csrc/fourier.hip irrep_gemm_xdma_kernel)
            xd = hip.use_planes(512) and hip.use_planes(256)
            X1 = hip.ft_nonlin(B, 256, coef_in=T0, bias=self.l_in.bias, bn=self.bn_1, split=sp, out_bound=b1, planes=xd)
            T1, b2 = hip.irrep_gemm(X1, None, 256, 512, B, f16x2=self.l_1.wsplit2, x_bound=b1, next_bound=self.nb_2, x_planes=xd)
            del X1
            X2 = hip.ft_nonlin(B, 512, coef_in=T1, bias=self.l_1.bias, bn=self.bn_2, split=sp, out_bound=b2, planes=xd)
            del T1
            T2, b3 = hip.irrep_gemm(X2, None, 512, 256, B, f16x2=self.l_2.wsplit2, x_bound=b2, next_bound=self.nb_3, add=T0, x_planes=xd)   # + identity short cut
            del X2, T0
            X3 = hip.ft_nonlin(B, 256, coef_in=T2, bias=self.l_2.bias, bias2=self.l_in.bias, bn=self.bn_3, split=sp, out_bound=b3)
            del T2
            T3 = hip.irrep_gemm(X3, None, 256, 32, B, f16x2=self.l_out.wsplit2, x_bound=b3)
            return hip.ft_nonlin(B, 32, coef_in=T3, bias=self.l_out.bias, resid_spatial=x, spatial_out=True, split=sp)

        def gemm(X, layer, C, O, add=None):
            return hip.irrep_gemm(X, layer.wpack, C, O, B, split=layer.wsplit if self.gemm == 'bf16x3' else None, add=add)

        def ft(C, **kw):
            return hip.ft_nonlin(B, C, split=sp, **kw)
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
        return ft(32, coef_in=T3, bias=self.l_out.bias, resid_spatial=x, spatial_out=True)


    def scale_headroom(self, x):
        """Diagnostics for the fp16 x 2 mode (bench.py `f16x2_scale_headroom_bits`): per GEMM input of the extractor, how many binades the
        per-keypoint block-scale BOUND (row_bound / the previous GEMM's propagated next_bound) sits above the keypoint's true coefficient
        maximum -- log2(bound / max|coef|), min / mean / max over the keypoints of x [B,32,60].  The hi + lo fp16 split resolves 22 bits
        below the bound, so this is the precision the conservative bound gives away relative to the data."""
        self._plan(); self._plan_bounds()
        hip.ensure_fourier()
        B = x.shape[0]
        out = {}

        def report(name, bound, coefs_f32, C):
            actual = hip.pack_coefs_f16x2(coefs_f32, C, B)[1][:B]
            bits = torch.log2(bound[:B].double() / actual.double().clamp_min(1e-300))
            out[name] = {'min': round(float(bits.min()), 2), 'mean': round(float(bits.mean()), 2), 'max': round(float(bits.max()), 2)}
        b0 = hip.row_bound(x)
        report('gemm_32_256_input', b0, hip.ft_nonlin(B, 32, x_spatial=x, split=True), 32)
        X0 = hip.ft_nonlin(B, 32, x_spatial=x, split='f16x2', out_bound=b0)
        T0, b1 = hip.irrep_gemm(X0, None, 32, 256, B, f16x2=self.l_in.wsplit2, x_bound=b0, next_bound=self.nb_1)
        report('gemm_256_512_input', b1, hip.ft_nonlin(B, 256, coef_in=T0, bias=self.l_in.bias, bn=self.bn_1, split=True), 256)
        X1 = hip.ft_nonlin(B, 256, coef_in=T0, bias=self.l_in.bias, bn=self.bn_1, split='f16x2', out_bound=b1)
        T1, b2 = hip.irrep_gemm(X1, None, 256, 512, B, f16x2=self.l_1.wsplit2, x_bound=b1, next_bound=self.nb_2)
        report('gemm_512_256_input', b2, hip.ft_nonlin(B, 512, coef_in=T1, bias=self.l_1.bias, bn=self.bn_2, split=True), 512)
        X2 = hip.ft_nonlin(B, 512, coef_in=T1, bias=self.l_1.bias, bn=self.bn_2, split='f16x2', out_bound=b2)
        T2, b3 = hip.irrep_gemm(X2, None, 512, 256, B, f16x2=self.l_2.wsplit2, x_bound=b2, next_bound=self.nb_3, add=T0)
        report('gemm_256_32_input', b3, hip.ft_nonlin(B, 256, coef_in=T2, bias=self.l_2.bias, bias2=self.l_in.bias, bn=self.bn_3, split=True), 256)
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
        if self.gemm == 'f16x2':                                  # per-keypoint block scales, see FourierGF.forward_raw
            if getattr(self, '_nb_key', None) != self._key:
                self.nb_out = hip.next_bound(self.bn_out, self.l_in.bias)
                self._nb_key = self._key
            bs = hip.row_bound(x, bn=self.bn_sc)
            Xs = hip.ft_nonlin(B, 32, x_spatial=x, bn=self.bn_sc, split=sp, out_bound=bs)
            S = hip.irrep_gemm(Xs, None, self.l_sc.C, self.l_sc.O, B, f16x2=self.l_sc.wsplit2, x_bound=bs)
            del Xs
            ba = hip.row_bound(x, bn=self.bn_in)
            Xa = hip.ft_nonlin(B, 32, x_spatial=x, bn=self.bn_in, split=sp, out_bound=ba)
            T1, b1 = hip.irrep_gemm(Xa, None, self.l_in.C, self.l_in.O, B, f16x2=self.l_in.wsplit2, x_bound=ba, next_bound=self.nb_out)
            del Xa
            X1 = hip.ft_nonlin(B, 64, coef_in=T1, bias=self.l_in.bias, bn=self.bn_out, split=sp, out_bound=b1)
            del T1
            T2 = hip.irrep_gemm(X1, None, self.l_out.C, self.l_out.O, B, f16x2=self.l_out.wsplit2, x_bound=b1, add=S)
            del X1, S
            return hip.ft_nonlin(B, 16, coef_in=T2, bias=self.l_out.bias, bias2=self.l_sc.bias, spatial_out=True, split=sp)

        def gemm(X, layer, add=None):
            return hip.irrep_gemm(X, layer.wpack, layer.C, layer.O, B, split=layer.wsplit if self.gemm == 'bf16x3' else None, add=add)

        def ft(C, **kw):
            return hip.ft_nonlin(B, C, split=sp, **kw)
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
        return ft(16, coef_in=T2, bias=self.l_out.bias, bias2=self.l_sc.bias, spatial_out=True)
