"""ET_test: local-rotation regressor (mirror of network/eqv_trans.py:78-138).

forward({'before_eqv0','before_eqv1','after_eqv0','after_eqv1': [B,32,60], 'pre_idx': [B]})
  -> {'quaternion_pre': [B,4], 'pre_idxs': [B]}

The reference evaluates its 1x1 head at all 60 group columns and keeps column g=0 (eqv_trans.py:133-136), so
only the group columns that can reach g=0 through the 13-stencil are live: 45 columns of Conv_init's output,
13 of comb_layer_in's, 1 of comb_layer_out's.  The pruned path computes exactly those (5.6x fewer MACs, same
values); `pruned=False` evaluates every column like the reference, for the equality test."""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from ..group import tables
from .ops import Comb_Conv, Residual_Comb_Conv, _Branch, _version_key


class ET_test(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.Conv_init = Comb_Conv(32 * 4, 256)
        self.PartII_SO3_Conv_layers = nn.ModuleList([Residual_Comb_Conv(256, 512, 256)])
        d = [256, 512, 128, 4]
        self.PartII_To_R_dims = d
        self.PartII_To_R_FC = nn.Sequential(
            nn.Conv2d(d[0], d[1], 1, 1), nn.BatchNorm2d(d[1]), nn.ReLU(),
            nn.Conv2d(d[1], d[2], 1, 1), nn.BatchNorm2d(d[2]), nn.ReLU(),
            nn.Conv2d(d[2], d[3], 1, 1))
        self.pruned = True
        self.fourier_init = True
        self.gemm = hip.GEMM_MODE           # 'f16x2' | 'bf16x3' | 'f32' (hip.GEMM_MODE)
        self.packed_trunk = os.environ.get('ROREG_ET_PACKED', '1') != '0'      # f16x2 mode: the trunk convolution reads packed words (see trunk_and_head)

    # ---- kernel plans -------------------------------------------------------------------------------------
    def _head_plans(self):
        fc = self.PartII_To_R_FC
        key = _version_key(fc)
        if getattr(self, '_head_key', None) != key:
            def bn(m):
                return (m.weight, m.bias, m.running_mean, m.running_var)
            self._head = [hip.ConvLayer(fc[0].weight, fc[0].bias, None),
                          hip.ConvLayer(fc[3].weight, fc[3].bias, bn(fc[1]), eps=fc[1].eps),
                          hip.ConvLayer(fc[6].weight, fc[6].bias, bn(fc[4]), eps=fc[4].eps)]
            self._head_key = key
        return self._head

    def _dense_plans(self):
        """Split-mode tail of the network as dense layers on row-major activations (hip.dense_split): the trunk's last layer
        (512 x 13 live columns -> the single column g = 0, the stencil's gather folded into the weight order) and the 1x1 head."""
        res = self.PartII_SO3_Conv_layers[0]
        fc = self.PartII_To_R_FC
        key = _version_key(res.comb_layer_out, fc)
        if getattr(self, '_dense_key', None) != key:
            def fold(bn):
                g, b, m, v = [t.detach().to('cpu', torch.float32).numpy() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
                sc = g / np.sqrt(v + bn.eps)
                return sc, b - m * sc
            T = tables()
            l0, l1 = T.live_sets(2)[0], T.live_sets(2)[1]
            pos1 = {g: i for i, g in enumerate(l1)}
            cols = [pos1[int(v)] for v in T.Nei[l0[0]]]                       # stencil position k reads live column cols[k]
            Wc = res.comb_layer_out[2].weight.detach().to('cpu', torch.float32).numpy()[:, :, 0, :]      # [256,512,13]
            Wd = np.zeros((Wc.shape[0], Wc.shape[1], 13), np.float32)
            for k, col in enumerate(cols):
                Wd[:, :, col] += Wc[:, :, k]
            sc, sh = fold(res.comb_layer_out[0])
            out = hip.DenseSplitLayer(Wd.reshape(Wd.shape[0], -1), res.comb_layer_out[2].bias.detach().cpu().numpy(), np.repeat(sc, 13), np.repeat(sh, 13))
            def w1(conv):
                return conv.weight.detach().to('cpu', torch.float32).numpy()[:, :, 0, 0], conv.bias.detach().cpu().numpy()
            h0 = hip.DenseSplitLayer(*w1(fc[0]))
            h1 = hip.DenseSplitLayer(*w1(fc[3]), *fold(fc[1]))
            h2 = hip.DenseSplitLayer(*w1(fc[6]), *fold(fc[4]))
            self._dense = (out, h0, h1, h2)
            self._dense_key = key
        return self._dense

    LIVE_PAD = 48          # the 45 live columns of Conv_init's output are stored with a 16-byte-friendly stride

    # LDS slot order of the 45 live input columns for the trunk's 13-column stencil convolution (hip.group_conv lds_order): found by
    # tools/lds_perm_search.py for THIS gather table (mean serialisation of the gathered 16-lane ds_read_b128 groups 1.95 instead of 3.21
    # with the natural order at stride 48, where 65 % of the kernel's LDS cycles were bank conflicts).  An execution hint only.
    _LDS_ORDER_45 = (32, 12, 28, 30, 42, 35, 15, 40, 17, 39, 29, 22, 26, 4, 11, 18, 6, 9, 33, 24, 8, 2, 7, 1, 14, 34, 37, 27, 13, 38, 20, 25, 10, 21, 43, 16,
                     36, 0, 19, 5, 41, 3, 31, 23, 44)

    @classmethod
    def _trunk_lds_order(cls):
        """(device int32 [LIVE_PAD] slot table, stride 45): the padding columns 45..47 are never gathered (-1)."""
        order = np.full(cls.LIVE_PAD, -1, np.int32)
        order[:45] = cls._LDS_ORDER_45
        assert sorted(order[:45].tolist()) == list(range(45))
        return hip.gather_table('et_lds_order', order), 45

    @staticmethod
    def _pruned_gathers():
        T = tables()
        live = T.live_sets(2)                         # [ [0], 13, 45 ]
        l0, l1, l2 = live[0], live[1], live[2]
        pos2 = {g: i for i, g in enumerate(l2)}
        pos1 = {g: i for i, g in enumerate(l1)}
        ga = T.Nei[l2]                                                    # [45,13] into the full 60 columns
        gb = np.array([[pos2[int(v)] for v in T.Nei[g]] for g in l1])     # [13,13] into the 45 live columns
        gc = np.array([[pos1[int(v)] for v in T.Nei[g]] for g in l0])     # [1,13]  into the 13 live columns
        gmap = np.full(60, -1, np.int32)
        for g, i in pos2.items():
            gmap[g] = i
        return (hip.gather_table('et_a', ga), hip.gather_table('et_b', gb), hip.gather_table('et_c', gc), pos2[0],
                hip.gather_table('et_gmap', gmap))

    def _fourier_init(self):
        """Conv_init (BN -> ReLU -> group conv 128 -> 256) in the irrep domain; only its 45 live output columns are
        transformed back (244 instead of 45*13 = 585 multiply-adds per channel pair)."""
        from .gf_fourier import _Layer, _fold_bn
        key = _version_key(self.Conv_init)
        if getattr(self, '_finit_key', None) != key:
            self._finit = (_Layer(self.Conv_init.comb_layer[2]), _fold_bn(self.Conv_init.comb_layer[0]))
            self._finit_key = key
        return self._finit

    def _trunk_bn(self):
        """((scale, shift) of the trunk convolution's folded BatchNorm, the (u, v) tables of the bound Conv_init's GEMM propagates to it)."""
        from .gf_fourier import _fold_bn
        res = self.PartII_SO3_Conv_layers[0]
        key = _version_key(res.comb_layer_in[0], self.Conv_init)
        if getattr(self, '_tbn_key', None) != key:
            bn = _fold_bn(res.comb_layer_in[0])
            self._tbn = (bn, hip.next_bound_spatial(bn, self._fourier_init()[0].bias))
            self._tbn_key = key
        return self._tbn

    def assemble(self, data):
        """x [B,128,60] = cat(before0[P[pre]], before1, after0[P[pre]], after1)  (eqv_trans.py:126-129)."""
        dev = 'cuda'
        b0 = data['before_eqv0'].to(dev, torch.float32).contiguous(); b1 = data['before_eqv1'].to(dev, torch.float32).contiguous()
        a0 = data['after_eqv0'].to(dev, torch.float32).contiguous(); a1 = data['after_eqv1'].to(dev, torch.float32).contiguous()
        pre = data['pre_idx'].to(dev, torch.int64).contiguous()
        # et_gather(before0, before1, after0, after1): channels = [before1-side permuted, before0-side, ...] with
        # "1" = the permuted side; in the reference's batch the permuted side is before_eqv0/after_eqv0.
        return hip.et_gather(b1, b0, a1, a0, pre), pre

    def conv_init_bn(self):
        """(scale, shift) device tensors of Conv_init's folded BatchNorm when the fp16 x 2 irrep path will use them, else None
        (lets the caller compute the rows' block-scale bound while it assembles x: hip.LtBatch.prepare(bound_bn=...))."""
        if self.pruned and self.fourier_init and self.gemm == 'f16x2':
            return self._fourier_init()[1]
        return None

    def trunk_and_head(self, x, x_bound=None):
        """x [B,128,60] -> un-normalised quaternion [B,4].  x_bound: hip.row_bound(x, Conv_init's BatchNorm) if the caller has it."""
        res = self.PartII_SO3_Conv_layers[0]
        h0p, h1p, h2p = self._head_plans()
        if self.pruned:
            ga, gb, gc, p0, gmap = self._pruned_gathers()
            B = x.shape[0]
            if self.fourier_init:
                layer, bn = self._fourier_init()
                hip.ensure_fourier()
                sp = self.gemm != 'f32'
                if self.gemm == 'f16x2':
                    # fp16 x 2 all the way, every block scale PER ROW (correspondence): Conv_init's coefficients are split under the
                    # row's own bound (hip.row_bound), every later kernel tracks max |output row| on the device as the next kernel's scale,
                    # so a correspondence's quaternion does not depend on which other correspondences share the batch
                    b0 = x_bound if x_bound is not None else hip.row_bound(x, bn=bn)
                    X0 = hip.ft_nonlin(B, 128, x_spatial=x, bn=bn, split='f16x2', out_bound=b0, planes=hip.use_planes(256))      # half-block layout: LDS-DMA GEMM
                    d_out, d0, d1, d2 = self._dense_plans()
                    if self.packed_trunk:
                        # The trunk convolution's operand leaves the inverse transform READY: ReLU(BN(h)) as fp16 hi / lo words under the row's block
                        # scale -- from a bound the GEMM's epilogue propagates (sqrt(60) |scale| |T| + |scale| |bias| + |shift| >= |ReLU(BN(IFT(T) + bias))|),
                        # so it exists before h does -- and the convolution's staging only regroups halves (it spent 4.4 vector instructions per MFMA
                        # on BatchNorm + conversion + split of the float tensor, profiles/r05_et_conv_pmc.txt).  The identity short cut's column g = 0
                        # of h (before BatchNorm) comes out beside the words.
                        bn_t, nb_t = self._trunk_bn()
                        T0, b1 = hip.irrep_gemm(X0, None, 128, 256, B, f16x2=layer.wsplit2, x_bound=b0, next_bound=nb_t, x_planes=hip.use_planes(256))
                        del X0
                        hw, h0 = hip.ft_nonlin_packed(B, 256, T0, layer.bias, bn_t, b1, g_map=gmap, Lout=self.LIVE_PAD, Lvalid=45, raw_g=0)   # [B,256,48] words, [B,256]
                        del T0
                        m, am = hip.group_conv_packed(hw, res._b_in.plan(), b1, gb, want_rowmax=True, lds_order=self._trunk_lds_order())       # [B,512,13]
                        t, at = hip.dense_split(m.view(B, -1), d_out, residual=h0, in_rowmax=am, want_rowmax=True)                             # [B,256]
                    else:
                        T0 = hip.irrep_gemm(X0, None, 128, 256, B, f16x2=layer.wsplit2, x_bound=b0, x_planes=hip.use_planes(256))
                        del X0
                        h, ah = hip.ft_nonlin(B, 256, coef_in=T0, bias=layer.bias, spatial_out=True, g_map=gmap, Lout=self.LIVE_PAD, Lvalid=45, split='f16x2',
                                              want_rowmax=True)                                                          # [B,256,48]
                        del T0
                        m, am = res._b_in(h, gather=gb, in_rowmax=ah, want_rowmax=True, lds_order=self._trunk_lds_order())   # [B,512,13]
                        # identity short cut = column g = 0 of h, read in place (element (b, o) at stride LIVE_PAD from h[0, 0, p0])
                        t, at = hip.dense_split(m.view(B, -1), d_out, residual=h[:, :, p0:], residual_stride=self.LIVE_PAD, in_rowmax=am, want_rowmax=True)   # [B,256]
                    z, az = hip.dense_split(t, d0, in_rowmax=at, want_rowmax=True)
                    z, az = hip.dense_split(z, d1, in_rowmax=az, want_rowmax=True)
                    return hip.dense_split(z, d2, in_rowmax=az)                                                      # [B,4]
                X0 = hip.ft_nonlin(B, 128, x_spatial=x, bn=bn, split=sp)
                T0 = hip.irrep_gemm(X0, layer.wpack, 128, 256, B, split=layer.wsplit if sp else None)
                del X0
                h = hip.ft_nonlin(B, 256, coef_in=T0, bias=layer.bias, spatial_out=True, g_map=gmap, Lout=self.LIVE_PAD, Lvalid=45, split=sp)   # [B,256,48]
                del T0
            else:
                h = self.Conv_init(x, gather=ga)                               # [B,256,45]
            m = res._b_in(h, gather=gb, split=self.gemm != 'f32',
                          lds_order=self._trunk_lds_order() if (self.fourier_init and self.gemm != 'f32') else None)       # [B,512,13]
            sc = h[:, :, p0:p0 + 1].contiguous()                               # identity short cut at g=0
            if self.gemm != 'f32':                                             # trunk tail + head as dense split layers
                d_out, d0, d1, d2 = self._dense_plans()
                t = hip.dense_split(m.view(B, -1), d_out, residual=sc.view(B, -1))      # [B,256]
                return hip.dense_split(hip.dense_split(hip.dense_split(t, d0), d1), d2)  # [B,4]
            t = res._b_out(m, gather=gc, residual=sc)                          # [B,256,1]
        else:
            h = self.Conv_init(x)
            t = res(h)[:, :, 0:1].contiguous()                                 # [B,256,1] (column g=0)
        z = hip.group_conv(t, h0p)
        z = hip.group_conv(z, h1p)
        z = hip.group_conv(z, h2p)                                             # [B,4,1]
        return z[:, :, 0].contiguous()

    def forward(self, data):
        x, pre = self.assemble(data)
        q = self.trunk_and_head(x)
        q = q / torch.norm(q, dim=1)[:, None]
        return {'quaternion_pre': q, 'pre_idxs': pre}
