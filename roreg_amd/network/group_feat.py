"""GF_test: the icosahedral group-conv descriptor embedder (mirror of network/group_feat.py:7-45,80-87).

forward(x [B,32,60] float32) -> {'inv': [B,32], 'eqv': [B,32,60]}, device tensors.  Four launches of the MFMA
group-conv kernel (Conv_in 32->256, 256->512, 512->256 + identity short cut, Conv_out 256->32 + input
residual) and one finalize kernel (L2 normalisation over the 32 channels, invariant mean)."""
import torch
import torch.nn as nn

from .. import hip
from .ops import Comb_Conv, Residual_Comb_Conv, _Branch


class Group_feat_network(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.Conv_in = nn.Sequential(nn.Conv2d(32, 256, (1, 13), 1))
        self.SO3_Conv_layers = nn.ModuleList([Residual_Comb_Conv(256, 512, 256)])
        self.Conv_out = Comb_Conv(256, 32)
        object.__setattr__(self, '_b_in', _Branch(self.Conv_in))
        # 'fourier': irrep-domain evaluation (3.2x fewer MACs, same function); 'direct': the 13-stencil MFMA group conv
        object.__setattr__(self, 'mode', 'fourier')
        object.__setattr__(self, '_fourier', None)

    def forward(self, feats, want_inv=True, out_dtype=torch.float32):
        """feats float32 or bfloat16 [B,32,60] (bfloat16: read as stored, float32 arithmetic); out_dtype: storage of 'eqv'."""
        if feats.dim() != 3 or feats.shape[1:] != (32, 60):
            raise ValueError(f'GF expects [B,32,60], got {tuple(feats.shape)}')
        x = feats.to('cuda', torch.bfloat16 if feats.dtype == torch.bfloat16 else torch.float32).contiguous()
        if self.mode == 'fourier':
            if self._fourier is None:
                from .gf_fourier import FourierGF
                object.__setattr__(self, '_fourier', FourierGF(self))
            raw = self._fourier.forward_raw(x)
        else:
            x = x.float()
            h = self._b_in(x)
            for layer in self.SO3_Conv_layers:
                h = layer(h)
            raw = self.Conv_out(h, residual=x)          # feats_eqv + feats  (group_feat.py:37)
        eqv, inv = hip.gf_finalize(raw, want_inv=want_inv, out_dtype=out_dtype)
        return {'inv': inv, 'eqv': eqv}


class GF_test(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.PartI_net = Group_feat_network(cfg)

    def forward(self, group_feat):
        return self.PartI_net(group_feat)
