"""detector_eqv_test: rotation-guided detector (mirror of network/rot_detect.py:36-55).
forward({'feats': [B,32,60]}) -> {'scores': [B]} (raw std scores; ranking happens in test/detector.py)."""
import torch
import torch.nn as nn

from .. import hip
from .ops import Residual_Comb_Conv


class detector_eqv_test(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.eqv_encoder = nn.ModuleList([Residual_Comb_Conv(32, 64, 16)])

        self.mode = 'fourier'          # 'fourier': irrep-domain evaluation (gf_fourier.FourierRD); 'direct': the 13-stencil kernels
        object.__setattr__(self, '_fourier', None)

    def encode(self, feats):
        x = feats.to('cuda', torch.float32).contiguous()
        if self.mode == 'fourier' and x.shape[0] > 0:
            if self._fourier is None:
                from .gf_fourier import FourierRD
                object.__setattr__(self, '_fourier', FourierRD(self.eqv_encoder[0]))
            return self._fourier.forward(x)            # [B,16,60]
        return self.eqv_encoder[0](x)                  # [B,16,60]

    def forward(self, batch):
        enc = self.encode(batch['feats'])
        return {'scores': hip.det_score(enc)}
