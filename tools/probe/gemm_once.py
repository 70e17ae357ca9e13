"""A few launches of the big fp16 x 2 irrep GEMM (256 -> 512 channels, 61440 keypoints) for profiler runs.  Usage: gemm_once.py [random|zeros] [launches]"""
import sys
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
B = 61440; C, O = 256, 512
torch.manual_seed(0)
kind = sys.argv[1] if len(sys.argv) > 1 else 'random'
L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
n = hip.coef_size(C, B)
X = torch.randn(n, device='cuda') if kind == 'random' else torch.zeros(n, device='cuda')
Xp, xb = hip.pack_coefs_f16x2(X, C, B)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb)
torch.cuda.synchronize()
