"""A/B of two GEMM loop variants in separate processes is needed (env read once) -- this script runs one variant and saves a checksum."""
import sys, hashlib
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
B = 61440
for (C, O) in ((256, 512), (256, 32), (32, 256), (128, 256)):          # GF's big layer, its two thin ones (4-wave and 8-wave tiles), ET's Conv_init
  torch.manual_seed(0)
  L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
  n = hip.coef_size(C, B)
  X = torch.randn(n, device='cuda') * torch.exp(torch.randn(n, device='cuda'))
  Xp, xb = hip.pack_coefs_f16x2(X, C, B)
  bn = (torch.rand(O, device='cuda') + 0.5, torch.randn(O, device='cuda') * 0.1)
  nb = hip.next_bound(bn, L.bias)
  planes = hip.XDMA and O % 256 == 0                             # ROREG_GEMM_XDMA=1: the plane layout + LDS-DMA kernel (bitwise the same result)
  if planes:
    Xp = hip.words_to_planes(Xp, C, B)
  out, bound = hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb, x_planes=planes)
  torch.cuda.synchronize()
  print('checksum', C, O, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(bound.cpu().numpy().tobytes()).hexdigest()[:16])
