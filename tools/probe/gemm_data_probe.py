import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
C, O = 256, 512
L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
B = 64512
n = hip.coef_size(C, B)
g = torch.Generator(device='cuda').manual_seed(1)
base = torch.randn(n, device='cuda', generator=g)
for name, X in (('randn', base), ('randn*exp(randn)', base * torch.exp(torch.randn(n, device='cuda', generator=g))),
                ('randn*1e-3 with one 1e3 outlier', None), ('randn*exp(2*randn)', base * torch.exp(2 * torch.randn(n, device='cuda', generator=g)))):
    if X is None:
        X = base * 1e-3; X[12345] = 1e3
    amax = X.abs().max().reshape(1).float()
    for mode, kw in (('f16x2', dict(f16x2=L.wsplit2, x_absmax=amax)), ('bf16x3', dict(split=L.wsplit))):
        for _ in range(2):
            out = hip.irrep_gemm(X, L.wpack, C, O, B, **kw)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            out = hip.irrep_gemm(X, L.wpack, C, O, B, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(f'{name:34s} {mode}: {dt*1e3:7.2f} ms  amax {float(amax):.3g}', flush=True)
