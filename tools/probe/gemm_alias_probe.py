"""Probe: does the irrep GEMM's time depend on where its operands sit in memory?  X is placed at different offsets inside a larger buffer."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
C, O = 256, 512
L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64512
n = hip.coef_size(C, B)
big = torch.randn(n + (1 << 24), device='cuda')
for off in (0, 128, 1024, 4096 + 128, 65536, 65536 + 384, 1 << 20, (1 << 20) + 128, (1 << 22) + 4096, (1 << 23) + 128 * 77):
    X = big[off:off + n]
    amax = X.abs().max().reshape(1).float()
    for _ in range(2):
        out = hip.irrep_gemm(X, L.wpack, C, O, B, f16x2=L.wsplit2, x_absmax=amax)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        out = hip.irrep_gemm(X, L.wpack, C, O, B, f16x2=L.wsplit2, x_absmax=amax)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f'B={B} X offset {off * 4:>9d} B: {dt*1e3:7.2f} ms   X@{X.data_ptr():#x} out@{out.data_ptr():#x}', flush=True)
