"""Probe for the occasional slow 256->512 launch seen in tools/time_gemm.py: fresh multi-GB allocations per call (outputs retained),
operand addresses printed next to the time."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
C, O = 256, 512
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64256
L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
n = hip.coef_size(C, B)
X = torch.randn(n, device='cuda') * torch.exp(torch.randn(n, device='cuda'))
amax = X.abs().max().reshape(1).float()
keep = []
for trial in range(6):
    outs = []
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        outs.append(hip.irrep_gemm(X, L.wpack, C, O, B, f16x2=L.wsplit2, x_absmax=amax))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f'trial {trial}: {dt*1e3:7.2f} ms  X@{X.data_ptr():#x} outs@{[hex(o.data_ptr()) for o in outs]}  reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB', flush=True)
    if trial % 2 == 0:
        keep.append(outs[0])          # retain some outputs so that later calls need fresh segments
