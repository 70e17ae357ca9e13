"""Time the irrep-domain GEMM kernels (exact f32 vs 3 x bf16 split) on GF's big layers."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
for (C, O) in [(256, 512), (512, 256)]:
    conv = torch.nn.Conv2d(C, O, (1, 13))
    L = _Layer(conv)
    X = torch.randn(hip.coef_size(C, B), device='cuda')
    for name, sp in [('f32', None), ('split', L.wsplit)]:
        for _ in range(2):
            hip.irrep_gemm(X, L.wpack, C, O, B, split=sp)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            hip.irrep_gemm(X, L.wpack, C, O, B, split=sp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(f'{C}->{O} B={B} {name}: {dt*1e3:.2f} ms  {2.0*O*C*B*244/dt/1e12:.1f} TFLOP/s (f32-equivalent)')
