"""Time ft_nonlin (irrep <-> group domain transform + nonlinearity) at the GF layer shapes."""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65000
hip.ensure_fourier()
for C in (256, 512):
    X = torch.randn(hip.coef_size(C, B), device='cuda')
    bias = torch.randn(C, device='cuda'); bn = (torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda'))
    for sp in (False, True):
        for _ in range(2):
            hip.ft_nonlin(B, C, coef_in=X, bias=bias, bn=bn, split=sp)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            hip.ft_nonlin(B, C, coef_in=X, bias=bias, bn=bn, split=sp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        gb = 2 * 60 * C * B * 4 / 1e9
        print(f'ft_nonlin coef->coef C={C} B={B} split={sp}: {dt*1e3:.2f} ms  {gb/dt/1e3:.2f} TB/s  {2*2*64*64*C*B/dt/1e12:.1f} TFLOP/s (padded 64x64 transforms)')
