"""Des2R / R_indicator kernel time (80000 correspondences)."""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
hip.ensure_tables()
M = 80000
f1 = torch.randn((M, 32, 60), device='cuda'); f0 = torch.randn((M, 32, 60), device='cuda')
for _ in range(2): idx = hip.des2r(f1, f0)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): idx = hip.des2r(f1, f0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f'des2r {M} items: {dt*1e3:.3f} ms  ({M * 115200 * 2 / dt / 1e12:.1f} TFLOP/s, {2 * M * 7680 / dt / 1e12:.2f} TB/s)  checksum {int(idx.sum())}')
cor = hip.group_corr(f1, f0, transpose=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): cor = hip.group_corr(f1, f0, transpose=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f'group_corr(transpose) {M} items: {dt*1e3:.3f} ms  checksum {float(cor.double().sum()):.6e}')
