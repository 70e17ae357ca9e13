"""Practical HBM rate of this GPU for a streaming read + write of the size of one ft_nonlin pass (torch device copy, ~4 GB each way)."""
import torch, time
n = 60 * 256 * 65024
x = torch.empty(n, device='cuda', dtype=torch.float32).normal_()
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f'copy {n * 4 / 1e9:.2f} GB: {dt * 1e3:.3f} ms  {2 * n * 4 / dt / 1e12:.2f} TB/s (read + write)')
t = time.perf_counter()
for _ in range(10): s = x.sum()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f'read-only reduction: {dt * 1e3:.3f} ms  {n * 4 / dt / 1e12:.2f} TB/s')
t = time.perf_counter()
for _ in range(10): y.fill_(1.5)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f'write-only fill: {dt * 1e3:.3f} ms  {n * 4 / dt / 1e12:.2f} TB/s')
t = time.perf_counter()
for _ in range(10): torch.add(x, 1.0, out=y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f'read + write elementwise add: {dt * 1e3:.3f} ms  {2 * n * 4 / dt / 1e12:.2f} TB/s')
