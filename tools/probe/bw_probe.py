import torch, time
x = torch.randn(2_000_000_000, device='cuda'); y = torch.empty_like(x)
for op, name in ((lambda: torch.relu(x, out=y) if False else y.copy_(x), 'copy 8GB->8GB'), (lambda: torch.mul(x, 2.0, out=y), 'mul out=')):
    for _ in range(2): op()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): op()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(name, f'{dt*1e3:.2f} ms', f'{16/dt/1e3:.2f} TB/s (read+write)')
s = x.sum(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): s = x.sum()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print('sum (read only)', f'{dt*1e3:.2f} ms', f'{8/dt/1e3:.2f} TB/s')
