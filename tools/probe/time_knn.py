import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
k = torch.from_numpy(np.random.default_rng(0).uniform(0, 3, (5000, 3)).astype(np.float32)).cuda()
for _ in range(3): hip.knn_search(k, k, 5)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): idx = hip.knn_search(k, k, 5)
torch.cuda.synchronize(); print(f'knn_search k=5 5000x5000x3: {(time.perf_counter() - t) * 100:.3f} ms')
t = time.perf_counter()
for _ in range(10): a = idx.cpu().numpy()
print(f'copy back: {(time.perf_counter() - t) * 100:.3f} ms')
