"""Evidence for DESIGN.md section 4b: how often a dot-product expansion of the matcher distance (|s|^2 + |t|^2 - 2 s.t, the MFMA form)
picks another nearest neighbour than the reference's literal formula sqrt(sum_f (s_f - t_f)^2 + 1e-7).  CPU/numpy, ~1 minute."""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import ref_numpy as O


def bf16(x):
    u = x.view(np.uint32); r = ((u >> 16) & 1) + 0x7fff
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def expansion(S, T, rnd):
    G = rnd(S) @ rnd(T).T
    return ((S * S).sum(1, dtype=np.float32)[:, None] + (T * T).sum(1, dtype=np.float32)[None, :] - np.float32(2) * G).argmin(1)


rng = np.random.default_rng(0)
A = rng.standard_normal((5000, 32)).astype(np.float32); A /= np.linalg.norm(A, axis=1, keepdims=True)
B = (A[rng.permutation(5000)] + 0.05 * rng.standard_normal((5000, 32))).astype(np.float32)
B /= np.linalg.norm(B, axis=1, keepdims=True).astype(np.float32)
C = (A[rng.integers(0, 50, 5000)] + 0.01 * rng.standard_normal((5000, 32))).astype(np.float32)      # clustered targets: near ties
for name, T in (('well-separated targets', A), ('clustered targets', C)):
    _, exact = O.knn(T, B, 1)
    print(f'{name}: f32 expansion differs on {int((expansion(B, T, lambda x: x) != exact).sum())} / 5000 rows, '
          f'bf16-input expansion on {int((expansion(B, T, bf16) != exact).sum())} / 5000')
