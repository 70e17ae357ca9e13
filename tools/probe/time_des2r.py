"""Des2R kernel time (80000 correspondences): the literal kernel vs the irrep-domain bound + exact re-check, on noise-level correspondences
(the hard case for the bound: every correlation is of the same size) and on well-matched ones; plus R_indicator (group_corr)."""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
hip.ensure_tables()
M = 80000


def timeit(fn, n=10):
    for _ in range(2):
        r = fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n, r


f1 = torch.randn((M, 32, 60), device='cuda'); f0 = torch.randn((M, 32, 60), device='cuda')
for name, a, b in (('noise', f1, f0), ('matched', f0 + 0.3 * f1, f0)):
    dt, want = timeit(lambda: hip.des2r(a, b))
    print(f'[{name}] literal des2r {M} items: {dt*1e3:.3f} ms  ({2 * M * 7680 / dt / 1e12:.2f} TB/s)')
    ca, cb = hip.feat_coefs(a), hip.feat_coefs(b)
    dtc, _ = timeit(lambda: hip.feat_coefs(a))
    hip.des2r_recheck_count()
    dt2, got = timeit(lambda: hip.des2r(a, b, coefs1=ca, coefs0=cb))
    n = hip.des2r_recheck_count()
    print(f'[{name}] irrep  des2r {M} items: {dt2*1e3:.3f} ms  ({2 * M * 7680 / dt2 / 1e12:.2f} TB/s)  speed-up {dt/dt2:.2f}x  identical={bool(torch.equal(got, want))}  '
          f'exact-path share {n / 12.0 / M:.4f}   (feat_coefs of {M} keypoints: {dtc*1e3:.3f} ms, once per cloud)')
dt, cor = timeit(lambda: hip.group_corr(f1, f0, transpose=True))
print(f'group_corr(transpose) {M} items: {dt*1e3:.3f} ms  checksum {float(cor.double().sum()):.6e}')
