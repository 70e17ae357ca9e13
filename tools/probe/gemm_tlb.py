"""Epilogue share of the big GEMM launch by batch size (row pitch of the [M][N] coefficient matrices = d * B * 4 bytes): is it address translation?
python tools/probe/gemm_tlb.py  (run twice: plain and ROREG_GEMM_DBG=1 = no epilogue)"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
torch.manual_seed(0)
C, O = 256, 512
L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
for B in [int(b) for b in os.environ.get('BS', '61440,61472,61504,61568,61696,61952,62464,65000,65024,65056,60000,60032').split(',')]:
    x = torch.randn(hip.coef_size(C, B), device='cuda')
    if os.environ.get('ROREG_AB_ZEROS'):
        x.zero_()
    Xp, xb = hip.pack_coefs_f16x2(x, C, B)
    Xp = hip.words_to_planes(Xp, C, B)
    del x
    f = lambda: hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, x_planes=2)
    for _ in range(5): f()
    torch.cuda.synchronize()
    n = max(5, int(200000 / B))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'B={B:7d}: {ms:8.3f} ms  {ms / B * 1e6:7.1f} ns/keypoint', flush=True)
    del Xp
