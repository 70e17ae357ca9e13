"""Does a coefficient tensor written by one kernel and read by the next come from the 256 MB Infinity Cache when the working set is small?
ft_nonlin (coefficient -> coefficient, 512 channels) on B keypoints, (a) back to back on the same buffers (input last written by the
previous call's neighbour), (b) with 2 GB of unrelated traffic in between.  Usage: python tools/probe/mall_probe.py"""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
hip.ensure_fourier()
C = 512
flush_src = torch.empty(512 << 20, device='cuda', dtype=torch.float32).normal_()     # 2 GB
flush_dst = torch.empty_like(flush_src)
for B in (512, 1024, 2048, 4096, 65536):
    X = torch.randn(hip.coef_size(C, B), device='cuda')
    bias = torch.randn(C, device='cuda'); bn = (torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda'))
    ob = torch.full((hip.coef_pitch(B),), 300.0, device='cuda')
    Y = hip.ft_nonlin(B, C, coef_in=X, bias=bias, bn=bn, split='f16x2', out_bound=ob)
    res = {}
    for mode in ('hot', 'flushed'):
        ts = []
        for it in range(12):
            X.mul_(1.0)                                     # a producer kernel writes the input (stands for the GEMM's epilogue)
            if mode == 'flushed':
                flush_dst.copy_(flush_src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.ft_nonlin(B, C, coef_in=X, bias=bias, bn=bn, split='f16x2', out_bound=ob)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = sorted(ts)[len(ts) // 2]
    gb = 2 * 60 * C * B * 4 / 1e9
    print(f'B={B:6d} ({gb * 1e3:7.1f} MB in+out): hot {res["hot"] * 1e3:8.1f} us = {gb / res["hot"]:.2f} TB/s   flushed {res["flushed"] * 1e3:8.1f} us = {gb / res["flushed"]:.2f} TB/s')
