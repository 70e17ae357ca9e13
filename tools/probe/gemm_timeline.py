"""Timeline of a workgroup's life in the big GEMM launch (debug stamps, wall clock 100 MHz): per tile the phases
prologue / loop / epilogue / tail, and per CU the gap between one workgroup's end and the next one's start.
NEEDS AN INSTRUMENTED BUILD (not in the tree): GemmSplitDescs gets `unsigned long long *trace` (set through a debug entry
roreg_gemm_trace_dbg(void *)), and thread 0 of a workgroup writes wall_clock64() into trace[slot * 8 + i] -- slot = blockIdx.x (per-tile launch)
or blockIdx.x * 160 + tile (persistent launch); i = 0 kernel / tile start, 1 loop start, 2 loop end, 3 epilogue end, 4 after s_waitcnt vmcnt(0)
(per-tile launch), 5 / 6 begin / end of the epilogue passes (persistent), 7 = nss << 48 | HW_REG_XCC_ID << 32 | HW_REG_HW_ID.  The output of the
round-5 run is profiles/r05_gemm_tile_timeline.txt."""
import os, sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
torch.manual_seed(0)
C, O, B = (256, 512, 61440)
resid = bool(int(os.environ.get('RESID', '0')))
if resid: C, O = 512, 256
conv = torch.nn.Conv2d(C, O, (1, 13))
x = torch.randn(hip.coef_size(C, B), device='cuda')
if os.environ.get('ROREG_AB_ZEROS'):
    with torch.no_grad(): conv.weight.zero_(); conv.bias.zero_()
    x.zero_()
L = _Layer(conv)
Xp, xb = hip.pack_coefs_f16x2(x, C, B); Xp = hip.words_to_planes(Xp, C, B)
add = torch.randn(hip.coef_size(O, B), device='cuda') if resid else None
nb = (torch.rand(O, device='cuda') + 0.5, torch.rand(O, device='cuda'))
f = lambda: hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, x_planes=2, add=add, next_bound=nb)
lib = hip.lib()
lib.roreg_gemm_trace_dbg.argtypes = [ctypes.c_void_p]; lib.roreg_gemm_trace_dbg.restype = None
for persist in (False, True):
    with hip.gemm_persistent(persist):
        for _ in range(3): f()
        torch.cuda.synchronize()
        tr = torch.zeros(256 * 160 * 8, dtype=torch.int64, device='cuda')
        lib.roreg_gemm_trace_dbg(tr.data_ptr())
        f(); torch.cuda.synchronize()
        lib.roreg_gemm_trace_dbg(None)
    t = tr.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    nss = (t[:, 7] >> 48) & 0xffff
    xcc = (t[:, 7] >> 32) & 7
    hw = t[:, 7] & 0xffff
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
    us = lambda a: a / 100.0
    print(f'=== persistent={persist}: {len(t)} tiles, launch span {us(max(t[:, 3].max(), t[:, 4].max()) - t0):.0f} us')
    for n in sorted(set(nss)):
        s = t[nss == n]
        pro = us(s[:, 1] - s[:, 0]); loop = us(s[:, 2] - s[:, 1]); epi = us(s[:, 3] - s[:, 2])
        line = f'  K32 steps {n:2d}: {len(s):5d} tiles  prologue {np.median(pro):6.2f}  loop {np.median(loop):6.2f} ({np.median(loop) / n:5.3f}/step)  epilogue {np.median(epi):6.2f}'
        if persist:
            line += f'  [begin {np.median(us(s[:, 5] - s[:, 2])):5.2f}  passes {np.median(us(s[:, 6] - s[:, 5])):5.2f}  end+barrier {np.median(us(s[:, 3] - s[:, 6])):5.2f}]'
        else:
            line += f'  store drain {np.median(us(s[:, 4] - s[:, 3])):5.2f}'
        print(line)
    # per CU: gap between consecutive tiles
    gaps = []
    for c in set(cu):
        s = t[cu == c]; s = s[np.argsort(s[:, 0])]
        end = s[:, 3] if persist else s[:, 4]
        gaps.extend(us(s[1:, 0] - end[:-1]))
    gaps = np.array(gaps)
    print(f'  {len(set(cu))} CUs; gap between a tile\'s end and the next start on the same CU: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f}, p90 {np.percentile(gaps, 90):.2f}')
    tot = us((t[:, 3] if persist else t[:, 4]) - t[:, 0])
    print(f'  sum of tile lives / CUs = {tot.sum() / len(set(cu)):.0f} us; sum of loops / CUs = {us(t[:, 2] - t[:, 1]).sum() / len(set(cu)):.0f} us')
