"""GF's first layer (32 -> 256) and last layer (256 -> 32) : the register-staged word-layout kernel vs the LDS-DMA 16x16x32 kernel (planes), ms per launch."""
import sys
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
for C, O in ((32, 256),):
    L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
    x = torch.randn(hip.coef_size(C, B), device='cuda')
    Xw, xb = hip.pack_coefs_f16x2(x, C, B)
    Xp = hip.words_to_planes(Xw, C, B)
    nb = (torch.rand(O, device='cuda') + 0.5, torch.rand(O, device='cuda'))
    fw = lambda: hip.irrep_gemm(Xw, None, C, O, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb)
    fp = lambda: hip.irrep_gemm(Xp, None, C, O, B, f16x2=L.wsplit2, x_bound=xb, next_bound=nb, x_planes=2)
    a, b = fw()[0], fp()[0]
    print('max diff / scale', float((a - b).abs().max() / a.abs().max()))
    for name, f in (('words (split kernel)', fw), ('planes (xdma16)', fp), ('words (split kernel)', fw), ('planes (xdma16)', fp)):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f'C={C} O={O} B={B} {name}: {e0.elapsed_time(e1) / 20:.3f} ms', flush=True)
