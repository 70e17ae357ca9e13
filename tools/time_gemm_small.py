"""GF's two thin irrep GEMMs (32 -> 256: write-bound, K = 32 d; 256 -> 32: read-bound, M = 32 d) at B keypoints, fp16 x 2.
Usage: python tools/time_gemm_small.py [B]      (ROREG_TILE_M128=1: 128-row tiles / 4-wave workgroups for every layer)"""
import os, sys
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(0)
for (C, O) in [(32, 256), (256, 32)]:
    L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
    X = torch.randn(hip.coef_size(C, B), device='cuda')
    Xp, xb = hip.pack_coefs_f16x2(X, C, B)
    planes = hip.XDMA and O % 256 == 0 and not os.environ.get('ROREG_TILE_M128')      # the LDS-DMA kernel (256-row tiles only)
    if planes:
        Xp = hip.words_to_planes(Xp, C, B)
    for _ in range(3): hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, x_planes=planes)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, x_planes=planes)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = 60 * B * 4 * (C + O) / 1e9
    print(f'{C}->{O} B={B}{" (LDS-DMA kernel)" if planes else ""}: {ms:.3f} ms   {gb / ms:.2f} TB/s of X + Out   {2.0 * 244 * C * O * B / ms / 1e9:.1f} TFLOP/s real')
