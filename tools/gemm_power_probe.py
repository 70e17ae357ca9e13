"""Is the fp16 x 2 irrep GEMM held back by the power limit or by stalls?  The same launch (same instruction stream, same memory traffic)
is timed with random operands, with constant operands and with all-zero operands: toggling-dependent power is the only thing that
differs, so a faster run with quiet operands is the package power limit at work (clocks), an equal time is stalls.
Usage: python tools/gemm_power_probe.py [B]"""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
C, O = 256, 512
torch.manual_seed(0)


def layer(kind):
    conv = torch.nn.Conv2d(C, O, (1, 13))
    with torch.no_grad():
        if kind == 'zeros':
            conv.weight.zero_(); conv.bias.zero_()
        elif kind == 'ones':
            conv.weight.fill_(0.01); conv.bias.zero_()
    return _Layer(conv)


def coefs(kind):
    n = hip.coef_size(C, B)
    if kind == 'random':
        return torch.randn(n, device='cuda') * torch.exp(torch.randn(n, device='cuda'))
    return torch.zeros(n, device='cuda') if kind == 'zeros' else torch.full((n,), 0.5, device='cuda')


for kind in ('random', 'ones', 'zeros', 'random'):
    L = layer(kind)
    Xp, xb = hip.pack_coefs_f16x2(coefs(kind), C, B)
    if hip.XDMA:
        Xp = hip.words_to_planes(Xp, C, B)
    t_end = time.perf_counter() + 1.5                       # let the power management settle on this operand class
    while time.perf_counter() < t_end:
        hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, x_planes=hip.XDMA)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, x_planes=hip.XDMA)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'operands={kind:7s} B={B}: {ms:.3f} ms / launch   {2.0 * 244 * C * O * B / ms / 1e9:.1f} TFLOP/s real   {3 * 2.0 * 244 * C * O * B / ms / 1e9:.1f} TFLOP/s executed', flush=True)
