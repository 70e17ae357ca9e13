"""The 16x16x32 LDS-DMA irrep GEMM in its three launch forms (hip.gemm_persistent: 0 one workgroup per tile "t", 1 persistent "P", 2 half tiles with two
workgroups per CU "H"; FORMS=0,2 selects): results bitwise
equal?  ms per launch, alternating, on random operands -- the big layers of the extractor (256 -> 512 plain; 512 -> 256 with the residual
and the bound of the next transform), a 61440-keypoint batch and two smaller ones (ragged last tiles).
Usage: python tools/gemm_persist_ab.py [reps]"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FORMS = [int(f) for f in os.environ.get('FORMS', '0,1,2').split(',')]
torch.manual_seed(0)


def run(C, O, B, resid, bound, n=20):
    conv = torch.nn.Conv2d(C, O, (1, 13))
    x = torch.randn(hip.coef_size(C, B), device='cuda') * torch.exp(torch.randn(hip.coef_size(C, B), device='cuda'))
    if os.environ.get('ROREG_AB_ZEROS'):                   # quiet operands: the same instruction stream and traffic without the toggling power
        with torch.no_grad():
            conv.weight.zero_(); conv.bias.zero_()
        x.zero_()
    L = _Layer(conv)
    Xp, xb = hip.pack_coefs_f16x2(x, C, B)
    Xp = hip.words_to_planes(Xp, C, B)
    add = torch.randn(hip.coef_size(O, B), device='cuda') if resid else None
    nb = (torch.rand(O, device='cuda') + 0.5, torch.rand(O, device='cuda')) if bound else None

    def once():
        return hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb, x_planes=2, add=add, next_bound=nb)

    outs = {}
    for on in FORMS:
        with hip.gemm_persistent(on):
            outs[on] = once()
    torch.cuda.synchronize()
    a = outs[FORMS[0]]
    same = all(((torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])) if bound else torch.equal(a, b)) for b in outs.values())
    line = f'C={C} O={O} B={B} resid={int(resid)} bound={int(bound)}: bitwise {same}'
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        once(); torch.cuda.synchronize()
    for rep in range(reps):
        for on in FORMS:
            with hip.gemm_persistent(on):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                once(); e0.record()
                for _ in range(n):
                    once()
                e1.record(); torch.cuda.synchronize()
                line += f'  {"tPH"[on]} {e0.elapsed_time(e1) / n:.3f}'
    print(line + ' ms', flush=True)
    return same


ok = True
if os.environ.get('SHAPES'):                                   # SHAPES="C,O,B,resid,bound;..."
    for sh in os.environ['SHAPES'].split(';'):
        C, O, B, resid, bound = [int(v) for v in sh.split(',')]
        ok = run(C, O, B, bool(resid), bool(bound)) and ok
    print('ALL BITWISE EQUAL' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)
for (C, O, B, resid, bound) in ((256, 512, 61440, False, True), (512, 256, 61440, True, True), (256, 512, 61440, False, False), (512, 256, 61440, True, False),
                                (256, 512, 5000, False, True), (512, 256, 4967, True, True), (256, 256, 14464, True, True), (32, 256, 1000, False, True)):
    ok = run(C, O, B, resid, bound, n=20 if B > 20000 else 50) and ok
print('ALL BITWISE EQUAL' if ok else 'MISMATCH')
sys.exit(0 if ok else 1)
