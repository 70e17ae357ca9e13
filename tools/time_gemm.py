"""Time and check the irrep-domain GEMM kernels (f32-input MFMA, bf16 x 3 split, fp16 x 2 with block scaling) on GF's big layers."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
for (C, O) in [(256, 512), (512, 256)]:
    conv = torch.nn.Conv2d(C, O, (1, 13))
    L = _Layer(conv)
    X = torch.randn(hip.coef_size(C, B), device='cuda') * torch.exp(torch.randn(hip.coef_size(C, B), device='cuda'))
    Xp, xb = hip.pack_coefs_f16x2(X, C, B)
    bn = (torch.rand(O, device='cuda') + 0.5, torch.randn(O, device='cuda') * 0.1)
    nb = hip.next_bound(bn, L.bias)
    variants = [('f32', X, dict()), ('bf16x3', X, dict(split=L.wsplit)), ('fp16x2', Xp, dict(f16x2=L.wsplit2, x_bound=xb)),
                ('fp16x2+bound', Xp, dict(f16x2=L.wsplit2, x_bound=xb, next_bound=nb))]
    outs = {}
    for name, Xin, kw in variants:
        for _ in range(2):
            hip.irrep_gemm(Xin, L.wpack, C, O, B, **kw)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            r = hip.irrep_gemm(Xin, L.wpack, C, O, B, **kw)
        outs[name] = r[0] if isinstance(r, tuple) else r
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(f'{C}->{O} B={B} {name}: {dt*1e3:.2f} ms  {2.0*O*C*B*244/dt/1e12:.1f} TFLOP/s (f32-equivalent)')
    # accuracy on a slice against float64 (irrep 4, d = 5)
    Bp = hip.coef_pitch(B)
    xv = hip.coef_views(X, C, B)[4][:, :4096].double().cpu().numpy()
    Wm = L.dense[4].astype(np.float64)
    ref = Wm @ xv
    s = np.abs(ref).max()
    for name in outs:
        got = hip.coef_views(outs[name], O, B)[4][:, :4096].double().cpu().numpy()
        print(f'   {name}: max err / max|ref| = {np.abs(got - ref[:got.shape[0]]).max() / s:.3e}   rms err / rms ref = {np.sqrt(np.mean((got - ref[:got.shape[0]])**2)) / np.sqrt(np.mean(ref**2)):.3e}')
