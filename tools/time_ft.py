"""Time ft_nonlin (irrep <-> group domain transform + nonlinearity) at the GF layer shapes."""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65000
hip.ensure_fourier()
for C in (256, 512):
    X = torch.randn(hip.coef_size(C, B), device='cuda')
    bias = torch.randn(C, device='cuda'); bn = (torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda'))
    ob = torch.full((hip.coef_pitch(B),), 300.0, device='cuda')        # per-keypoint bound of the fp16 x 2 output split
    for sp in (False, True, 'f16x2'):
        kw = dict(out_bound=ob) if sp == 'f16x2' else {}
        for _ in range(2):
            hip.ft_nonlin(B, C, coef_in=X, bias=bias, bn=bn, split=sp, **kw)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            hip.ft_nonlin(B, C, coef_in=X, bias=bias, bn=bn, split=sp, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        gb = 2 * 60 * C * B * 4 / 1e9
        print(f'ft_nonlin coef->coef C={C} B={B} split={sp}: {dt*1e3:.2f} ms  {gb/dt/1e3:.2f} TB/s  {2*2*64*64*C*B/dt/1e12:.1f} TFLOP/s (padded 64x64 transforms)')
# ET Conv_init output: irrep -> 45 live group columns (pitch 48), and the group -> irrep input transform
B2 = 60000
X = torch.randn(hip.coef_size(256, B2), device='cuda'); bias = torch.randn(256, device='cuda')
gmap = torch.full((60,), -1, dtype=torch.int32); gmap[:45] = torch.arange(45, dtype=torch.int32); gmap = gmap.cuda()
xs = torch.randn(B2, 128, 60, device='cuda'); bn = (torch.rand(128, device='cuda') + 0.5, torch.randn(128, device='cuda'))
ob2 = torch.full((hip.coef_pitch(B2),), 300.0, device='cuda')
for sp in (False, True, 'f16x2'):
    for name, fn, gb in (('irrep->group(45 of 60) C=256', lambda: hip.ft_nonlin(B2, 256, coef_in=X, bias=bias, spatial_out=True, g_map=gmap, Lout=48, Lvalid=45, split=sp), (60 + 48) * 256 * B2 * 4 / 1e9),
                         ('group->irrep C=128', lambda: hip.ft_nonlin(B2, 128, x_spatial=xs, bn=bn, split=sp, **(dict(out_bound=ob2) if sp == 'f16x2' else {})), 120 * 128 * B2 * 4 / 1e9)):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f'ft_nonlin {name} B={B2} split={sp}: {dt*1e3:.2f} ms  {gb/dt/1e3:.2f} TB/s')
# GF's last transform: irrep -> group with the input residual (32 channels, all 60 columns)
B3 = 65536
X = torch.randn(hip.coef_size(32, B3), device='cuda'); bias = torch.randn(32, device='cuda')
xr = torch.randn(B3, 32, 60, device='cuda')
for sp in ('f16x2',):
    fn = lambda: hip.ft_nonlin(B3, 32, coef_in=X, bias=bias, resid_spatial=xr, spatial_out=True, split=sp)
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f'ft_nonlin irrep->group + residual C=32 B={B3} split={sp}: {dt*1e3:.2f} ms  {3 * 60 * 32 * B3 * 4 / dt / 1e12:.2f} TB/s')
