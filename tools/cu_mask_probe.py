"""Does the power-limited GEMM lose throughput when it runs on fewer CUs?  And what does ft_nonlin reach on the remaining ones, alone and
concurrently?  (hipExtStreamCreateWithCUMask; the mask has one bit per CU, 8 x 32.)  Usage: python tools/cu_mask_probe.py"""
import sys, ctypes, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
rt = ctypes.CDLL('libamdhip64.so')
rt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
rt.hipExtStreamCreateWithCUMask.restype = ctypes.c_int

def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

torch.manual_seed(0)
hip.ensure_fourier()
B, C, O = 61440, 256, 512
L = _Layer(torch.nn.Conv2d(C, O, (1, 13)))
X = torch.randn(hip.coef_size(C, B), device='cuda')
Xp, xb = hip.pack_coefs_f16x2(X, C, B)
T = torch.randn(hip.coef_size(512, B), device='cuda')
bias = torch.randn(512, device='cuda'); bn = (torch.rand(512, device='cuda') + 0.5, torch.randn(512, device='cuda'))
ob = torch.full((hip.coef_pitch(B),), 300.0, device='cuda')
gemm = lambda: hip.irrep_gemm(Xp, L.wpack, C, O, B, f16x2=L.wsplit2, x_bound=xb)
ft = lambda: hip.ft_nonlin(B, 512, coef_in=T, bias=bias, bn=bn, split='f16x2', out_bound=ob)

def timed(fn, stream, n=8):
    with torch.cuda.stream(stream):
        for _ in range(2): fn()
        stream.synchronize()
        t = time.perf_counter()
        for _ in range(n): fn()
        stream.synchronize()
    return (time.perf_counter() - t) / n * 1e3

full = [0xffffffff] * 8
for name, mg, mf in (('256 / 256', full, full), ('224 + 32', [0x0fffffff] * 8, [0xf0000000] * 8), ('192 + 64', [0x00ffffff] * 8, [0xff000000] * 8),
                     ('160 + 96', [0x000fffff] * 8, [0xfff00000] * 8)):
    sg, sf = masked_stream(mg), masked_stream(mf)
    tg, tf = timed(gemm, sg), timed(ft, sf)
    # concurrently: n GEMMs on one stream, ft_nonlin launches on the other until the GEMMs are done
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(sg):
        for _ in range(8): gemm()
    nft = 0
    with torch.cuda.stream(sf):
        while not sg.query():
            ft(); nft += 1
            if nft % 4 == 0: sf.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) * 1e3
    print(f'CUs GEMM + transform {name}: GEMM alone {tg:.2f} ms, ft_nonlin alone {tf:.2f} ms;  together: 8 GEMMs + {nft} transforms in {dt:.1f} ms '
          f'(serial at full width would be {8 * 10.6 + nft * 3.45:.1f} ms)', flush=True)
