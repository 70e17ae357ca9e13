import sys
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import hip
from roreg_amd.network.gf_fourier import _Layer
rng = np.random.default_rng(3)
B, C, O = 256, 256, 512
conv = torch.nn.Conv2d(C, O, (1, 13)); L = _Layer(conv)
for outlier in (1.0, 1e2, 1e4, 1e6):
    X = torch.from_numpy(rng.standard_normal(hip.coef_size(C, B)).astype(np.float32)).cuda()
    Bp = hip.coef_pitch(B)
    v = hip.coef_views(X, C, B)
    v[4][:, 7] *= outlier                       # one column (one keypoint / i) of irrep 4 is huge
    amax = X.abs().max().reshape(1)
    T16 = hip.irrep_gemm(X, L.wpack, C, O, B, f16x2=L.wsplit2, x_absmax=amax)
    T32 = hip.irrep_gemm(X, L.wpack, C, O, B)
    xv = v[4].double().cpu().numpy(); ref = L.dense[4].astype(np.float64) @ xv
    g16 = hip.coef_views(T16, O, B)[4].double().cpu().numpy(); g32 = hip.coef_views(T32, O, B)[4].double().cpu().numpy()
    cols = [c for c in range(ref.shape[1]) if c != 7][:500]
    colscale = np.abs(ref[:, cols]).max(0)
    e16 = (np.abs(g16[:ref.shape[0]][:, cols] - ref[:, cols]).max(0) / colscale).max()
    e32 = (np.abs(g32[:ref.shape[0]][:, cols] - ref[:, cols]).max(0) / colscale).max()
    print(f'outlier x{outlier:g}: worst per-column relative error on the NORMAL columns: fp16x2 {e16:.2e}   f32-input {e32:.2e}')
