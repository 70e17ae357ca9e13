"""Achievable HBM rates of plain streaming kernels (tools/hbm_rate.hip): copy, read-only, write-only, copy with nontemporal hints, over
workgroup counts.  The reference point for ft_nonlin and the other streaming kernels (profiles/)."""
import ctypes, os, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = f'{HERE}/hbm_rate.so'
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(f'{HERE}/hbm_rate.hip'):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', f'{HERE}/hbm_rate.hip', '-o', so])
lib = ctypes.CDLL(so)
lib.hbm_rate_run.restype = ctypes.c_double
lib.hbm_rate_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
nbytes = 4 << 30
src = torch.empty(nbytes // 4, device='cuda').normal_(); dst = torch.empty_like(src)
torch.cuda.synchronize()
for mode, name, factor in ((0, 'copy', 2), (3, 'copy, nontemporal', 2), (1, 'read only', 1), (2, 'write only', 1)):
    for blocks in (1024, 2048, 4096, 16384, 65536):
        ms = lib.hbm_rate_run(mode, blocks, src.data_ptr(), dst.data_ptr(), nbytes, 10)
        print(f'{name:18s} {blocks:6d} workgroups: {ms:7.3f} ms  {factor * nbytes / ms / 1e9:6.2f} TB/s', flush=True)
