"""Read rate of a buffer that is re-read by back-to-back launches, over buffer sizes: below ~200 MB the re-reads come from the 256 MB
Infinity Cache (below 32 MB partly from the L2s), above they come from HBM (tools/hbm_rate.hip, read-only kernel).  The figure that
decides how many stacked Sinkhorn coupling matrices a group may hold (csrc/rm.hip)."""
import ctypes, os, subprocess
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = f'{HERE}/hbm_rate.so'
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(f'{HERE}/hbm_rate.hip'):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', f'{HERE}/hbm_rate.hip', '-o', so])
lib = ctypes.CDLL(so)
lib.hbm_rate_run.restype = ctypes.c_double
lib.hbm_rate_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
src = torch.empty((4 << 30) // 4, device='cuda').normal_(); dst = torch.empty(1 << 20, device='cuda')
torch.cuda.synchronize()
for mb in (8, 16, 25, 50, 100, 150, 200, 225, 256, 300, 400, 800, 4096):
    nbytes = mb << 20
    for blocks in (2048, 16384):
        ms = lib.hbm_rate_run(1, blocks, src.data_ptr(), dst.data_ptr(), nbytes, 50)
        print(f'read only {mb:5d} MB re-read x50, {blocks:6d} workgroups: {ms * 1e3:9.1f} us  {nbytes / ms / 1e9:6.2f} TB/s', flush=True)
