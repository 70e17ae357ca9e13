"""Sustained fp16 MFMA rate from registers (tools/mfma_peak.hip), for the occupancy / duty settings the GEMM runs at."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
if not os.path.exists(f'{HERE}/mfma_peak.so'):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', f'{HERE}/mfma_peak.hip', '-o', f'{HERE}/mfma_peak.so'])
lib = ctypes.CDLL(f'{HERE}/mfma_peak.so')
lib.mfma_peak_run.restype = ctypes.c_double
lib.mfma_peak_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.empty(1 << 22, device='cuda')
for blocks, threads, label in ((256, 256, '1 wave/SIMD'), (512, 256, '2 waves/SIMD'), (1024, 256, '4 waves/SIMD'), (2048, 256, '8 waves/SIMD (two rounds)')):
    for sleep, rnd in ((0, 0), (0, 1), (1, 1)):
        tf = lib.mfma_peak_run(blocks, threads, 20000, sleep, rnd, ctypes.c_void_p(out.data_ptr()))
        clk = tf * 1e12 / (256 * 4 * 1024.0) / 1e9
        print(f'{label:28s} operands={"random" if rnd else "smooth"} sleep={sleep}: {tf:7.1f} TFLOP/s  (= {clk:.2f} GHz x 100 % duty of the 1024 flop/clk/SIMD pipes)')
