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
    for sleep, rnd in ((0, 2), (0, 0), (0, 1), (1, 1)):
        iters = 20000
        tf = lib.mfma_peak_run(blocks, threads, iters, sleep, rnd, ctypes.c_void_p(out.data_ptr()))
        torch.cuda.synchronize()
        cyc = float(out[(1 << 21):(1 << 21) + 256].mean())                    # s_memtime ticks of one wave's loop
        per = cyc / (iters * 32.0)                                                # ticks per MFMA of that wave (32 = back to back on a free pipe)
        print(f'{label:28s} operands={("smooth", "random", "zeros")[rnd]:6s} sleep={sleep}: {tf:7.1f} TFLOP/s   {per:6.1f} s_memtime ticks per MFMA of one wave')

# the 16x16x32 shape (same flops per instruction, half the accumulator traffic per flop), 8 accumulators per wave
lib.mfma_peak_run16.restype = ctypes.c_double
lib.mfma_peak_run16.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for blocks, threads, label in ((512, 256, '2 waves/SIMD'), (1024, 256, '4 waves/SIMD')):
    for rnd in (2, 0, 1):
        tf = lib.mfma_peak_run16(blocks, threads, 20000, 0, rnd, ctypes.c_void_p(out.data_ptr()))
        torch.cuda.synchronize()
        print(f'16x16x32  {label:18s} operands={("smooth", "random", "zeros")[rnd]:6s}: {tf:7.1f} TFLOP/s')
