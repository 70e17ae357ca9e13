"""Time the ET network's dense tail (hip.dense_split: 6656 -> 256 with BatchNorm + ReLU on the input and a residual, then the 1x1 head
256 -> 512 -> 128 -> 4) on B rows.  Usage: python tools/time_dense.py [rows]   (OLD=1: load roreg_amd/libroreg_hip_old.so, for A/B runs)"""
import sys, os
sys.path.insert(0, '.')
import torch
from roreg_amd import hip, synth
if os.environ.get('OLD'): hip._LIB_PATH = hip._LIB_PATH.replace('libroreg_hip.so', 'libroreg_hip_old.so')
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
cfg = default_config(ET='yohoo')
torch.manual_seed(0)
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202)); et = et.cuda().eval()
d_out, d0, d1, d2 = et._dense_plans()
m = torch.randn(B, 6656, device='cuda'); sc = torch.randn(B, 256, device='cuda')
am = m.abs().amax(dim=1).contiguous()
def run():
    t, at = hip.dense_split(m, d_out, residual=sc, in_rowmax=am, want_rowmax=True)
    z, az = hip.dense_split(t, d0, in_rowmax=at, want_rowmax=True)
    z, az = hip.dense_split(z, d1, in_rowmax=az, want_rowmax=True)
    return hip.dense_split(z, d2, in_rowmax=az)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): out = run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 2.0 * B * (6656 * 256 + 256 * 512 + 512 * 128 + 128 * 4)
import hashlib
print(f'dense tail on {B} rows: {ms:.3f} ms   {fl / ms / 1e9:.1f} TFLOP/s real, {3 * fl / ms / 1e9:.1f} executed (fp16 x 2)   checksum {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}')
