"""Power limit or stalls?  The ET trunk's 13-column stencil convolution (256 -> 512 channels, fp16 x 2: group_conv_split_kernel) timed with
random and with all-zero activations / weights (same instruction stream and traffic; cf. tools/gemm_power_probe.py).
Usage: python tools/et_conv_power_probe.py [rows]"""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip, synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
cfg = default_config(ET='yohoo')
for kind in ('random', 'zeros', 'random'):
    et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202)); et = et.cuda().eval()
    res = et.PartII_SO3_Conv_layers[0]
    if kind == 'zeros':
        with torch.no_grad():
            for q in res.comb_layer_in.parameters(): q.zero_()
    ga, gb, gc, p0, gmap = et._pruned_gathers()
    h = (torch.randn(B, 256, 48, device='cuda') if kind == 'random' else torch.zeros(B, 256, 48, device='cuda'))
    ah = h.abs().amax(dim=(1, 2)).contiguous()
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        res._b_in(h, gather=gb, in_rowmax=ah, want_rowmax=True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): res._b_in(h, gather=gb, in_rowmax=ah, want_rowmax=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * 512 * 13 * 256 * 13 * B
    print(f'operands={kind:7s} rows={B}: {ms:.3f} ms   {fl / ms / 1e9:.1f} TFLOP/s real   {3 * fl / ms / 1e9:.1f} TFLOP/s executed', flush=True)
