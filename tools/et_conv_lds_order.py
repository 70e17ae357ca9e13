"""The ET trunk's 13-column stencil convolution (256 -> 512 channels, fp16 x 2: group_conv_split_kernel) with the natural LDS slot order
(round 2: stride 48, 65 % of the LDS cycles bank conflicts) against the searched order (tools/lds_perm_search.py): time per launch and
bitwise equality of the outputs.  Usage: python tools/et_conv_lds_order.py [rows]"""
import sys, time
sys.path.insert(0, '.')
import torch
from roreg_amd import hip, synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
cfg = default_config(ET='yohoo')
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202)); et = et.cuda().eval()
res = et.PartII_SO3_Conv_layers[0]
ga, gb, gc, p0, gmap = et._pruned_gathers()
h = torch.randn(B, 256, 48, device='cuda'); h[:, :, 45:] = 0
ah = h.abs().amax(dim=(1, 2)).contiguous()
outs = {}
for name, order in (('natural', None), ('searched', et._trunk_lds_order()), ('natural', None), ('searched', et._trunk_lds_order())):
    for _ in range(3):
        out = res._b_in(h, gather=gb, in_rowmax=ah, want_rowmax=True, lds_order=order)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = res._b_in(h, gather=gb, in_rowmax=ah, want_rowmax=True, lds_order=order)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * 512 * 13 * 256 * 13 * B
    print(f'{name:9s} rows={B}: {ms:.3f} ms   {3 * fl / ms / 1e9:.1f} TFLOP/s executed = {3 * fl / ms / 1e9 / 2500:.3f} of the fp16 peak', flush=True)
    outs[name] = out
print('bitwise equal:', torch.equal(outs['natural'][0], outs['searched'][0]) and torch.equal(outs['natural'][1], outs['searched'][1]))
