"""A few launches of the ET trunk's stencil convolution (group_conv_split_kernel, 256 -> 512 channels, 32000 rows) for profiler runs."""
import sys
sys.path.insert(0, '.')
import torch
from roreg_amd import hip, synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
cfg = default_config(ET='yohoo')
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202)); et = et.cuda().eval()
res = et.PartII_SO3_Conv_layers[0]
ga, gb, gc, p0, gmap = et._pruned_gathers()
B = 32000
torch.manual_seed(0)
h = torch.randn(B, 256, 48, device='cuda'); ah = h.abs().amax(dim=(1, 2)).contiguous()
for _ in range(6): out = res._b_in(h, gather=gb, in_rowmax=ah, want_rowmax=True)
torch.cuda.synchronize()
import hashlib
print('checksum', [hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:16] for t in (out if isinstance(out, tuple) else (out,))])
