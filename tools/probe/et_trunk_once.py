"""A few passes of the ET network's trunk + head (f16x2 mode, the packed trunk convolution; 32000 rows) for profiler runs."""
import sys
sys.path.insert(0, '.')
import torch
from roreg_amd import hip, synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
cfg = default_config(ET='yohoo')
et = name2network['ET_test'](cfg); et.load_state_dict(synth.seeded_state_dict(et, 202)); et = et.cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((B, 128, 60), device='cuda', generator=g)
x = x / x.norm(dim=1, keepdim=True)
with torch.no_grad():
    for _ in range(6):
        q = et.trunk_and_head(x)
torch.cuda.synchronize()
print('ok', float(q.abs().mean()))
