"""One GF forward on a 5000-keypoint cloud (the dominant group-conv launches), for rocprofv3 --pmc passes."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
gf = name2network['GF_test'](default_config()); synth.seeded_state_dict(gf, 101)
x = torch.from_numpy(np.random.default_rng(0).standard_normal((5000, 32, 60)).astype(np.float32)).cuda()
with torch.no_grad():
    for _ in range(3):
        y = gf(x)['eqv']
torch.cuda.synchronize()
print('ok', float(y.abs().mean()))
