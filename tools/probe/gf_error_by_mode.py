"""GF / ET output error against the reference golden, per matrix-core mode."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from roreg_amd import synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
z = load_golden('gf_forward')
net = name2network['GF_test'](default_config()); synth.seeded_state_dict(net, int(z['seed']))
x = torch.from_numpy(z['x']); net.PartI_net.mode = 'fourier'; net(x)
for mode in ('f32', 'bf16x3', 'f16x2'):
    net.PartI_net._fourier.gemm = mode
    e = net(x)['eqv'].cpu().numpy()
    print(f'GF  {mode:7s}: max |eqv - reference| = {np.abs(e - z["eqv"]).max():.3e}   rms = {np.sqrt(np.mean((e - z["eqv"])**2)):.3e}   (|eqv| max {np.abs(z["eqv"]).max():.3f})')
z = load_golden('et_forward')
net = name2network['ET_test'](default_config()); synth.seeded_state_dict(net, int(z['seed']))
for mode in ('f32', 'bf16x3', 'f16x2'):
    net.gemm = mode
    b = {k: torch.from_numpy(z[k].copy()) for k in ('before_eqv0', 'before_eqv1', 'after_eqv0', 'after_eqv1', 'pre_idx')}
    q = net(b)['quaternion_pre'].cpu().numpy()
    print(f'ET  {mode:7s}: max |q - reference| = {np.abs(q - z["quaternion"]).max():.3e}')
