"""Experiment: group-feature extraction of 80000 keypoints as one sequence of launches vs two halves on two HIP streams
(GEMMs of one half overlapping the memory-bound transforms of the other)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from roreg_amd import synth
from roreg_amd.network import name2network
from roreg_amd.parses.parses_test import default_config
cfg = default_config()
gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
net = gf.PartI_net
x = torch.randn((80000, 32, 60), device='cuda'); x = x / x.norm(dim=1, keepdim=True)
def seq(chunks):
    return [net(c, want_inv=False)['eqv'] for c in chunks]
def par(chunks, streams):
    main = torch.cuda.current_stream(); outs = []
    for c, st in zip(chunks, streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(net(c, want_inv=False)['eqv'])
    for st in streams:
        main.wait_stream(st)
    return outs
streams = [torch.cuda.Stream() for _ in range(4)]
with torch.no_grad():
    for name, fn in (('1 x 65536 + 14464 sequential', lambda: seq([x[:65536], x[65536:]])),
                     ('2 x 40000 sequential', lambda: seq([x[:40000], x[40000:]])),
                     ('2 x 40000 on 2 streams', lambda: par([x[:40000], x[40000:]], streams[:2])),
                     ('4 x 20000 on 4 streams', lambda: par([x[i * 20000:(i + 1) * 20000] for i in range(4)], streams)),
                     ('4 x 20000 on 2 streams', lambda: par([x[i * 20000:(i + 1) * 20000] for i in range(4)], [streams[0], streams[1], streams[0], streams[1]]))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f'{name}: {dt * 1e3:.2f} ms')
