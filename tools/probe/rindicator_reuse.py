"""How often does the second Merge_info_block ask R_indicator for the SAME (point, nearest neighbour) pair as the first one did?
(R_indicator depends on the two clouds' fixed extractor outputs and the neighbour index only: an unchanged pair's row could be copied.)"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roreg_amd import hip, synth
from roreg_amd.engine import RegistrationEngine
from roreg_amd.network import name2network, rot_coh_match
from roreg_amd.parses.parses_test import default_config
import bench
cfg = default_config(keynum=5000, max_iter=1000, ET='yohoo', RD=True, RM=True)
gf = name2network['GF_test'](cfg); synth.seeded_state_dict(gf, 101)
et = name2network['ET_test'](cfg); synth.seeded_state_dict(et, 202)
rd, rm, _ = bench.rd_rm_nets(cfg)
eng = RegistrationEngine(cfg, gf, et, rd_net=rd, rm_net=rm)
calls = []
orig = rot_coh_match.hip.group_corr
def spy(perm_feats, bcast_feats, perm_rows=None, bcast_rows=None, **kw):
    calls.append((perm_rows if perm_rows is not None else bcast_rows).clone())
    return orig(perm_feats, bcast_feats, perm_rows=perm_rows, bcast_rows=bcast_rows, **kw)
rot_coh_match.hip.group_corr = spy
for seed, overlap in ((500, 0.6), (501, 0.3)):
    feats, keys, poses = synth.make_scene_device(seed, 24, 5000, overlap)
    pairs = [(str(a), str(b)) for a, b in synth.scene_pair_list(24, 60, 900, locality=8.0)]
    seeds = [(7 + zlib.crc32(f'k:{a}:{b}'.encode())) % (2 ** 32) for a, b in pairs]
    calls.clear()
    eng.run_scene(feats, keys, pairs, pair_seeds=seeds)
    torch.cuda.synchronize()
    print(f'scene overlap {overlap}: {len(calls)} R_indicator calls')
    for g in range(0, len(calls), 4):
        a, b, c, d = calls[g:g + 4]
        print(f'   group {g // 4}: {a.numel()} rows; same neighbour in block 2 as in block 1: source side {float((a == c).float().mean()):.3f}, target side {float((b == d).float().mean()):.3f}')
