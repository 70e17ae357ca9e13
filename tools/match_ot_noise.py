"""How far is the reference's own float32 Match_ot from exact arithmetic?  Runs the imported reference (build container only, like
tools/gen_golden.py) on the two golden Match_ot cases in float32 and in float64 and prints / stores max |f32 - f64| per output:
the float32 evaluation noise that any other float32 implementation of the same graph shares.  tests/golden/match_ot_noise.json holds the
numbers; the GPU tests take max(1e-4, 4 x noise) as their tolerance for the outputs whose noise exceeds SURVEY 8c's 1e-4.

    python tools/match_ot_noise.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg                                   # noqa: E402
from gen_golden import synth, name2network, REF, OUT      # noqa: E402


def run(net, batch, dtype):
    net = net.to(dtype)
    b = {k: v.to(dtype) for k, v in batch.items()}
    with torch.no_grad():
        r = net(b)
    return {k: v.double().numpy() for k, v in r.items() if v.dtype.is_floating_point}, {k: v.numpy() for k, v in r.items() if not v.dtype.is_floating_point}


def main():
    cfg = gg.make_cfg('/tmp/match_ot_noise_cfg')
    net = name2network['RM_test'](cfg)
    ck = torch.load(f'{REF}/checkpoints/FCGF/RM/model_best.pth')
    net.load_state_dict(ck['network_state_dict'], strict=True); net.eval()
    cases = {}
    z = np.load(os.path.join(OUT, 'match_ot.npz'))
    cases['golden_120x112'] = {k: torch.from_numpy(z[k]) for k in ('feats0', 'feats1', 'keys0', 'keys1')}
    for tag, name in (('full_match_ot', 'full_2500x2500'), ('full_match_ot_5000', 'full_5000x5000')):
        zf = np.load(os.path.join(OUT, f'{tag}.npz'))
        n = int(zf['n'])
        ds = synth.make_scene(int(zf['scene_seed']), n_clouds=2, n_kpts=n, overlap=0.6, coord_noise=0.005, portable=True)
        f0 = ds.feats[0]; f1 = ds.feats[1]
        f0 = f0 / np.sqrt((f0 * f0).sum(1, keepdims=True)); f1 = f1 / np.sqrt((f1 * f1).sum(1, keepdims=True))
        cases[name] = {'feats0': torch.from_numpy(f1[None].copy()), 'feats1': torch.from_numpy(f0[None].copy()),
                       'keys0': torch.from_numpy(ds.get_kps('1').astype(np.float32)[None].copy()),
                       'keys1': torch.from_numpy(ds.get_kps('0').astype(np.float32)[None].copy())}
    out = {}
    for name, batch in cases.items():
        f32, i32 = run(net, batch, torch.float32)
        f64, i64 = run(net, batch, torch.float64)
        res = {k: float(np.abs(f32[k] - f64[k]).max()) for k in f32 if k in f64 and f32[k].shape == f64[k].shape}
        d = np.abs(f32['scores'] - f64['scores'])[0]
        # how the float32 evaluation noise is distributed: a top-k neighbour that flips between the two precisions moves whole rows / columns
        res['scores_median'] = float(np.median(d)); res['scores_p99'] = float(np.quantile(d, 0.99)); res['scores_p9999'] = float(np.quantile(d, 0.9999))
        res['scores_rows_above_1e-4'] = int((d.max(1) > 1e-4).sum()); res['scores_cols_above_1e-4'] = int((d.max(0) > 1e-4).sum())
        res['scores_fraction_above_1e-4'] = float((d > 1e-4).mean())
        # ... and how it is SIGNED: when one point's descriptor moves, the iteration's mass balance shifts every other log-coupling by a common
        # offset -- at 5000 x 5000 the difference is mostly that offset (median of the signed difference ~ median of its magnitude)
        sd = (f32['scores'] - f64['scores'])[0]
        res['scores_signed_median'] = float(np.median(sd)); res['scores_signed_mean'] = float(sd.mean())
        res['scores_spread_around_the_signed_median'] = float(np.median(np.abs(sd - np.median(sd))))
        res['matches0_equal'] = bool(np.array_equal(i32['matches0'], i64['matches0']))
        res['matches0_differing'] = int((i32['matches0'] != i64['matches0']).sum())
        out[name] = res
        print(name, json.dumps(res, indent=1))
    net.to(torch.float32)
    json.dump(out, open(os.path.join(OUT, 'match_ot_noise.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
