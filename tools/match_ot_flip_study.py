"""How stable are the reference's OWN matches under float32 rounding?  For each config-4 fixture (tests/golden/full_pipeline_rd_rm*.npz) the
imported reference (build container only, like tools/gen_golden.py) is run up to the matcher -- GF, detector, NMS sampling -- and then
Match_ot once in float32 (what the fixtures hold) and once in float64 on the SAME float32 inputs: rows of the match list that differ, and
the largest difference of the matching scores on the common rows.  That is the noise any other float32 evaluation of the same graph shares
(top-k neighbour selections flip on near-ties); tests/golden/match_ot_flip_study.json holds the numbers, DESIGN.md section 2 quotes them and
tests/test_hip_fullsize.py::test_full_pipeline_rd_rm_vs_reference takes its stage-wise tolerances from them.

    python tools/match_ot_flip_study.py
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg                                   # noqa: E402
from gen_golden import synth, name2network, name2extractor, REF, OUT      # noqa: E402

TAGS = ['full_pipeline_rd_rm', 'full_pipeline_rd_rm_o60', 'full_pipeline_rd_rm_o60_s1', 'full_pipeline_rd_rm_o60_s2', 'full_pipeline_rd_rm_o60_s3',
        'full_pipeline_rd_rm_k5000']


def main():
    out = {}
    for tag in TAGS:
        z = np.load(os.path.join(OUT, tag + '.npz'))
        kn = int(z['keynum']) if 'keynum' in z.files else 2500
        root = tempfile.mkdtemp(prefix='flip_')
        try:
            cfg = gg.make_cfg(root, RD=True, RM=True, keynum=kn, match_n=0.5)
            cfg.bs_GF = 250
            ds = synth.make_scene(int(z['scene_seed']), n_clouds=2, n_kpts=5000, overlap=float(z['overlap']), coord_noise=0.005, name='synth/scene0', portable=True)
            ds.write_inputs(cfg.output_cache_fn)
            name2extractor['yoho_des'](cfg).run(ds)
            base = f'{cfg.output_cache_fn}/{ds.name}'
            f = [np.load(f'{base}/YOHO_Output_Group_feature/{pc}.npy') for pc in ds.pc_ids]
            s = [z[f'nms_{pc}'].astype(np.int64) for pc in ds.pc_ids]                       # the reference's own samples (stored)
            k = [ds.get_kps(pc) for pc in ds.pc_ids]
            net = name2network['RM_test'](cfg)
            ck = torch.load(f'{REF}/checkpoints/FCGF/RM/model_best.pth')
            net.load_state_dict(ck['network_state_dict'], strict=True); net.eval()
            batch = {'feats0': torch.from_numpy(f[1][s[1]][None].astype(np.float32)), 'feats1': torch.from_numpy(f[0][s[0]][None].astype(np.float32)),
                     'keys0': torch.from_numpy(k[1][s[1]][None].astype(np.float32)), 'keys1': torch.from_numpy(k[0][s[0]][None].astype(np.float32))}
            res = {}
            for dt in (torch.float32, torch.float64):
                net = net.to(dt)
                with torch.no_grad():
                    r = net({kk: v.to(dt) for kk, v in batch.items()})
                m0 = r['matches0'][0].numpy(); sc = r['matching_scores0'][0].double().numpy()
                res[dt] = (m0, sc)
            net.to(torch.float32)
            m32, s32 = res[torch.float32]; m64, s64 = res[torch.float64]
            rows32 = {(int(i), int(j)) for i, j in enumerate(m32) if j >= 0}; rows64 = {(int(i), int(j)) for i, j in enumerate(m64) if j >= 0}
            common = m32 == m64
            # consistency with the stored fixture: the float32 run above must reproduce the fixture's match list
            want = {(int(b), int(a)) for a, b in z['match_0_1'].astype(np.int64)}           # (fixture rows are (pc0 row, pc1 row) in cloud coordinates)
            mine = {(int(s[1][i]), int(s[0][j])) for i, j in rows32}
            out[tag] = {'keynum': kn, 'matches_f32': len(rows32), 'matches_f64': len(rows64), 'rows_only_in_f32': len(rows32 - rows64), 'rows_only_in_f64': len(rows64 - rows32),
                        'max_score_diff_on_common_rows': float(np.abs(s32 - s64)[common & (m32 >= 0)].max(initial=0.0)),
                        'f32_run_reproduces_fixture': mine == want}
            print(tag, json.dumps(out[tag]))
        finally:
            shutil.rmtree(root, ignore_errors=True)
    json.dump(out, open(os.path.join(OUT, 'match_ot_flip_study.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
