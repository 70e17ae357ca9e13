"""How stable is the reference's OWN `--RD --RM --ET yohoo` chain under float32 rounding of the detector?  For each config-4 fixture
(tests/golden/full_pipeline_rd_rm*.npz = the imported reference's float32 run, end to end) the imported reference (build container only, like
tools/gen_golden.py) is run once more on the same inputs with ONE change: its detector network evaluates in float64 (network/rot_detect.py:43-55
on the float32 extractor output) before the rank transform of test/detector.py:45-46.  Everything after it is the reference's own code: NMS
sampling (test/matcher.py:11-42), yoho_mat (:152-210), yohoo (test/estimator.py:405-443), same generator seeds as the fixture.  Stored per
fixture: how far the ranks move, how many NMS samples the two runs share, how many match rows, and the distance of the final transforms --
the noise level of the reference against itself, which tests/test_hip_fullsize.py::test_full_pipeline_rd_rm_vs_reference uses as its bar
for this build's end-to-end run (tests/golden/rd_chain_flip_study.json).

    python tools/rd_chain_flip_study.py [tag ...]
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
from gen_golden import synth, name2network, name2extractor, name2matcher, name2estimator, REF, OUT      # noqa: E402

TAGS = ['full_pipeline_rd_rm', 'full_pipeline_rd_rm_o60', 'full_pipeline_rd_rm_o60_s1', 'full_pipeline_rd_rm_o60_s2', 'full_pipeline_rd_rm_o60_s3',
        'full_pipeline_rd_rm_k5000']


def main():
    path = os.path.join(OUT, 'rd_chain_flip_study.json')
    out = json.load(open(path)) if os.path.exists(path) else {}
    for tag in (sys.argv[1:] or TAGS):
        z = np.load(os.path.join(OUT, tag + '.npz'))
        kn = int(z['keynum']) if 'keynum' in z.files else 2500
        root = tempfile.mkdtemp(prefix='rdflip_')
        try:
            cfg = gg.make_cfg(root, RD=True, RM=True, keynum=kn, match_n=0.5)
            cfg.bs_GF = 250; cfg.bs_ET = 500
            ds = synth.make_scene(int(z['scene_seed']), n_clouds=2, n_kpts=5000, overlap=float(z['overlap']), coord_noise=0.005, name='synth/scene0', portable=True)
            ds.write_inputs(cfg.output_cache_fn)
            name2extractor['yoho_des'](cfg).run(ds)
            base = f'{cfg.output_cache_fn}/{ds.name}'
            # the detector in float64 on the float32 extractor output, then the reference's rank transform
            net = name2network['RD_test'](cfg)
            net.load_state_dict(torch.load(f'{REF}/checkpoints/FCGF/RD/model_best.pth')['network_state_dict'], strict=True)
            net = net.double().eval()
            os.makedirs(f'{base}/det_score')
            shift = []
            for pc in ds.pc_ids:
                feats = np.load(f'{base}/YOHO_Output_Group_feature/{pc}.npy')
                with torch.no_grad():
                    scores = net({'feats': torch.from_numpy(feats.astype(np.float64))})['scores'].cpu().numpy()
                arg = np.argsort(scores)
                scores[arg] = np.arange(scores.shape[0]) / scores.shape[0]
                np.save(f'{base}/det_score/{pc}.npy', scores.astype(np.float32))
                rank64 = np.rint(scores * scores.shape[0]).astype(np.int64)
                shift.append(int(np.abs(rank64 - z[f'det_rank_{pc}'].astype(np.int64)).max()))
            np.random.seed(1234)
            name2matcher['yoho_mat'](cfg).run(ds, kn)
            np.random.seed(4321)
            name2estimator['yohoo'](cfg).run(ds, kn, 1000)
            RefNMS = gg.ref_mat.NMS_sample
            nms_shared = []
            for pc in ds.pc_ids:
                s64 = RefNMS(kn, 5).sample(ds.get_kps(pc), np.load(f'{base}/det_score/{pc}.npy'))
                nms_shared.append(int(np.intersect1d(s64, z[f'nms_{pc}'].astype(np.int64)).shape[0]))
            md = f'{base}/match_{kn}'
            a, b = ds.pair_ids[0]
            m64 = np.load(f'{md}/{a}-{b}.npy'); m32 = z[f'match_{a}_{b}'].astype(np.int64)
            rows64 = {tuple(r) for r in m64.tolist()}; rows32 = {tuple(r) for r in m32.tolist()}
            T64 = np.load(f'{md}/yohoo/1000iters/{a}-{b}.npz', allow_pickle=True)['trans']; T32 = z[f'trans_{a}_{b}']
            gt = ds.get_transform(a, b)
            ok = lambda T: bool(gg.ref_reval.compute_R_diff(gt[:3, :3], T[:3, :3]) < 15 and np.linalg.norm(gt[:3, 3] - T[:3, 3]) < 0.3)
            out[tag] = {'keynum': kn, 'rank_shift_max': max(shift), 'nms_samples': kn, 'nms_shared': nms_shared, 'rows_f32': len(rows32), 'rows_f64_detector': len(rows64),
                        'rows_shared': len(rows32 & rows64), 'max_abs_diff_of_transforms': float(np.abs(T64[:3] - T32[:3]).max()),
                        'registered_f32': ok(T32), 'registered_f64_detector': ok(T64)}
            print(tag, json.dumps(out[tag]), flush=True)
            json.dump(out, open(path, 'w'), indent=1)
        finally:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
