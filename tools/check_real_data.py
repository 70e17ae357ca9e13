"""Manifest check before a REAL-DATA run (3DMatch / 3DLoMatch / ETH + the GF / ET / RD / RM checkpoints): says what is mounted where, what is
missing, and which command to run against which published numbers.  Host only (no GPU, no reference checkout needed).

    python tools/check_real_data.py [--testset 3dmatch|3dLomatch|ETH|demo] [--origin_data_dir ./data/origin_data]
                                    [--output_cache_fn ./data/YOHO_FCGF] [--model_fn ./checkpoints/FCGF] [--keynum 5000]

Layout (the reference's, README.md:90-106 and parses_test.py): {origin_data_dir}/{testset}/{scene}/PointCloud/{cloud_bin_i.ply, gt.log, gt.info},
{origin_data_dir}/{testset}/{scene}/Keypoints/cloud_bin_iKeypoints.txt (5000 rows), {model_fn}/{GF,ET,RD,RM}/model_best.pth, and the
backbone's group features {output_cache_fn}/{testset}/{scene}/FCGF_Input_Group_feature/{i}.npy float32 [5000,32,60] written by the
reference's `python testset.py --dataset <testset>` (MinkowskiEngine, out of scope here) or by roreg_amd.testset.write_group_features with
a plug-in backbone.  3dLomatch reads 3dmatch's feature directories and its own gtLo.log / gtLo.info."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# published accuracy (media/results.jpg of the reference), full RoReg = --RD --RM --ET yohoo, per keypoint count
PUBLISHED = {
    '3dmatch': {'RR': {5000: 92.9, 2500: 93.2, 1000: 92.7, 500: 93.3, 250: 91.2}, 'FMR': {5000: 98.2, 2500: 97.9, 1000: 98.2, 500: 97.8, 250: 97.2},
                'IR': {5000: 81.6, 2500: 80.2, 1000: 75.1, 500: 74.1, 250: 75.2}},
    '3dLomatch': {'RR': {5000: 70.3, 2500: 71.2, 1000: 69.5, 500: 67.9, 250: 64.3}, 'FMR': {5000: 82.1, 2500: 82.1, 1000: 81.7, 500: 81.6, 250: 80.2},
                  'IR': {5000: 39.6, 2500: 39.6, 1000: 34.0, 500: 31.9, 250: 34.5}},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--testset', default='3dmatch')
    ap.add_argument('--origin_data_dir', default='./data/origin_data')
    ap.add_argument('--output_cache_fn', default='./data/YOHO_FCGF')
    ap.add_argument('--model_fn', default='./checkpoints/FCGF')
    ap.add_argument('--keynum', type=int, default=5000)
    ap.add_argument('--gpus', type=int, default=8)
    a = ap.parse_args()
    from roreg_amd.dataops.dataset import get_dataset_name
    problems, notes = [], []
    for d in ('GF', 'ET', 'RD', 'RM'):
        fn = f'{a.model_fn}/{d}/model_best.pth'
        (notes if os.path.exists(fn) else problems).append(f'checkpoint {fn}' + ('' if os.path.exists(fn) else ' MISSING' + (
            ' (absent from the public checkout too: .MISSING_LARGE_BLOBS)' if d in ('GF', 'ET') else '')))
    try:
        datasets = get_dataset_name(a.testset, a.origin_data_dir)
    except Exception as e:                                          # an unknown test set name
        print(f'cannot enumerate test set {a.testset!r}: {e}'); return 2
    scenes = [s for s in datasets if s not in ('wholesetname', 'valscenes')]
    n_pairs = n_clouds = 0
    for s in scenes:
        ds = datasets[s]
        gt = ds.gt_dir
        if not os.path.exists(gt):
            problems.append(f'{gt} MISSING'); continue
        info = gt[:gt.rfind('.')] + '.info'
        if not os.path.exists(info):
            problems.append(f'{info} MISSING (RR(predator) needs the covariances; FMR / IR / RR(pointdsc) do not)')
        try:
            pairs = ds.pair_ids; clouds = ds.pc_ids
        except Exception as e:
            problems.append(f'{s}: cannot parse {gt}: {e}'); continue
        n_pairs += len(pairs); n_clouds += len(clouds)
        name = f'3d{ds.name[4:]}' if ds.name[0:4] == '3dLo' else ds.name
        fdir = f'{a.output_cache_fn}/{name}/FCGF_Input_Group_feature'
        root = os.path.dirname(os.path.dirname(gt))
        for pc in clouds:
            kp = f'{root}/Keypoints/cloud_bin_{pc}Keypoints.txt'
            ply = f'{root}/PointCloud/cloud_bin_{pc}.ply'
            if not os.path.exists(kp) and not os.path.exists(f'{root}/Keypoints_PC/cloud_bin_{pc}Keypoints.npy'):
                problems.append(f'{kp} MISSING')
            if not os.path.exists(ply) and not os.path.exists(f'{root}/Keypoints_PC/cloud_bin_{pc}Keypoints.npy'):
                problems.append(f'{ply} MISSING (needed once, to turn keypoint indices into coordinates)')
            f = f'{fdir}/{pc}.npy'
            if not os.path.exists(f):
                problems.append(f'{f} MISSING (backbone group feature)')
            else:
                arr = np.load(f, mmap_mode='r')
                if arr.dtype != np.float32 or arr.ndim != 3 or arr.shape[1:] != (32, 60):
                    problems.append(f'{f}: expected float32 [N,32,60], found {arr.dtype} {arr.shape}')
    print(f'test set {a.testset}: {len(scenes)} scenes, {n_clouds} clouds, {n_pairs} pairs '
          f'(the 3DMatch benchmark: 8 scenes, 433 clouds, 1623 pairs)')
    for n in notes:
        print('  ok     ', n)
    for p in problems[:40]:
        print('  PROBLEM', p)
    if len(problems) > 40:
        print(f'  ... and {len(problems) - 40} more')
    extra = ' --tau_2 0.2 --tau_3 0.5 --ransac_ird 0.5' if a.testset == 'ETH' else ''
    kn = a.keynum
    print('\ncommands (the reference\'s flags; README.md:141-176):')
    print(f'  file-coupled, one GPU  : python -m roreg_amd.dropin Test.py --RD --RM --ET yohoo --keynum {kn} --testset {a.testset}{extra}')
    print(f'  device-resident engine : python -m torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 -m roreg_amd.run_distributed '
          f'--RD --RM --ET yohoo --keynum {kn} --testset {a.testset}{extra} --seed 0')
    pub = PUBLISHED.get(a.testset)
    if pub and kn in pub['RR']:
        print(f'\nexpected (published, full RoReg at {kn} keypoints): registration recall {pub["RR"][kn]} %, feature matching recall {pub["FMR"][kn]} %, '
              f'inlier ratio {pub["IR"][kn]} %   -- results.log prints the same three lines; a run within ~0.5 points (RANSAC sampling noise) reproduces the paper')
    elif a.testset == 'ETH':
        print('\nexpected: the paper reports generalisation to ETH with --tau_2 0.2 --tau_3 0.5 --ransac_ird 0.5 (no table in the checkout: compare with the reference run on the same box)')
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
