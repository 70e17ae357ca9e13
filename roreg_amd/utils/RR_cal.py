"""3DMatch / Redwood registration-recall benchmark (mirror of utils/RR_cal.py:12-398; host numpy, evaluation only).

`nibabel.quaternions.mat2quat` (RR_cal.py:61) is restated through quaternion_from_matrix: both build the same
symmetric 4x4 K matrix and return the eigenvector of its largest eigenvalue with w >= 0."""
import os
from collections import defaultdict

import numpy as np

from .r_eval import quaternion_from_matrix


def mat2quat(r):
    return quaternion_from_matrix(np.asarray(r, np.float64)[:3, :3])


def rotation_error(R1, R2):
    """degrees, [b,1]; RR_cal.py:13-33."""
    R_ = np.matmul(np.transpose(R1, (0, 2, 1)), R2)
    e = ((np.trace(R_, axis1=1, axis2=2) - 1) / 2)[:, None]
    e = np.clip(e, -1, 1)
    return 180.0 * np.arccos(e) / np.pi


def translation_error(t1, t2):
    return np.sqrt(np.sum(np.square(t1 - t2), axis=(1, 2)))


def computeTransformationErr(trans, info):
    t = trans[:3, 3]
    r = trans[:3, :3]
    q = mat2quat(r)
    er = np.concatenate([t, q[1:]], axis=0)
    p = er.reshape(1, 6) @ info @ er.reshape(6, 1) / info[0, 0]
    return p.item()


def read_trajectory(filename, dim=4):
    """A Redwood .log trajectory: blocks of one header line 'i <tab> j <tab> n' and `dim` matrix rows -> (keys [n,3] str, [n,dim,dim])."""
    with open(filename) as f:
        rows = [line.split('\t') for line in f.readlines()]
    heads = rows[0::dim + 1]
    body = [r[0:dim] for k, r in enumerate(rows) if k % 5 != 0]            # the reference's `i % 5` (dim = 4), RR_cal.py:108
    keys = np.asarray([[c.strip() for c in h[0:3]] for h in heads])
    return keys, np.asarray(body, dtype=float).reshape(-1, dim, dim)


read_pre_trajectory = read_trajectory


def read_trajectory_info(filename, dim=6):
    """A Redwood .info file: blocks of 'i j n_frames' + a dim x dim information matrix -> (n_frames, [n,dim,dim])."""
    with open(filename) as fid:
        lines = fid.readlines()
    block = dim + 1
    assert len(lines) % block == 0
    n_frame, mats = 0, []
    for k in range(0, len(lines), block):
        n_frame = int(lines[k].strip().split()[2])
        mats.append([np.array(l.strip().split(), dtype=float) for l in lines[k + 1:k + block]])
    return n_frame, np.asarray(mats, dtype=float).reshape(-1, dim, dim)


def extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj):
    """Ground-truth pose of every estimated pair (its frame count is overwritten with the ground truth's first, RR_cal.py:226)."""
    out = np.zeros((len(est_pairs), 4, 4))
    for k, pair in enumerate(est_pairs):
        pair[2] = gt_pairs[0][2]
        out[k] = gt_traj[np.where((gt_pairs == pair).all(axis=1))[0]]
    return out


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2=0.2, nonconsecutive=True):
    """Redwood protocol (RR_cal.py:236-317): a pair counts when its ground-truth index is > 0 in the lookup (so the very first
    ground-truth pair is evaluated only in the consecutive protocol, where it is scored up front) and, in the default protocol, when
    its fragments are not consecutive; success = information-weighted pose error <= err2^2.
    -> (precision, recall, flags per estimate: 0 hit / 1 miss / 2 not evaluated, errors)."""
    limit = err2 ** 2
    lookup = np.zeros((num_fragment, num_fragment), dtype=int)
    for k in range(gt_pairs.shape[0]):
        i, j = int(gt_pairs[k, 0]), int(gt_pairs[k, 1])
        if not nonconsecutive or abs(j - i) > 1:
            lookup[i, j] = k
    n_gt = np.sum(lookup > 0) + (0 if nonconsecutive else 1)
    flags, errors = [], []
    tally = {'good': 0, 'seen': 0}

    def score(pose, k):
        p = computeTransformationErr(np.linalg.inv(gt[k, :, :]) @ pose, gt_info[k, :, :])
        errors.append(np.sqrt(p))
        tally['seen'] += 1
        tally['good'] += 1 if p <= limit else 0
        flags.append(0 if p <= limit else 1)

    first = 0
    if not nonconsecutive:
        score(result[0, :, :], 0)
        first = 1
    for idx in range(first, result_pairs.shape[0]):
        k = lookup[int(result_pairs[idx, 0]), int(result_pairs[idx, 1])]
        if k > 0:
            score(result[idx, :, :], k)
        else:
            flags.append(2)
    seen = tally['seen'] if tally['seen'] else 1e6
    return tally['good'] * 1.0 / seen, tally['good'] * 1.0 / n_gt, flags, errors


def benchmark(cfg, datasets, keynum, max_iter, yoho_sign='YOHO_O'):
    """RR_cal.py:321-398: scene-mean registration recall + Eval_results/.../result.txt."""
    c_flags, c_errors = {}, {}
    re_per_scene, te_per_scene = defaultdict(list), defaultdict(list)
    precision, recall, n_valids = [], [], []
    wholesetname = datasets['wholesetname']
    nonconsecutive = wholesetname != 'WHU-TLS'
    result_dir = f'{cfg.output_cache_fn}/{wholesetname}/Eval_results/{yoho_sign}_RR/{max_iter}iters'
    os.makedirs(result_dir, exist_ok=True)
    f = open(f'{result_dir}/result.txt', 'w')
    f.write("Scene\t prec.\t rec.\t re\t te\t samples\t\n")
    for scene, dataset in datasets.items():
        if scene == 'wholesetname':
            continue
        pre_dir = f'{cfg.output_cache_fn}/{dataset.name}/match_{keynum}/{yoho_sign}/{max_iter}iters'
        gt_dir = dataset.gt_dir[0:str.rfind(dataset.gt_dir, '.')]
        gt_pairs, gt_traj = read_trajectory(f'{gt_dir}.log')
        n_valid = 0
        for ele in gt_pairs:
            n_valid += (abs(int(ele[0]) - int(ele[1])) > 1) if nonconsecutive else 1
        n_valids.append(n_valid)
        n_fragments, gt_traj_cov = read_trajectory_info(f'{gt_dir}.info')
        est_pairs, est_traj = read_pre_trajectory(os.path.join(pre_dir, 'pre.log'))
        temp_precision, temp_recall, c_flag, c_error = evaluate_registration(
            n_fragments, est_traj, est_pairs, gt_pairs, gt_traj, gt_traj_cov, err2=cfg.tau_3, nonconsecutive=nonconsecutive)
        c_flags[dataset.name] = c_flag
        c_errors[dataset.name] = c_error
        ext_gt_traj = extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj)
        ok = np.array(c_flag) == 0
        re = rotation_error(ext_gt_traj[:, 0:3, 0:3], est_traj[:, 0:3, 0:3])[ok]
        te = translation_error(ext_gt_traj[:, 0:3, 3:4], est_traj[:, 0:3, 3:4])[ok]
        if re.shape[0] == 0:
            re = np.ones([n_valid]) * 180
        if te.shape[0] == 0:
            te = np.ones([n_valid])
        for name, fn in [('mean', np.mean), ('median', np.median), ('min', np.min), ('max', np.max)]:
            re_per_scene[name].append(fn(re)); te_per_scene[name].append(fn(te))
        precision.append(temp_precision)
        recall.append(temp_recall)
        f.write("{}\t {:.3f}\t {:.3f}\t {:.3f}\t {:.3f}\t {:3d}\n".format(dataset.name, temp_precision, temp_recall, np.median(re), np.median(te), n_valid))
        f.write("Mean precision: {:.3f}".format(temp_precision))
        f.write("Registration Recall: {:.3f}\n".format(temp_recall))
        f.write("Mean median RRE: {:.3f}: +- {:.3f}\n".format(np.mean(re), np.median(re)))
        f.write("Mean median RTE: {:.3F}: +- {:.3f}\n".format(np.mean(te), np.median(te)))
    weighted_precision = (np.array(n_valids) * np.array(precision)).sum() / np.sum(n_valids)
    Registration_Recall = np.mean(np.array(recall))
    f.write("Mean precision: {:.3f}: +- {:.3f}\n".format(np.mean(precision), np.std(precision)))
    f.write("Weighted precision: {:.3f}\n".format(weighted_precision))
    f.write("Registration Recall: {:.3f}: +- {:.3f}\n".format(Registration_Recall, np.std(np.array(recall))))
    f.write("Mean median RRE: {:.3f}: +- {:.3f}\n".format(np.mean(re_per_scene['median']), np.std(re_per_scene['median'])))
    f.write("Mean median RTE: {:.3F}: +- {:.3f}\n".format(np.mean(te_per_scene['median']), np.std(te_per_scene['median'])))
    f.close()
    return Registration_Recall, c_flags, c_errors
