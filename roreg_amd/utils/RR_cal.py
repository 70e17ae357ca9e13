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
    with open(filename) as f:
        lines = f.readlines()
    keys = lines[0::(dim + 1)]
    final_keys = [[c.strip() for c in k.split('\t')[0:3]] for k in keys]
    traj = [lines[i].split('\t')[0:dim] for i in range(len(lines)) if i % 5 != 0]
    traj = np.asarray(traj, dtype=float).reshape(-1, dim, dim)
    return np.asarray(final_keys), traj


read_pre_trajectory = read_trajectory


def read_trajectory_info(filename, dim=6):
    with open(filename) as fid:
        contents = fid.readlines()
    n_pairs = len(contents) // 7
    assert len(contents) == 7 * n_pairs
    info_list = []
    n_frame = 0
    for i in range(n_pairs):
        frame_idx0, frame_idx1, n_frame = [int(item) for item in contents[i * 7].strip().split()]
        info_matrix = np.concatenate([np.array(item.strip().split(), dtype=float).reshape(1, -1)
                                      for item in contents[i * 7 + 1:i * 7 + 7]], axis=0)
        info_list.append(info_matrix)
    return n_frame, np.asarray(info_list, dtype=float).reshape(-1, dim, dim)


def extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj):
    ext_traj = np.zeros((len(est_pairs), 4, 4))
    for est_idx, pair in enumerate(est_pairs):
        pair[2] = gt_pairs[0][2]
        gt_idx = np.where((gt_pairs == pair).all(axis=1))[0]
        ext_traj[est_idx, :, :] = gt_traj[gt_idx, :, :]
    return ext_traj


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2=0.2, nonconsecutive=True):
    """RR_cal.py:236-317."""
    err2 = err2 ** 2
    gt_mask = np.zeros((num_fragment, num_fragment), dtype=int)
    flags, errors = [], []
    if nonconsecutive:
        for idx in range(gt_pairs.shape[0]):
            i = int(gt_pairs[idx, 0]); j = int(gt_pairs[idx, 1])
            if abs(j - i) > 1:
                gt_mask[i, j] = idx
        n_gt = np.sum(gt_mask > 0)
    else:
        for idx in range(gt_pairs.shape[0]):
            i = int(gt_pairs[idx, 0]); j = int(gt_pairs[idx, 1])
            gt_mask[i, j] = idx
        n_gt = np.sum(gt_mask > 0) + 1
    good = 0
    n_res = 0
    start_check = 0
    if not nonconsecutive:
        start_check = 1
        n_res += 1
        pose = result[0, :, :]
        p = computeTransformationErr(np.linalg.inv(gt[0, :, :]) @ pose, gt_info[0, :, :])
        errors.append(np.sqrt(p))
        if p <= err2:
            good += 1; flags.append(0)
        else:
            flags.append(1)
    for idx in range(start_check, result_pairs.shape[0]):
        i = int(result_pairs[idx, 0]); j = int(result_pairs[idx, 1])
        pose = result[idx, :, :]
        if gt_mask[i, j] > 0:
            n_res += 1
            gt_idx = gt_mask[i, j]
            p = computeTransformationErr(np.linalg.inv(gt[gt_idx, :, :]) @ pose, gt_info[gt_idx, :, :])
            errors.append(np.sqrt(p))
            if p <= err2:
                good += 1; flags.append(0)
            else:
                flags.append(1)
        else:
            flags.append(2)
    if n_res == 0:
        n_res += 1e6
    return good * 1.0 / n_res, good * 1.0 / n_gt, flags, errors


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
