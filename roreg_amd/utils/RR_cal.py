"""3DMatch / Redwood registration-recall benchmark (mirror of utils/RR_cal.py:12-398; host numpy, evaluation only).

`nibabel.quaternions.mat2quat` (RR_cal.py:61) is restated through quaternion_from_matrix: both build the same
symmetric 4x4 K matrix and return the eigenvector of its largest eigenvalue with w >= 0."""
import os

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
    """Redwood's information-weighted pose error of a relative transform: e = (translation, vector part of the rotation's unit quaternion),
    e^T info e / info[0, 0]  (RR_cal.py:47-65)."""
    e = np.concatenate([trans[:3, 3], mat2quat(trans[:3, :3])[1:]], axis=0)
    return (e.reshape(1, 6) @ info @ e.reshape(6, 1) / info[0, 0]).item()


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


# result.txt, block per scene and closing block (text contract of RR_cal.py:332-397, including its missing line break after the first
# "Mean precision" and the upper-case F of the RTE lines)
_SCENE_TEXT = ("{name}\t {prec:.3f}\t {rec:.3f}\t {re_med:.3f}\t {te_med:.3f}\t {n:3d}\n"
               "Mean precision: {prec:.3f}Registration Recall: {rec:.3f}\n"
               "Mean median RRE: {re_mean:.3f}: +- {re_med:.3f}\n"
               "Mean median RTE: {te_mean:.3F}: +- {te_med:.3f}\n")
_TOTAL_TEXT = ("Mean precision: {prec_mean:.3f}: +- {prec_std:.3f}\n"
               "Weighted precision: {prec_w:.3f}\n"
               "Registration Recall: {rec_mean:.3f}: +- {rec_std:.3f}\n"
               "Mean median RRE: {re_mean:.3f}: +- {re_std:.3f}\n"
               "Mean median RTE: {te_mean:.3F}: +- {te_std:.3f}\n")


def _scene_recall(cfg, dataset, keynum, max_iter, yoho_sign, nonconsecutive):
    """One scene of the Redwood protocol from its pre.log, gt.log and gt.info -> dict of the per-scene figures (RR_cal.py:338-386)."""
    gt_stem = dataset.gt_dir[:dataset.gt_dir.rfind('.')]
    gt_pairs, gt_traj = read_trajectory(f'{gt_stem}.log')
    n_fragments, gt_cov = read_trajectory_info(f'{gt_stem}.info')
    est_pairs, est_traj = read_pre_trajectory(f'{cfg.output_cache_fn}/{dataset.name}/match_{keynum}/{yoho_sign}/{max_iter}iters/pre.log')
    n_valid = sum(1 for i, j, _ in gt_pairs if not nonconsecutive or abs(int(i) - int(j)) > 1)
    prec, rec, flags, errors = evaluate_registration(n_fragments, est_traj, est_pairs, gt_pairs, gt_traj, gt_cov, err2=cfg.tau_3,
                                                     nonconsecutive=nonconsecutive)
    gt_of_est = extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj)
    hit = np.array(flags) == 0
    re = rotation_error(gt_of_est[:, 0:3, 0:3], est_traj[:, 0:3, 0:3])[hit]
    te = translation_error(gt_of_est[:, 0:3, 3:4], est_traj[:, 0:3, 3:4])[hit]
    if re.shape[0] == 0:                                           # no success in the scene: the reference's placeholders
        re = np.ones([n_valid]) * 180
    if te.shape[0] == 0:
        te = np.ones([n_valid])
    return {'name': dataset.name, 'prec': prec, 'rec': rec, 'n': int(n_valid), 'flags': flags, 'errors': errors,
            're_mean': np.mean(re), 're_med': np.median(re), 'te_mean': np.mean(te), 'te_med': np.median(te)}


def benchmark(cfg, datasets, keynum, max_iter, yoho_sign='YOHO_O'):
    """Registration recall of a whole test set under the Redwood / Predator protocol (RR_cal.py:321-398): the scene mean of the per-scene
    recalls; writes Eval_results/{yoho_sign}_RR/{max_iter}iters/result.txt.  -> (recall, {scene: flags}, {scene: errors})."""
    wholesetname = datasets['wholesetname']
    result_dir = f'{cfg.output_cache_fn}/{wholesetname}/Eval_results/{yoho_sign}_RR/{max_iter}iters'
    os.makedirs(result_dir, exist_ok=True)
    scenes = [_scene_recall(cfg, ds, keynum, max_iter, yoho_sign, nonconsecutive=wholesetname != 'WHU-TLS')
              for key, ds in datasets.items() if key != 'wholesetname']
    prec = np.array([sc['prec'] for sc in scenes]); rec = np.array([sc['rec'] for sc in scenes]); n = np.array([sc['n'] for sc in scenes])
    re_med = [sc['re_med'] for sc in scenes]; te_med = [sc['te_med'] for sc in scenes]
    with open(f'{result_dir}/result.txt', 'w') as out:
        out.write("Scene\t prec.\t rec.\t re\t te\t samples\t\n")
        for sc in scenes:
            out.write(_SCENE_TEXT.format(**sc))
        out.write(_TOTAL_TEXT.format(prec_mean=np.mean(prec), prec_std=np.std(prec), prec_w=(n * prec).sum() / np.sum(n), rec_mean=np.mean(rec),
                                     rec_std=np.std(rec), re_mean=np.mean(re_med), re_std=np.std(re_med), te_mean=np.mean(te_med), te_std=np.std(te_med)))
    return np.mean(rec), {sc['name']: sc['flags'] for sc in scenes}, {sc['name']: sc['errors'] for sc in scenes}
