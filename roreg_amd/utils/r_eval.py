"""Rotation metrics (mirror of utils/r_eval.py:5-115; host numpy, evaluation only)."""
import math

import numpy as np


def quaternion_from_matrix(matrix, isprecise=False):
    """Quaternion (w,x,y,z) of a rotation matrix: eigenvector of the symmetric K matrix for the largest
    eigenvalue (r_eval.py:66-88); the `isprecise` shortcut of :43-65 is kept for API parity."""
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    if isprecise:
        if M.shape != (4, 4):
            Mh = np.eye(4); Mh[:M.shape[0], :M.shape[1]] = M; M = Mh
        q = np.empty((4,))
        t = np.trace(M)
        if t > M[3, 3]:
            q[0] = t; q[3] = M[1, 0] - M[0, 1]; q[2] = M[0, 2] - M[2, 0]; q[1] = M[2, 1] - M[1, 2]
        else:
            i, j, k = 1, 2, 3
            if M[1, 1] > M[0, 0]:
                i, j, k = 2, 3, 1
            if M[2, 2] > M[i, i]:
                i, j, k = 3, 1, 2
            t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
            q[i] = t; q[j] = M[i, j] + M[j, i]; q[k] = M[k, i] + M[i, k]; q[3] = M[k, j] - M[j, k]
        q *= 0.5 / math.sqrt(t * M[3, 3])
    else:
        m00, m01, m02 = M[0, 0], M[0, 1], M[0, 2]
        m10, m11, m12 = M[1, 0], M[1, 1], M[1, 2]
        m20, m21, m22 = M[2, 0], M[2, 1], M[2, 2]
        K = np.array([[m00 - m11 - m22, 0.0, 0.0, 0.0],
                      [m01 + m10, m11 - m00 - m22, 0.0, 0.0],
                      [m02 + m20, m12 + m21, m22 - m00 - m11, 0.0],
                      [m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22]])
        K /= 3.0
        w, V = np.linalg.eigh(K)
        q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        np.negative(q, q)
    return q


def matrix_from_quaternion(quaternion):
    w, x, y, z = quaternion[0], quaternion[1], quaternion[2], quaternion[3]
    mat = np.eye(3)
    mat[0, 0] = 1 - 2 * y * y - 2 * z * z
    mat[0, 1] = 2 * x * y - 2 * z * w
    mat[0, 2] = 2 * x * z + 2 * y * w
    mat[1, 0] = 2 * x * y + 2 * z * w
    mat[1, 1] = 1 - 2 * x * x - 2 * z * z
    mat[1, 2] = 2 * y * z - 2 * x * w
    mat[2, 0] = 2 * x * z - 2 * y * w
    mat[2, 1] = 2 * y * z + 2 * x * w
    mat[2, 2] = 1 - 2 * x * x - 2 * y * y
    return mat


def compute_R_diff(R_gt, R):
    eps = 1e-15
    q_gt = quaternion_from_matrix(R_gt)
    q = quaternion_from_matrix(R)
    q = q / (np.linalg.norm(q) + eps)
    q_gt = q_gt / (np.linalg.norm(q_gt) + eps)
    loss_q = np.maximum(eps, (1.0 - np.sum(q * q_gt) ** 2))
    err_q = np.arccos(1 - 2 * loss_q)
    return np.rad2deg(np.abs(err_q))
