"""Rotation metrics under the reference's names (utils/r_eval.py:5-115); host numpy, used by the evaluation and by the
quaternion -> matrix step of the local-transform assembly."""
import math

import numpy as np


def _quaternion_precise(M):
    """Trace / largest-diagonal branch for exact homogeneous rotations (r_eval.py:43-65)."""
    if M.shape != (4, 4):
        H = np.eye(4)
        H[:M.shape[0], :M.shape[1]] = M
        M = H
    q = np.empty((4,))
    t = np.trace(M)
    if t > M[3, 3]:
        q[:] = (t, M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1])
    else:
        i, j, k = 1, 2, 3
        if M[1, 1] > M[0, 0]:
            i, j, k = 2, 3, 1
        if M[2, 2] > M[i, i]:
            i, j, k = 3, 1, 2
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i], q[j], q[k], q[3] = t, M[i, j] + M[j, i], M[k, i] + M[i, k], M[k, j] - M[j, k]
    return q * (0.5 / math.sqrt(t * M[3, 3]))


def quaternion_from_matrix(matrix, isprecise=False):
    """(w, x, y, z) of a rotation matrix, w >= 0.  Default: the eigenvector, for the largest eigenvalue, of the symmetric 4x4 matrix
    built from the nine entries (lower triangle filled, divided by 3; r_eval.py:66-88) -- robust to slightly non-orthonormal input."""
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    if isprecise:
        q = _quaternion_precise(M)
    else:
        (a, b, c), (d, e, f), (g, h, i) = M[0, :3], M[1, :3], M[2, :3]
        K = np.zeros((4, 4))
        K[0, 0] = a - e - i
        K[1, 0], K[1, 1] = b + d, e - a - i
        K[2, 0], K[2, 1], K[2, 2] = c + g, f + h, i - a - e
        K[3, 0], K[3, 1], K[3, 2], K[3, 3] = h - f, c - g, d - b, a + e + i
        K /= 3.0
        w, V = np.linalg.eigh(K)                       # reads the lower triangle
        q = V[[3, 0, 1, 2], np.argmax(w)]
    if q[0] < 0.0:
        np.negative(q, q)
    return q


def matrix_from_quaternion(quaternion):
    """Rotation matrix of a unit quaternion (w, x, y, z) (r_eval.py:90-106); every entry keeps the reference's evaluation order."""
    w, x, y, z = (quaternion[k] for k in range(4))
    mat = np.eye(3)
    mat[0] = (1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w)
    mat[1] = (2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w)
    mat[2] = (2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y)
    return mat


def compute_R_diff(R_gt, R):
    """Angle in degrees between two rotations through their quaternions (r_eval.py:108-115)."""
    eps = 1e-15
    qa, qb = quaternion_from_matrix(R_gt), quaternion_from_matrix(R)
    qb = qb / (np.linalg.norm(qb) + eps)
    qa = qa / (np.linalg.norm(qa) + eps)
    loss = np.maximum(eps, (1.0 - np.sum(qb * qa) ** 2))
    return np.rad2deg(np.abs(np.arccos(1 - 2 * loss)))
