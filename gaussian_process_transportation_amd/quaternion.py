"""Minimal quaternion helpers for transport_orientation (w, x, y, z float arrays).

The reference calls a third-party `Quaternion` module that is not vendored
(policy_transportation.py:9, :70-73: from_float_array, from_rotation_matrix(J, nonorthogonal=True),
quaternion product, as_float_array).  PARITY UNPINNED: no source or fixture for that module exists in
the reference; this file implements the published algorithm such packages use for a non-orthogonal
matrix — Bar-Itzhack (2000), "New method for extracting the quaternion from a rotation matrix":
the unit quaternion of the rotation closest to M is the dominant eigenvector of a symmetric 4x4
matrix built from M."""
import numpy as np


def quaternion_from_nonorthogonal(M):
    M = np.asarray(M, dtype=np.float64)
    batch = M.shape[:-2]
    M = M.reshape(-1, 3, 3)
    K = np.empty((M.shape[0], 4, 4))
    K[:, 0, 0] = M[:, 0, 0] - M[:, 1, 1] - M[:, 2, 2]
    K[:, 1, 1] = M[:, 1, 1] - M[:, 0, 0] - M[:, 2, 2]
    K[:, 2, 2] = M[:, 2, 2] - M[:, 0, 0] - M[:, 1, 1]
    K[:, 3, 3] = M[:, 0, 0] + M[:, 1, 1] + M[:, 2, 2]
    K[:, 0, 1] = K[:, 1, 0] = M[:, 1, 0] + M[:, 0, 1]
    K[:, 0, 2] = K[:, 2, 0] = M[:, 2, 0] + M[:, 0, 2]
    K[:, 1, 2] = K[:, 2, 1] = M[:, 2, 1] + M[:, 1, 2]
    K[:, 0, 3] = K[:, 3, 0] = M[:, 2, 1] - M[:, 1, 2]
    K[:, 1, 3] = K[:, 3, 1] = M[:, 0, 2] - M[:, 2, 0]
    K[:, 2, 3] = K[:, 3, 2] = M[:, 1, 0] - M[:, 0, 1]
    K /= 3.0
    _, vecs = np.linalg.eigh(K)
    v = vecs[:, :, -1]                       # eigenvector of the largest eigenvalue: (x, y, z, w)
    q = np.stack([v[:, 3], v[:, 0], v[:, 1], v[:, 2]], axis=1)
    q[q[:, 0] < 0] *= -1.0                   # canonical sign: w >= 0
    return q.reshape(batch + (4,))


def quaternion_multiply(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    aw, ax, ay, az = np.moveaxis(a, -1, 0)
    bw, bx, by, bz = np.moveaxis(b, -1, 0)
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def rotation_matrix_from_quaternion(q):
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = np.moveaxis(q / np.linalg.norm(q, axis=-1, keepdims=True), -1, 0)
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], axis=-1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], axis=-1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=-1)], axis=-2)
