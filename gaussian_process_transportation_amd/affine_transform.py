"""AffineTransform — rigid (optionally scaled) Procrustes alignment of two matched point sets.
Host numpy (O(N D^2), stays off the GPU); same API and attributes as the reference's
policy_transportation/models/affine_trasformation.py:8-57."""
import numpy as np


class AffineTransform:
    def __init__(self, do_scale=False, do_rotation=True, verbose=True):
        self.do_scale = do_scale
        self.do_rotation = do_rotation
        self.scale = 1
        self.verbose = verbose

    def fit(self, source_points, target_points):
        source_points = np.asarray(source_points, dtype=np.float64)
        target_points = np.asarray(target_points, dtype=np.float64)
        if len(source_points) != len(target_points):
            raise AssertionError("matched point sets must have the same length")
        n, dim = source_points.shape
        self.S_centroid = source_points.mean(axis=0)
        self.T_centroid = target_points.mean(axis=0)
        self.source_points_centered = source_points - self.S_centroid
        self.target_points_centered = target_points - self.T_centroid
        too_few = (dim == 2 and n < 2) or (dim == 3 and n < 3)       # reference :25
        if self.do_rotation and not too_few:
            cross_cov = self.source_points_centered.T @ self.target_points_centered
            U, _, Vt = np.linalg.svd(cross_cov)
            V = Vt.T
            R = V @ U.T
            if np.linalg.det(R) < 0:          # reflection: flip the weakest singular direction
                V[:, -1] = -V[:, -1]
                R = V @ U.T
            self.rotation_matrix = R
        else:
            self.rotation_matrix = np.eye(dim)
        if self.do_scale:
            rotated = self.source_points_centered @ self.rotation_matrix.T
            self.scale = np.sum(rotated * self.target_points_centered) / np.sum(rotated ** 2)
        if self.verbose:
            print("Rotation Matrix of the Affine Matrix:")
            print(self.rotation_matrix)
            print("Scaling factor:", self.scale)
        self.translation = self.T_centroid - self.S_centroid   # kept for API parity; predict does not use it
        return self

    def predict(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self.scale * ((x - self.S_centroid) @ self.rotation_matrix.T) + self.T_centroid

    def derivative(self, x):
        # the reference ignores `scale` here (affine_trasformation.py:55-57); preserved
        return np.broadcast_to(self.rotation_matrix, (np.shape(x)[0],) + self.rotation_matrix.shape).copy()
