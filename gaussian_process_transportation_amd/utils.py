"""Host helpers used by the 2-D examples of the reference (policy_transportation/utils.py:7-45)."""
import numpy as np


def resample(surface, num_points=20):
    """Arc-length resampling of a 2-D polyline to `num_points` equally spaced samples."""
    pts = np.asarray(surface, dtype=np.float64)
    seg = np.sqrt(np.diff(pts[:, 0]) ** 2 + np.diff(pts[:, 1]) ** 2)
    spacing = np.sum(seg) / (num_points - 1)
    out = [pts[0]]
    here = pts[0]
    left = spacing
    for nxt in pts[1:]:
        step = np.sqrt((nxt[0] - here[0]) ** 2 + (nxt[1] - here[1]) ** 2)
        if left <= step:
            here = here + (left / step) * (nxt - here)
            out.append(here)
            left = spacing
        else:
            here = nxt
            left -= step
    while len(out) < num_points:
        out.append(pts[-1])
    return np.array(out)
