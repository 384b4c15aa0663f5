"""Random box sets + an independent float64 rotated-rectangle intersection (Sutherland-Hodgman)."""
import numpy as np


def random_boxes(seed, n, spread=20.0, clustered=True):
    rng = np.random.default_rng(seed)
    if clustered:  # proposals cluster around a few objects, like detector output
        centers = rng.uniform(-spread, spread, size=(max(n // 12, 1), 2))
        xy = centers[rng.integers(0, centers.shape[0], n)] + rng.normal(0, 0.6, size=(n, 2))
    else:
        xy = rng.uniform(-spread, spread, size=(n, 2))
    z = rng.uniform(-1.5, 0.5, size=(n, 1))
    dims = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.7, 1.3, size=(n, 3))
    yaw = rng.uniform(-np.pi, np.pi, size=(n, 1))
    return np.concatenate([xy, z, dims, yaw], 1).astype(np.float32)


def _corners(b):
    x, y, dx, dy, a = float(b[0]), float(b[1]), float(b[3]), float(b[4]), float(b[6])
    c, s = np.cos(a), np.sin(a)
    pts = np.array([[-dx / 2, -dy / 2], [dx / 2, -dy / 2], [dx / 2, dy / 2], [-dx / 2, dy / 2]])
    rot = np.array([[c, -s], [s, c]])
    return pts @ rot.T + np.array([x, y])


def _clip(subject, a, b):
    out = []
    n = len(subject)
    def inside(p):
        return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0
    def inter(p, q):
        d1, d2 = b - a, q - p
        den = d1[0] * d2[1] - d1[1] * d2[0]
        t = ((p[0] - a[0]) * d2[1] - (p[1] - a[1]) * d2[0]) / den
        return a + t * d1
    for i in range(n):
        cur, prev = subject[i], subject[i - 1]
        if inside(cur):
            if not inside(prev):
                out.append(inter(prev, cur))
            out.append(cur)
        elif inside(prev):
            out.append(inter(prev, cur))
    return out


def exact_overlap(box_a, box_b):
    """float64 intersection area of two rotated rectangles."""
    poly = [p for p in _corners(box_a)]
    clip = _corners(box_b)
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, clip[i], clip[(i + 1) % 4])
    if len(poly) < 3:
        return 0.0
    p = np.array(poly)
    return 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))
