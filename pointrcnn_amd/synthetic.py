"""Seeded synthetic inputs of the shapes the reference's data layer produces.

There is no KITTI data in the build or on the GPU box, so every test and bench
runs on these generators.  NumPy only; host side.

  kitti_scene     mirrors dataset/rpn_dataset.py:205-213 of the reference: camera-rect
                  coordinates after the FOV crop, sampled WITH replacement to 16384
                  points, so exact duplicate points exist (exercises FPS tie-breaks).
  frustum_batch   mirrors dataset/frustum_dataset.py:308-314: <=600 points inside an
                  expanded proposal box, re-sampled with replacement to 1024 / 512.
  dense_scene     131072 distinct points (stand-in for dataset/dense_pointcloud.py:34-52).
  nms_boxes       clustered rotated BEV boxes [cx, cz, l, w, deg] with tie-free scores
                  (dataset/frustum_dataset.py:126-128 builds the same 5-vector).
"""
import numpy as np


def _base_points(rng, P):
    # z ~ U(2,70) weighted ~1/z: inverse-CDF of 1/z on [2,70]
    u = rng.random(P)
    z = 2.0 * (70.0 / 2.0) ** u
    phi = np.deg2rad(rng.uniform(-40.0, 40.0, P))
    x = z * np.tan(phi)
    y = 1.65 + rng.normal(0.0, 0.05, P)
    nobj = int(0.15 * P)
    centers = np.stack([rng.uniform(-20, 20, 60), rng.uniform(5, 60, 60)], 1)
    which = rng.integers(0, 60, nobj)
    ox = centers[which, 0] + rng.uniform(-2.0, 2.0, nobj)
    oz = centers[which, 1] + rng.uniform(-1.0, 1.0, nobj)
    oy = rng.uniform(-1.5, 1.6, nobj)
    sel = rng.choice(P, nobj, replace=False)
    x[sel], y[sel], z[sel] = ox, oy, oz
    inten = rng.random(P)
    return np.stack([x, y, z, inten], 1).astype(np.float32)


def kitti_scene(seed, n=16384, P=19000):
    """(n,4) float32: xyz + intensity, sampled with replacement (duplicates exist)."""
    rng = np.random.default_rng(seed)
    base = _base_points(rng, P)
    choice = rng.choice(P, n, replace=True)
    return base[choice]


def kitti_batch(first_scene, b, n=16384):
    """(b,n,4) float32; scene i uses seed 1000+i."""
    return np.stack([kitti_scene(1000 + first_scene + i, n) for i in range(b)], 0)


def dense_scene(seed=5000, n=131072):
    rng = np.random.default_rng(seed)
    return _base_points(rng, n)


def frustum_batch(seed, b=32, n=1024, channels=4):
    rng = np.random.default_rng(seed)
    out = np.empty((b, n, channels), np.float32)
    for i in range(b):
        k = int(rng.integers(40, 600))
        l, w, h = 3.88 + 0.5, 1.63 + 0.5, 1.56 + 0.5
        pts = np.stack([rng.uniform(-l / 2, l / 2, k), rng.uniform(-h / 2, h / 2, k),
                        rng.uniform(-w / 2, w / 2, k)], 1)
        ang = rng.uniform(-np.pi, np.pi)
        c, s = np.cos(ang), np.sin(ang)
        pts = pts @ np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]).T
        feat = rng.random((k, channels - 3))
        full = np.concatenate([pts, feat], 1).astype(np.float32)
        out[i] = full[rng.choice(k, n, replace=True)]
    return out


_CLASS_LW = np.array([[3.88, 1.63], [0.84, 0.66], [1.76, 0.60]])  # utils/box_encoder.py:6-8


def nms_boxes(seed=9000, N=9000, clusters=150):
    """boxes (N,5) float64 [cx, cz, l, w, deg], scores (N,) float64 without ties."""
    rng = np.random.default_rng(seed)
    per = (N + clusters - 1) // clusters
    cx = rng.uniform(-40, 40, clusters)
    cz = rng.uniform(2, 70, clusters)
    head = rng.uniform(-np.pi, np.pi, clusters)
    cls = rng.integers(0, 3, clusters)
    cid = np.repeat(np.arange(clusters), per)[:N]
    x = cx[cid] + rng.normal(0, 0.3, N)
    z = cz[cid] + rng.normal(0, 0.3, N)
    lw = _CLASS_LW[cls[cid]] * (1.0 + rng.normal(0, 0.1, (N, 2)))
    lw = np.maximum(lw, 0.05)
    ry = head[cid] + rng.normal(0, 0.15, N)
    boxes = np.stack([x, z, lw[:, 0], lw[:, 1], 180.0 * ry / np.pi], 1)
    scores = rng.permutation(np.linspace(0.0, 1.0, N))
    return boxes, scores


def xavier_mlp(rng, cin, widths):
    """Per-layer (W (cin,cout), scale, shift) with BN folded: gamma=1, beta=0, mean=0,
    var=1, eps=1e-3 (models/tf_util.py:527 default eps) and zero conv bias."""
    layers = []
    for cout in widths:
        lim = np.sqrt(6.0 / (cin + cout))
        W = rng.uniform(-lim, lim, (cin, cout)).astype(np.float32)
        scale = np.full(cout, 1.0 / np.sqrt(1.0 + 1e-3), np.float32)
        shift = np.zeros(cout, np.float32)
        layers.append((W, scale, shift))
        cin = cout
    return layers
