"""NumPy transcription of the GPU-only sampling kernels (TEST INFRASTRUCTURE ONLY).

Follows models/tf_ops/sampling/tf_sampling_g.cu:105-170 (farthestpointsamplingKernel)
and :172-181 (gatherpointKernel) of the reference.  BASELINE.json's north_star
designates this transcription (non-FMA, fp32, left-to-right) as the FPS oracle;
oracle/pointrcnn_oracle.c holds the same algorithm in C for speed and the two are
checked against each other in tests/test_oracle.py.

Kernel semantics reproduced:
  * idx[0] = 0 (:114-116); temp[k] = 1e38 (:117-119)
  * d = (x2-x1)^2 + (y2-y1)^2 + (z2-z1)^2 in fp32, left to right (:142)
  * temp[k] = min(d, temp[k]) (:143-145)
  * thread t of 512 scans k = t, t+512, ... keeping the FIRST strict maximum,
    starting from best=-1, besti=0 (:125-126,:130,:146-149)
  * 512-slot tree reduce that takes the right child only if strictly greater
    (:153-163)  =>  winner = max value, then smallest k mod 512, then smallest k.
"""
import numpy as np

BLOCK = 512


def farthest_point_sample(npoint, inp):
    inp = np.ascontiguousarray(inp, np.float32)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    if npoint <= 0:
        return out
    q = (n + BLOCK - 1) // BLOCK
    for i in range(b):
        pts = inp[i]
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        temp = np.full(n, np.float32(1e38), np.float32)
        old = 0
        for j in range(1, npoint):
            dx = x - x[old]
            dy = y - y[old]
            dz = z - z[old]
            d = (dx * dx + dy * dy) + dz * dz          # fp32, left to right, no FMA
            temp = np.minimum(d, temp)
            pad = np.full(q * BLOCK, np.float32(-1), np.float32)
            pad[:n] = temp
            grid = pad.reshape(q, BLOCK)               # grid[r, t] = temp[t + 512 r]
            r = np.argmax(grid, axis=0)                # first strict max per thread slot
            best = grid[r, np.arange(BLOCK)]
            besti = np.where(best > np.float32(-1), r * BLOCK + np.arange(BLOCK), 0)
            t = int(np.argmax(best))                   # tree reduce: leftmost slot wins ties
            old = int(besti[t])
            out[i, j] = old
    return out


def gather_point(inp, idx):
    inp = np.asarray(inp, np.float32)
    idx = np.asarray(idx)
    return np.stack([inp[i][idx[i]] for i in range(inp.shape[0])], 0)
