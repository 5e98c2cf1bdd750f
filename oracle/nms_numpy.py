"""Second, independent rotated-rectangle intersection (TEST INFRASTRUCTURE ONLY).

oracle/pointrcnn_oracle.c computes the intersection area by Sutherland-Hodgman
clipping.  This module computes the same quantity the way the reference's OpenCV
calls do (utils/nms_rotate.py:64-67: rotatedRectangleIntersection -> convexHull
-> contourArea): collect (a) edge/edge intersection points, (b) vertices of one
rectangle inside the other, order them by angle around their centroid, shoelace.
It exists so the oracle is not self-referential; float64 throughout.

Also a pure-Python restatement of nms_rotate_cpu's control flow
(utils/nms_rotate.py:39-73) built on this intersection, for small cases.
"""
import math

import numpy as np


def rect_corners(box):
    """OpenCV RotatedRect((b1,b0),(b3,b2),b4).points() (public definition)."""
    v, u, H, W, deg = [float(t) for t in box]   # box = [b0,b1,b2,b3,b4]; centre=(b1,b0), size=(b3,b2)
    ang = deg * math.pi / 180.0
    a = math.sin(ang) * 0.5
    c = math.cos(ang) * 0.5
    p0 = (u - a * H - c * W, v + c * H - a * W)
    p1 = (u + a * H - c * W, v - c * H - a * W)
    p2 = (2 * u - p0[0], 2 * v - p0[1])
    p3 = (2 * u - p1[0], 2 * v - p1[1])
    return np.array([p0, p1, p2, p3], np.float64)


def _inside(pt, poly):
    """pt inside-or-on convex quad poly (either orientation)."""
    s = []
    for i in range(4):
        a, b = poly[i], poly[(i + 1) % 4]
        s.append((b[0] - a[0]) * (pt[1] - a[1]) - (b[1] - a[1]) * (pt[0] - a[0]))
    s = np.array(s)
    return bool(np.all(s >= -1e-12) or np.all(s <= 1e-12))


def _seg_intersect(p, p2, q, q2):
    r = p2 - p
    s = q2 - q
    den = r[0] * s[1] - r[1] * s[0]
    if abs(den) < 1e-18:
        return None
    qp = q - p
    t = (qp[0] * s[1] - qp[1] * s[0]) / den
    u = (qp[0] * r[1] - qp[1] * r[0]) / den
    if -1e-12 <= t <= 1 + 1e-12 and -1e-12 <= u <= 1 + 1e-12:
        return p + t * r
    return None


def rotated_intersection(b1, b2):
    p1, p2 = rect_corners(b1), rect_corners(b2)
    pts = []
    for i in range(4):
        for j in range(4):
            x = _seg_intersect(p1[i], p1[(i + 1) % 4], p2[j], p2[(j + 1) % 4])
            if x is not None:
                pts.append(x)
    for v in p1:
        if _inside(v, p2):
            pts.append(v)
    for v in p2:
        if _inside(v, p1):
            pts.append(v)
    if len(pts) < 3:
        return 0.0
    pts = np.array(pts)
    c = pts.mean(0)
    ang = np.arctan2(pts[:, 1] - c[1], pts[:, 0] - c[0])
    pts = pts[np.argsort(ang)]
    x, y = pts[:, 0], pts[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def rotated_iou(b1, b2):
    ia = rotated_intersection(b1, b2)
    return ia * 1.0 / (b1[2] * b1[3] + b2[2] * b2[3] - ia + 0.0000001)


def nms_rotate_cpu(boxes, scores, iou_threshold, max_output_size):
    """utils/nms_rotate.py:39-73, line by line, on the intersection above."""
    boxes = np.asarray(boxes, np.float64)
    keep = []
    order = np.asarray(scores).argsort(kind='stable')[::-1]
    num = boxes.shape[0]
    suppressed = np.zeros((num), dtype=np.int64)
    for _i in range(num):
        if len(keep) >= max_output_size:
            break
        i = order[_i]
        if suppressed[i] == 1:
            continue
        keep.append(i)
        for _j in range(_i + 1, num):
            j = order[_j]
            inter = rotated_iou(boxes[i], boxes[j])
            if inter >= iou_threshold:
                suppressed[j] = 1
    return np.array(keep, np.int64)
