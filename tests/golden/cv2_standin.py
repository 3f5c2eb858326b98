"""tests/golden/cv2_standin.py -- used ONLY by tests/golden/make_golden.py, in the build container.

OpenCV (`cv2`) is not installed.  To let the reference's own Python control flow run far
enough to produce golden vectors, make_golden.py registers this module as `sys.modules['cv2']`.  Every
function here is an independent numpy/scipy restatement of the OpenCV primitive per SURVEY Appendix B
(it shares no code with oracle/ or with the product), so goldens that pass through one of these are
labelled "control flow pinned by the reference, cv2 primitive unpinned" in tests/golden/README.md.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi

# constants the reference touches
COLOR_BGR2LAB = 44; COLOR_Lab2BGR = 56; COLOR_GRAY2BGR = 8
IMREAD_COLOR = 1; IMREAD_GRAYSCALE = 0
MORPH_RECT = 0; MORPH_CROSS = 1; MORPH_ELLIPSE = 2; MORPH_OPEN = 2; MORPH_CLOSE = 3
TERM_CRITERIA_EPS = 2; TERM_CRITERIA_MAX_ITER = 1; KMEANS_PP_CENTERS = 2
LINE_8 = 8; CV_8U = 0; BORDER_CONSTANT = 0


def imread(path, flags=IMREAD_COLOR):
    from PIL import Image
    try:
        im = Image.open(path)
    except Exception:
        return None
    if flags == IMREAD_GRAYSCALE:
        return np.array(im.convert("L"))
    a = np.array(im.convert("RGB"))
    return a[:, :, ::-1].copy()


def imwrite(path, img):
    from PIL import Image
    a = np.asarray(img)
    if a.ndim == 3:
        a = a[:, :, ::-1]
    Image.fromarray(a).save(path)
    return True


def connectedComponents(img, connectivity=8):
    fg = np.asarray(img) > 0
    lab, n = ndi.label(fg, structure=np.ones((3, 3), int))
    if n == 0:
        return 1, lab.astype(np.int32)
    h, w = fg.shape
    wb = (w + 1) // 2
    ys, xs = np.nonzero(fg)
    key = (ys // 2) * wb + (xs // 2)
    first = np.full(n + 1, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first, lab[ys, xs], key)
    order = np.argsort(first[1:], kind="stable")
    remap = np.zeros(n + 1, np.int32)
    remap[order + 1] = np.arange(1, n + 1)
    return n + 1, remap[lab].astype(np.int32)


def filter2D(src, ddepth, kernel, borderType=BORDER_CONSTANT):
    out = ndi.correlate(np.asarray(src).astype(np.int32), np.asarray(kernel).astype(np.int32), mode="constant", cval=0)
    return np.clip(out, 0, 255).astype(np.uint8)


def arcLength(curve, closed):
    p = np.asarray(curve).reshape(-1, 2).astype(np.float32)
    if len(p) <= 1:
        return 0.0
    prev = p[-1] if closed else p[0]
    per = 0.0
    for q in p:
        dx = np.float32(q[0] - prev[0]); dy = np.float32(q[1] - prev[1])
        per += float(np.sqrt(np.float32(np.float32(dx * dx) + np.float32(dy * dy))))
        prev = q
    return per


def _capsule(img, a, b, r, val):
    h, w = img.shape[:2]
    ax, ay = int(a[0]), int(a[1]); bx, by = int(b[0]), int(b[1])
    x0, x1 = max(0, min(ax, bx) - r), min(w - 1, max(ax, bx) + r)
    y0, y1 = max(0, min(ay, by) - r), min(h - 1, max(ay, by) + r)
    if x0 > x1 or y0 > y1:
        return
    ys, xs = np.mgrid[y0:y1 + 1, x0:x1 + 1].astype(np.int64)
    qx, qy = xs - ax, ys - ay
    dx, dy = bx - ax, by - ay
    L2 = dx * dx + dy * dy
    q2 = qx * qx + qy * qy
    r2 = r * r
    if L2 == 0:
        inside = q2 <= r2
    else:
        t = qx * dx + qy * dy
        e2 = (xs - bx) ** 2 + (ys - by) ** 2
        inside = np.where(t <= 0, q2 <= r2, np.where(t >= L2, e2 <= r2, q2 * L2 - t * t <= r2 * L2))
    img[y0:y1 + 1, x0:x1 + 1][inside] = val


def line(img, p0, p1, color, thickness=1, lineType=LINE_8):
    _capsule(img, p0, p1, int(thickness) // 2, color)
    return img


def polylines(img, pts, isClosed, color, thickness=1, lineType=LINE_8):
    for arr in pts:
        p = np.asarray(arr).reshape(-1, 2)
        for i in range(len(p) - 1):
            _capsule(img, p[i], p[i + 1], int(thickness) // 2, color)
    return img


def circle(img, center, radius, color, thickness=-1, lineType=LINE_8):
    assert thickness < 0
    _capsule(img, center, center, int(radius), color)
    return img


# ---- minEnclosingCircle (float path), recalled from OpenCV 4.x shapedescr.cpp ----
_F = np.float32
_EPS = _F(1.0e-4)


def _nrm(dx, dy):
    return float(np.sqrt(float(dx) * float(dx) + float(dy) * float(dy)))


def _circle3(p0, p1, p2):
    v1 = (_F(p1[0] - p0[0]), _F(p1[1] - p0[1])); v2 = (_F(p2[0] - p0[0]), _F(p2[1] - p0[1]))
    m1 = (_F(_F(p0[0] + p1[0]) / _F(2)), _F(_F(p0[1] + p1[1]) / _F(2)))
    c1 = _F(_F(m1[0] * v1[0]) + _F(m1[1] * v1[1]))
    m2 = (_F(_F(p0[0] + p2[0]) / _F(2)), _F(_F(p0[1] + p2[1]) / _F(2)))
    c2 = _F(_F(m2[0] * v2[0]) + _F(m2[1] * v2[1]))
    det = _F(_F(v1[0] * v2[1]) - _F(v1[1] * v2[0]))
    if abs(det) <= _EPS:
        def sq(a, b):
            dx = _F(a[0] - b[0]); dy = _F(a[1] - b[1]); return _F(_F(dx * dx) + _F(dy * dy))
        d1, d2, d3 = sq(p0, p1), sq(p0, p2), sq(p1, p2)
        r = _F(_F(np.sqrt(max(d1, max(d2, d3))) * _F(0.5)) + _EPS)
        if d1 >= d2 and d1 >= d3: c = (_F(_F(p0[0] + p1[0]) * _F(.5)), _F(_F(p0[1] + p1[1]) * _F(.5)))
        elif d2 >= d1 and d2 >= d3: c = (_F(_F(p0[0] + p2[0]) * _F(.5)), _F(_F(p0[1] + p2[1]) * _F(.5)))
        else: c = (_F(_F(p1[0] + p2[0]) * _F(.5)), _F(_F(p1[1] + p2[1]) * _F(.5)))
        return c, r
    cx = _F(_F(_F(c1 * v2[1]) - _F(c2 * v1[1])) / det); cy = _F(_F(_F(v1[0] * c2) - _F(v2[0] * c1)) / det)
    c = (cx, cy)
    ex = _F(cx - p0[0]); ey = _F(cy - p0[1])
    r = _F(np.sqrt(_F(_F(ex * ex) + _F(ey * ey))) + _EPS)
    return c, r


def _third(pts, i, j):
    c = (_F(_F(pts[j][0] + pts[i][0]) / _F(2)), _F(_F(pts[j][1] + pts[i][1]) / _F(2)))
    r = _F(_F(_F(_nrm(_F(pts[j][0] - pts[i][0]), _F(pts[j][1] - pts[i][1]))) / _F(2)) + _EPS)
    for k in range(j):
        if _nrm(_F(c[0] - pts[k][0]), _F(c[1] - pts[k][1])) < float(r):
            continue
        nc, nr = _circle3(pts[i], pts[j], pts[k])
        if nr > 0: c, r = nc, nr
    return c, r


def _second(pts, i):
    c = (_F(_F(pts[0][0] + pts[i][0]) / _F(2)), _F(_F(pts[0][1] + pts[i][1]) / _F(2)))
    r = _F(_F(_F(_nrm(_F(pts[0][0] - pts[i][0]), _F(pts[0][1] - pts[i][1]))) / _F(2)) + _EPS)
    for j in range(1, i):
        if _nrm(_F(c[0] - pts[j][0]), _F(c[1] - pts[j][1])) < float(r):
            continue
        nc, nr = _third(pts, i, j)
        if nr > 0: c, r = nc, nr
    return c, r


def minEnclosingCircle(points):
    pts = [(_F(x), _F(y)) for x, y in np.asarray(points).reshape(-1, 2)]
    n = len(pts)
    if n == 0:
        return (0.0, 0.0), 0.0
    if n == 1:
        return (float(pts[0][0]), float(pts[0][1])), float(_EPS)
    if n == 2:
        c = (_F(_F(pts[0][0] + pts[1][0]) / _F(2)), _F(_F(pts[0][1] + pts[1][1]) / _F(2)))
        r = _F(_F(_nrm(_F(pts[0][0] - pts[1][0]), _F(pts[0][1] - pts[1][1])) / 2.0) + _EPS)
        return (float(c[0]), float(c[1])), float(r)
    c = (_F(_F(pts[0][0] + pts[1][0]) / _F(2)), _F(_F(pts[0][1] + pts[1][1]) / _F(2)))
    r = _F(_F(_F(_nrm(_F(pts[0][0] - pts[1][0]), _F(pts[0][1] - pts[1][1]))) / _F(2)) + _EPS)
    for i in range(2, n):
        d = _F(_nrm(_F(pts[i][0] - c[0]), _F(pts[i][1] - c[1])))
        if d < r:
            continue
        nc, nr = _second(pts, i)
        if nr > 0: c, r = nc, nr
    return (float(c[0]), float(c[1])), float(r)


def __getattr__(name):  # anything else the reference might touch is an explicit error
    raise AttributeError(f"cv2 stand-in: '{name}' is not provided (OpenCV is not installed; SURVEY 8c)")
