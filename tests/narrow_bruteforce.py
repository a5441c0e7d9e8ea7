"""A second opinion on cuboid / cuboid contact manifolds, written independently of oracle/avo_narrow.hpp and avian_amd/csrc/avn_narrow.h
(which share their structure): float64 numpy, no feature ids, no incremental clipping of vertex loops -- the contact patch is computed as the
INTERSECTION OF THE TWO FACE POLYGONS PROJECTED ALONG THE NORMAL (a generic convex-polygon intersection), an edge / edge contact as the
closest points of the two supporting edges.  It does not try to reproduce parry's choice among nearly equal axes: it takes the manifold
normal under test, checks that this axis is (within tolerance) a least-penetration axis of the SAT, and then predicts the point set."""
from __future__ import annotations

import itertools

import numpy as np


def rot_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def sat_separations(he1, p1, R1, he2, p2, R2):
    """[(separation, unit axis pointing from 1 to 2, kind)] over the 15 axes (degenerate edge crosses skipped)."""
    d = p2 - p1
    out = []
    axes = [(R1[:, i], "face1") for i in range(3)] + [(R2[:, i], "face2") for i in range(3)]
    for i, j in itertools.product(range(3), range(3)):
        c = np.cross(R1[:, i], R2[:, j])
        if np.linalg.norm(c) > 1e-6:
            axes.append((c / np.linalg.norm(c), "edge"))
    for a, kind in axes:
        if d @ a < 0:
            a = -a
        r1 = np.abs(R1.T @ a) @ he1
        r2 = np.abs(R2.T @ a) @ he2
        out.append((d @ a - r1 - r2, a, kind))
    return out


def face_polygon(he, p, R, n):
    """World-space corners (in order around the face) of the face of the box whose outward normal is most aligned with n."""
    loc = R.T @ n
    k = int(np.argmax(np.abs(loc)))
    s = np.sign(loc[k])
    u, v = [a for a in range(3) if a != k]
    corners = []
    for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
        c = np.zeros(3); c[k] = s * he[k]; c[u] = su * he[u]; c[v] = sv * he[v]
        corners.append(p + R @ c)
    return np.array(corners), R[:, k] * s


def clip_convex(subject, clip):
    """Sutherland-Hodgman: intersection of two convex 2-D polygons (any orientation)."""
    def ccw(poly):
        a = 0.5 * np.sum(poly[:, 0] * np.roll(poly[:, 1], -1) - np.roll(poly[:, 0], -1) * poly[:, 1])
        return poly if a > 0 else poly[::-1]
    out = list(ccw(np.asarray(subject, float)))
    cl = ccw(np.asarray(clip, float))
    for i in range(len(cl)):
        a, b = cl[i], cl[(i + 1) % len(cl)]
        e = b - a
        inside = lambda q: e[0] * (q[1] - a[1]) - e[1] * (q[0] - a[0]) >= -1e-12
        src, out = out, []
        for j in range(len(src)):
            c, dd = src[j], src[(j + 1) % len(src)]
            ic, idd = inside(c), inside(dd)
            if ic:
                out.append(c)
            if ic != idd:
                den = e[0] * (dd[1] - c[1]) - e[1] * (dd[0] - c[0])
                t = (e[0] * (a[1] - c[1]) - e[1] * (a[0] - c[0])) / den
                out.append(c + t * (dd - c))
        if not out:
            return np.zeros((0, 2))
    return np.array(out)


def dedupe(points, tol):
    keep = []
    for q in points:
        if not any(np.linalg.norm(q - k) < tol for k in keep):
            keep.append(q)
    return np.array(keep).reshape(-1, points.shape[1]) if len(keep) else points[:0]


def predict(he1, p1, q1, he2, p2, q2, normal, prediction, tol=1e-6):
    """('face' | 'edge' | 'vertex', best separation over all axes, separation along `normal`, predicted world contact points, penetrations)."""
    he1, p1, he2, p2, n = (np.asarray(a, float) for a in (he1, p1, he2, p2, normal))
    R1, R2 = rot_matrix(np.asarray(q1, float)), rot_matrix(np.asarray(q2, float))
    seps = sat_separations(he1, p1, R1, he2, p2, R2)
    best = max(s for s, _, _ in seps)
    r1 = np.abs(R1.T @ n) @ he1; r2 = np.abs(R2.T @ n) @ he2
    sep_n = (p2 - p1) @ n - r1 - r2
    # parry builds the manifold from the two SUPPORT FACES along the separating normal, whatever kind of axis the normal is (for an
    # edge x edge axis the faces are the ones most aligned with it): one formulation covers face and edge axes
    if True:
        fa, na = face_polygon(he1, p1, R1, n)      # face of box 1 looking at box 2
        fb, nb = face_polygon(he2, p2, R2, -n)     # face of box 2 looking at box 1
        # both faces must really be faces of contact: the other box's face is used even when it is tilted (its corners then differ in depth)
        t1 = np.cross(n, [1.0, 0, 0]) if abs(n[0]) < 0.9 else np.cross(n, [0, 1.0, 0])
        t1 /= np.linalg.norm(t1); t2 = np.cross(n, t1)
        pa = np.stack([fa @ t1, fa @ t2], 1); pb = np.stack([fb @ t1, fb @ t2], 1)
        poly = dedupe(clip_convex(pa, pb), 1e-9)
        pts, pens = [], []
        for u in poly:
            base = u[0] * t1 + u[1] * t2
            # lift along n onto each face plane: (base + h n - c) . nf = 0
            h1 = ((fa[0] - base) @ na) / (n @ na)
            h2 = ((fb[0] - base) @ nb) / (n @ nb)
            dist = h2 - h1
            # (no per-point cut at the prediction distance: parry reports every vertex of the clipped patch -- the prediction distance gates
            #  the manifold as a whole -- and Avian prunes single points itself, narrow_phase/system_param.rs:731-757)
            pts.append(base + 0.5 * (h1 + h2) * n); pens.append(-dist)
        kind = "face" if max(np.max(np.abs(R1.T @ n)), np.max(np.abs(R2.T @ n))) > 1 - 1e-6 else "edge"
        return kind, best, sep_n, np.array(pts).reshape(-1, 3), np.array(pens)
