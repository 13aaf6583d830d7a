"""Tiny numpy helpers for data generation and trajectory assembly (host side).

Once-per-sequence / once-per-frame 3x3 and 4x4 algebra; the per-frame hot path
lives in the HIP library.
"""
import numpy as np


def F_from_P(P1, P2):
    """F(i,j) = det[P1 without row j; P2 without row i] (cyclic row order),
    then F /= F(2,2) when F(2,2) > DBL_MIN — reference src/mvg.h:41-66 and
    src/viso.cpp:1176-1180."""
    P1 = np.asarray(P1, np.float64).reshape(3, 4)
    P2 = np.asarray(P2, np.float64).reshape(3, 4)
    pick = ((1, 2), (2, 0), (0, 1))
    F = np.empty((3, 3))
    for i in range(3):
        for j in range(3):
            M = np.vstack([P1[list(pick[j])], P2[list(pick[i])]])
            F[i, j] = np.linalg.det(M)
    if F[2, 2] > np.finfo(np.float64).tiny:
        F = F / F[2, 2]
    return F


def tr2mat(tr):
    """Reference src/viso.cpp:109-133."""
    rx, ry, rz, tx, ty, tz = tr
    sx, cx, sy, cy, sz, cz = np.sin(rx), np.cos(rx), np.sin(ry), np.cos(ry), np.sin(rz), np.cos(rz)
    return np.array([[+cy * cz, -cy * sz, +sy, tx],
                     [+sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy, ty],
                     [-cx * sy * cz + sx * sz, +cx * sy * sz + sx * cz, +cx * cy, tz],
                     [0, 0, 0, 1.0]])


def chain_poses(tr, ok, aliasing_quirk=False):
    """poses[0] = I; pose <- pose * inv(tr2mat(tr)) for every frame whose solve
    succeeded (reference src/viso.cpp:1189-1190, 1315-1321).  Frames with
    ok == 0 push nothing (:1287, :1323), so the list can be shorter than the
    frame count; `valid` says which frames contributed.

    aliasing_quirk=True reproduces what the reference's `Mat pose =
    poses.back(); pose = pose*tr_mat.inv();` appears to do with OpenCV's
    ref-counted Mat (the product is written into the buffer poses.back()
    shares, so the previous entry is overwritten before the clone is pushed).
    """
    poses = [np.eye(4)]
    valid = []
    for t in range(len(tr)):
        if not ok[t]:
            continue
        new = poses[-1] @ np.linalg.inv(tr2mat(tr[t]))
        if aliasing_quirk:
            poses[-1] = new
        poses.append(new.copy())
        valid.append(t)
    return poses, valid
