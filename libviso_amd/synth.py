"""Seeded synthetic stereo+temporal scenes.

There is no KITTI data in the build environment, so bench.py and the parity
tests run on synthetic frames with the structure the reference's front-end
produces: integer-pixel keypoints (HarrisBinnedFeatureDetector emits
Point2f(int,int), reference src/viso.cpp:967) and 121-element integer-valued
float32 descriptors in [-1020, 1020] (3x3 Sobel-x of uint8 over an 11x11
window, src/viso.cpp:1004-1024).  "Descriptor-only" mode: every world point
carries a base descriptor; each view adds small integer noise, so
corresponding windows are similar but not identical; outlier keypoints carry
unrelated descriptors; `dup_frac` re-uses descriptors to force exact SAD ties.

Layout matches viso_batch (include/viso_hip.h):
    kp   [n_frames][2][cap][2]   float32
    desc [n_frames][2][cap][121] float32
    n    [n_frames][2]           int32
"""
import numpy as np

from .abi import DESC_LEN, Param

KITTI_F, KITTI_CU, KITTI_CV = 718.856, 607.1928, 185.2157
KITTI_BASE = 386.1448 / 718.856  # |P2(0,3)/P2(0,0)|, reference src/viso.cpp:1184

KITTI_P1 = np.array([[KITTI_F, 0, KITTI_CU, 0], [0, KITTI_F, KITTI_CV, 0], [0, 0, 1, 0]], np.float64)
KITTI_P2 = np.array([[KITTI_F, 0, KITTI_CU, -386.1448], [0, KITTI_F, KITTI_CV, 0], [0, 0, 1, 0]], np.float64)


def rot_from_tr(tr):
    """R,t of tr2mat (reference src/viso.cpp:109-133)."""
    rx, ry, rz, tx, ty, tz = tr
    sx, cx, sy, cy, sz, cz = np.sin(rx), np.cos(rx), np.sin(ry), np.cos(ry), np.sin(rz), np.cos(rz)
    R = np.array([[+cy * cz, -cy * sz, +sy],
                  [+sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy],
                  [-cx * sy * cz + sx * sz, +cx * sy * sz + sx * cz, +cx * cy]])
    return R, np.array([tx, ty, tz])


def _new_points(rng, k, width, height, zmin, zmax, f, cu, cv, blobs=None, cluster_frac=0.0):
    u = rng.uniform(0, width - 1, k)
    v = rng.uniform(0, height - 1, k)
    if blobs is not None and cluster_frac > 0 and k > 0:
        # clustered features (what a corner detector finds on real images: texture comes in patches): a share of the
        # points is drawn around a few blob centres instead of uniformly
        pick = rng.random(k) < cluster_frac
        c = blobs[rng.integers(0, len(blobs), k)]
        uc = np.clip(c[:, 0] + rng.normal(0, c[:, 2], k), 0, width - 1)
        vc = np.clip(c[:, 1] + rng.normal(0, c[:, 2], k), 0, height - 1)
        u = np.where(pick, uc, u)
        v = np.where(pick, vc, v)
    z = rng.uniform(zmin, zmax, k)
    P = np.stack([(u - cu) * z / f, (v - cv) * z / f, z], 1)
    return P


def _new_desc(rng, k):
    # Sobel-like: mostly small gradients, occasional strong edges.
    d = rng.normal(0.0, 90.0, (k, DESC_LEN)) * (1.0 + 3.0 * (rng.random((k, 1)) < 0.15))
    return np.clip(np.rint(d), -1020, 1020).astype(np.int16)


def make_sequence(seed, n_frames, n_kp=2000, width=1241, height=376, outlier_frac=0.2,
                  noise_sigma=6.0, dup_frac=0.0, cap=None, ragged=False,
                  zmin=4.0, zmax=60.0, cluster_frac=0.0, n_blobs=14):
    """Returns dict(kp, desc, n, tr_gt, param, F, P1, P2, width, height).

    tr_gt[t] maps 3-D points of frame t-1 (left camera) into frame t, the
    convention of compute_J (reference src/viso.cpp:1441-1443)."""
    rng = np.random.default_rng(seed)
    cap = cap or n_kp
    f, cu, cv, base = KITTI_F, KITTI_CU, KITTI_CV, KITTI_BASE
    n_in = int(round(n_kp * (1.0 - outlier_frac)))
    kp = np.zeros((n_frames, 2, cap, 2), np.float32)
    desc = np.zeros((n_frames, 2, cap, DESC_LEN), np.float32)
    n = np.zeros((n_frames, 2), np.int32)
    tr_gt = np.zeros((n_frames, 6))
    blobs = None
    if cluster_frac > 0:   # blob centres (u, v) and radii (sigma, px) in the image
        blobs = np.stack([rng.uniform(0, width - 1, n_blobs), rng.uniform(0, height - 1, n_blobs),
                          rng.uniform(12, 45, n_blobs)], 1)
    P = _new_points(rng, n_in, width, height, zmin, zmax, f, cu, cv, blobs, cluster_frac)
    D = _new_desc(rng, n_in)
    for t in range(n_frames):
        if t > 0:
            tr = np.concatenate([rng.uniform(-0.02, 0.02, 3), rng.uniform(-0.05, 0.05, 2),
                                 -rng.uniform(0.5, 1.5, 1)])
            tr_gt[t] = tr
            R, tt = rot_from_tr(tr)
            P = P @ R.T + tt
            u = f * P[:, 0] / P[:, 2] + cu
            v = f * P[:, 1] / P[:, 2] + cv
            keep = (P[:, 2] > 2.0) & (u >= 0) & (u <= width - 1) & (v >= 0) & (v <= height - 1)
            P, D = P[keep], D[keep]
            k_new = n_in - len(P)
            if k_new > 0:
                P = np.concatenate([P, _new_points(rng, k_new, width, height, zmin, zmax, f, cu, cv, blobs, cluster_frac)])
                D = np.concatenate([D, _new_desc(rng, k_new)])
        uL = np.rint(f * P[:, 0] / P[:, 2] + cu)
        vL = np.rint(f * P[:, 1] / P[:, 2] + cv)
        uR = np.rint(f * (P[:, 0] - base) / P[:, 2] + cu)
        for side in (0, 1):
            n_img = n_kp if not ragged else int(n_kp - rng.integers(0, max(1, n_kp // 5)))
            n_img = min(n_img, cap)
            if side == 0:
                vis = np.ones(len(P), bool)
                uu = uL
            else:
                vis = (uR >= 0) & (uR <= width - 1)
                uu = uR
            idx = np.nonzero(vis)[0][:n_img]
            k_real = len(idx)
            k_out = n_img - k_real
            xy = np.empty((n_img, 2), np.float32)
            xy[:k_real, 0] = uu[idx]
            xy[:k_real, 1] = vL[idx]
            xy[k_real:, 0] = rng.integers(0, width, k_out)
            xy[k_real:, 1] = rng.integers(0, height, k_out)
            dd = np.empty((n_img, DESC_LEN), np.int16)
            noise = np.rint(rng.normal(0.0, noise_sigma, (k_real, DESC_LEN))).astype(np.int16)
            dd[:k_real] = np.clip(D[idx] + noise, -1020, 1020)
            dd[k_real:] = _new_desc(rng, k_out)
            if dup_frac > 0 and k_real > 0 and k_out > 0:
                k_dup = min(k_out, int(dup_frac * n_img))
                src = rng.integers(0, k_real, k_dup)
                dd[k_real:k_real + k_dup] = dd[src]          # exact duplicate patches
                xy[k_real:k_real + k_dup] = xy[src] + rng.integers(-3, 4, (k_dup, 2))
                xy[:, 0] = np.clip(xy[:, 0], 0, width - 1)
                xy[:, 1] = np.clip(xy[:, 1], 0, height - 1)
            perm = rng.permutation(n_img)
            kp[t, side, :n_img] = xy[perm]
            desc[t, side, :n_img] = dd[perm].astype(np.float32)
            n[t, side] = n_img
    from . import hostmath
    F = hostmath.F_from_P(KITTI_P1, KITTI_P2)
    param = Param.default(base=base, f=f, cu=cu, cv=cv)
    return dict(kp=kp, desc=desc, n=n, tr_gt=tr_gt, param=param, F=F, P1=KITTI_P1, P2=KITTI_P2,
                width=width, height=height)


def make_images(seed, rows=96, cols=128):
    """Small band-limited uint8 texture for descriptor-extractor tests."""
    rng = np.random.default_rng(seed)
    img = rng.normal(0, 1, (rows, cols))
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    for ax in (0, 1):
        img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), ax, img)
    img = (img - img.min()) / (img.max() - img.min())
    return np.rint(img * 255).astype(np.uint8)


def make_solver_case(seed, m=200, outlier_frac=0.25, tr=None, noise=0.3):
    """3-D points of the previous frame + their (noisy) stereo observations in
    the current frame: inputs of ransac_minimize_reproj (reference
    src/viso.cpp:1543).  Recipe of the reference's disabled test
    test/test.cpp:51-114 with a correct baseline."""
    rng = np.random.default_rng(seed)
    f, cu, cv, base = KITTI_F, KITTI_CU, KITTI_CV, KITTI_BASE
    if tr is None:
        tr = np.concatenate([rng.uniform(-0.02, 0.02, 3), rng.uniform(-0.05, 0.05, 2),
                             -rng.uniform(0.5, 1.5, 1)])
    tr = np.asarray(tr, np.float64)
    Xp = _new_points(rng, m, 1241, 376, 5.0, 50.0, f, cu, cv)
    R, t = rot_from_tr(tr)
    Xc = Xp @ R.T + t
    obs = np.stack([f * Xc[:, 0] / Xc[:, 2] + cu, f * Xc[:, 1] / Xc[:, 2] + cv,
                    f * (Xc[:, 0] - base) / Xc[:, 2] + cu, f * Xc[:, 1] / Xc[:, 2] + cv], 0)
    obs = obs + rng.normal(0, noise, obs.shape)
    k_out = int(outlier_frac * m)
    if k_out:
        bad = rng.choice(m, k_out, replace=False)
        obs[:, bad] += rng.uniform(-40, 40, (4, k_out))
    param = Param.default(base=base, f=f, cu=cu, cv=cv)
    return np.ascontiguousarray(Xp.T), np.ascontiguousarray(obs), tr, param


def make_image_sequence(seed, n_frames, n_kp=2000, width=1241, height=376, outlier_frac=0.2,
                        noise_sigma=2.0, cap=None, zmin=4.0, zmax=60.0):
    """Image-in variant of make_sequence for the device-side descriptor
    extractor (reference MyFeatureExtractor, src/viso.cpp:981-1025): returns
    uint8 images [n_frames][2][height][width] plus keypoints.  Every world point
    owns a 13x13 texture patch that is painted (with a little per-view noise)
    around its projection in every view, so the 11x11 Sobel windows of
    corresponding keypoints are similar and the rest of the pipeline sees real
    matches; outlier keypoints sit on background noise."""
    rng = np.random.default_rng(seed)
    cap = cap or n_kp
    f, cu, cv, base = KITTI_F, KITTI_CU, KITTI_CV, KITTI_BASE
    n_in = int(round(n_kp * (1.0 - outlier_frac)))
    R = 6                                                   # patch radius (13x13 covers the 11x11 window + Sobel)
    images = np.zeros((n_frames, 2, height, width), np.uint8)
    kp = np.zeros((n_frames, 2, cap, 2), np.float32)
    n = np.zeros((n_frames, 2), np.int32)
    tr_gt = np.zeros((n_frames, 6))

    def new_patches(k):
        p = rng.normal(128.0, 45.0, (k, 2 * R + 1, 2 * R + 1))
        return np.clip(p, 0, 255)

    P = _new_points(rng, n_in, width, height, zmin, zmax, f, cu, cv)
    T = new_patches(n_in)
    for t in range(n_frames):
        if t > 0:
            tr = np.concatenate([rng.uniform(-0.02, 0.02, 3), rng.uniform(-0.05, 0.05, 2),
                                 -rng.uniform(0.5, 1.5, 1)])
            tr_gt[t] = tr
            Rm, tt = rot_from_tr(tr)
            P = P @ Rm.T + tt
            u = f * P[:, 0] / P[:, 2] + cu
            v = f * P[:, 1] / P[:, 2] + cv
            keep = (P[:, 2] > 2.0) & (u >= 0) & (u <= width - 1) & (v >= 0) & (v <= height - 1)
            P, T = P[keep], T[keep]
            k_new = n_in - len(P)
            if k_new > 0:
                P = np.concatenate([P, _new_points(rng, k_new, width, height, zmin, zmax, f, cu, cv)])
                T = np.concatenate([T, new_patches(k_new)])
        uL = np.rint(f * P[:, 0] / P[:, 2] + cu).astype(int)
        vL = np.rint(f * P[:, 1] / P[:, 2] + cv).astype(int)
        uR = np.rint(f * (P[:, 0] - base) / P[:, 2] + cu).astype(int)
        order = rng.permutation(len(P))                      # same paint order in both views
        for side in (0, 1):
            img = rng.normal(128.0, 6.0, (height, width))
            uu = uL if side == 0 else uR
            vis = np.ones(len(P), bool) if side == 0 else ((uR >= 0) & (uR <= width - 1))
            for j in order:
                if not vis[j]:
                    continue
                x, y = uu[j], vL[j]
                x0, x1, y0, y1 = max(0, x - R), min(width, x + R + 1), max(0, y - R), min(height, y + R + 1)
                img[y0:y1, x0:x1] = T[j][y0 - (y - R):y1 - (y - R), x0 - (x - R):x1 - (x - R)]
            img = img + rng.normal(0.0, noise_sigma, img.shape)
            images[t, side] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
            idx = np.nonzero(vis)[0][:n_kp]
            k_real = len(idx)
            k_out = n_kp - k_real
            xy = np.empty((n_kp, 2), np.float32)
            xy[:k_real, 0] = uu[idx]
            xy[:k_real, 1] = vL[idx]
            xy[k_real:, 0] = rng.integers(0, width, k_out)
            xy[k_real:, 1] = rng.integers(0, height, k_out)
            perm = rng.permutation(n_kp)
            kp[t, side, :n_kp] = xy[perm]
            n[t, side] = n_kp
    from . import hostmath
    F = hostmath.F_from_P(KITTI_P1, KITTI_P2)
    param = Param.default(base=base, f=f, cu=cu, cv=cv)
    return dict(images=images, kp=kp, n=n, tr_gt=tr_gt, param=param, F=F, P1=KITTI_P1, P2=KITTI_P2,
                width=width, height=height)
