"""Randomised differential test of viso_match_desc against the oracle: many
small problems with adversarial structure (duplicated keypoints and patches,
negative / fractional / huge coordinates, NaN and inf keypoints, radius 0 and
very large radii, K from 1 to beyond n, both gates on and off, odd descriptor
lengths, non-integer descriptors)."""
import os

import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu


def _random_F(rng, oracle):
    """Fundamental matrices of every kind the epipolar band (match_dev.h, epipolar_band) must stay sound for:
    rectified, slightly de-rectified (finite band of a few pixels), general motion, arbitrary, rescaled, degenerate."""
    K = synth.KITTI_P1[:, :3]
    kind = rng.integers(0, 6)
    if kind == 0:
        return oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    if kind in (1, 2):
        a = rng.uniform(-1, 1, 3) * (10.0 ** rng.uniform(-5, -1.5) if kind == 1 else 0.3)
        R, _ = synth.rot_from_tr(np.r_[a, 0, 0, 0])
        t = np.array([-0.54, 0, 0]) + (rng.uniform(-0.05, 0.05, 3) if kind == 2 else rng.uniform(-1e-3, 1e-3, 3))
        P2 = K @ np.c_[R, t]
        return oracle.F_from_P(synth.KITTI_P1, P2)
    if kind == 3:
        return rng.normal(0, 1, (3, 3)) * 10.0 ** rng.uniform(-6, 3)
    if kind == 4:
        return oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2) * 10.0 ** rng.uniform(-12, 12)
    F = np.zeros((3, 3))
    if rng.random() < 0.5:
        F[rng.integers(0, 3), rng.integers(0, 3)] = rng.normal()
    return F


def _case(rng, F):
    n1, n2 = int(rng.integers(1, 140)), int(rng.integers(0, 140))
    span = float(rng.choice([8, 40, 200, 2000]))
    kind = rng.integers(0, 5)
    kp1 = rng.integers(0, span, (n1, 2)).astype(np.float32)
    kp2 = rng.integers(0, span, (n2, 2)).astype(np.float32)
    if kind == 1:                                   # fractional and negative coordinates
        kp1 = (kp1 - span / 2 + rng.uniform(-0.5, 0.5, kp1.shape)).astype(np.float32)
        kp2 = (kp2 - span / 2 + rng.uniform(-0.5, 0.5, kp2.shape)).astype(np.float32)
    elif kind == 2 and n2 > 3:                      # heavy duplication of keypoints
        kp2[:] = kp2[rng.integers(0, 3, n2)]
        kp1[:] = kp2[rng.integers(0, n2, n1)]
    elif kind == 3:                                 # NaN / inf / huge values sprinkled in
        for a in (kp1, kp2):
            if len(a):
                a[rng.integers(0, len(a), max(1, len(a) // 10)), rng.integers(0, 2)] = rng.choice([np.nan, np.inf, -np.inf, 3e38, -1e30])
    dlen = int(rng.choice([121, 121, 121, 1, 7, 64, 128, 130]))
    lo, hi = (-3, 4) if rng.random() < 0.4 else (-1020, 1021)          # tiny range => many exact SAD ties
    d1 = rng.integers(lo, hi, (n1, dlen)).astype(np.float32)
    d2 = rng.integers(lo, hi, (n2, dlen)).astype(np.float32)
    if n2 and rng.random() < 0.5:                   # planted matches and duplicated patches
        k = rng.integers(0, n2, n1)
        d1[:] = d2[k] + rng.integers(-2, 3, (n1, dlen))
        if rng.random() < 0.5:
            d2[rng.integers(0, n2, max(1, n2 // 4))] = d2[rng.integers(0, n2, max(1, n2 // 4))]
    if rng.random() < 0.15:
        d1 = (d1 * 0.5).astype(np.float32); d2 = (d2 * 0.5).astype(np.float32)   # non-integers: general path
    mp = MatchParams.stereo(F) if rng.random() < 0.4 else MatchParams.temporal()
    mp.enforce_2nd_best = int(rng.random() < 0.5)
    mp.ratio_2nd_best = float(rng.choice([0.5, 0.8, 0.9, 1.0, 1.5]))
    mp.max_neighbors = int(rng.choice([1, 2, 5, 50, 200, 250, 1000]))
    mp.radius = float(rng.choice([0, 1, 7.5, 80, 80, 500, 1e6]))
    mp.sampson_thresh = float(rng.choice([1.0, 1.0, 0.3, 4.0]))
    return kp1, kp2, d1, d2, mp


@pytest.mark.parametrize("variant", libviso_amd.MATCHER_VARIANTS)
def test_match_desc_randomised(viso, oracle, variant):
    F = oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    rng = np.random.default_rng(20260 + variant)
    libviso_amd.set_matcher_variant(variant)
    try:
        n_nonempty = 0
        for it in range(int(os.environ.get("VISO_FUZZ_ITERS", "220"))):   # VISO_FUZZ_ITERS=5000 for a long soak
            kp1, kp2, d1, d2, mp = _case(rng, F if it % 2 else _random_F(rng, oracle))
            want = oracle.match_desc(kp1, kp2, d1, d2, mp)
            got = libviso_amd.match_desc(kp1, kp2, d1, d2, mp)
            assert np.array_equal(got, want), (variant, it, len(kp1), len(kp2), d1.shape[1], mp.max_neighbors, mp.radius,
                                               mp.enforce_epipolar, mp.enforce_2nd_best)
            n_nonempty += len(want) > 0
        assert n_nonempty > 100
    finally:
        libviso_amd.set_matcher_variant(libviso_amd.DEFAULT_MATCHER)


def _medium_case(rng, F):
    """Image-sized problems: hundreds to thousands of keypoints with clusters, so that the tile / round / window
    machinery of the batch kernels (union lists, stereo passes, overflow queue, bucket sort of the match list) runs at
    its real sizes."""
    n1, n2 = int(rng.integers(200, 2600)), int(rng.integers(200, 2600))
    w, h = float(rng.choice([300, 1241, 2048])), float(rng.choice([100, 376, 1024]))

    def pts(n):
        k = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1)
        if rng.random() < 0.5:   # clusters: dense blobs that overflow lists and K caps
            nb = int(rng.integers(1, 6))
            c = np.stack([rng.uniform(0, w, nb), rng.uniform(0, h, nb)], 1)
            m = rng.random(n) < rng.uniform(0.2, 0.8)
            k[m] = c[rng.integers(0, nb, m.sum())] + rng.normal(0, rng.uniform(2, 30), (m.sum(), 2))
        if rng.random() < 0.5:
            k = np.round(k)
        return k.astype(np.float32)
    kp2 = pts(n2)
    kp1 = pts(n1)
    lo, hi = (-4, 5) if rng.random() < 0.25 else (-1020, 1021)
    d2 = rng.integers(lo, hi, (n2, 121)).astype(np.float32)
    d1 = rng.integers(lo, hi, (n1, 121)).astype(np.float32)
    k = int(rng.uniform(0, 0.9) * min(n1, n2))
    if k:   # planted correspondences (small motion, descriptor noise)
        src = rng.choice(n2, k, replace=False)
        kp1[:k] = kp2[src] + rng.normal(0, 3, (k, 2)).astype(np.float32)
        if rng.random() < 0.6:
            kp1[:k, 1] = kp2[src, 1] + rng.normal(0, 0.4, k).astype(np.float32)   # near-rectified rows for the stereo gate
        d1[:k] = d2[src] + rng.integers(-6, 7, (k, 121))
    mp = MatchParams.stereo(F) if rng.random() < 0.45 else MatchParams.temporal()
    mp.enforce_2nd_best = int(rng.random() < 0.6)
    mp.max_neighbors = int(rng.choice([20, 50, 200, 250, 250, 1000]))
    mp.radius = float(rng.choice([20, 80, 80, 80, 200]))
    return kp1, kp2, d1, d2, mp


def test_match_desc_randomised_image_sized(viso, oracle):
    F = oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    rng = np.random.default_rng(777)
    for it in range(int(os.environ.get("VISO_FUZZ_MEDIUM_ITERS", "6"))):   # VISO_FUZZ_MEDIUM_ITERS=200 for a soak
        kp1, kp2, d1, d2, mp = _medium_case(rng, F if it % 3 else _random_F(rng, oracle))
        want = oracle.match_desc(kp1, kp2, d1, d2, mp)
        got = libviso_amd.match_desc(kp1, kp2, d1, d2, mp)
        assert np.array_equal(got, want), (it, len(kp1), len(kp2), mp.max_neighbors, mp.radius, mp.enforce_epipolar, mp.enforce_2nd_best)


def test_stereo_gate_with_a_tiny_scaled_F(viso, oracle):
    """Sampson is scale invariant in exact arithmetic, but the reference squares a FLOAT: ad * ad (src/viso.cpp:664-665).
    With |F| ~ 1e-25 that product flushes to 0, the gate value is 0 and every in-radius candidate passes; with
    thresh = 0 only such candidates pass.  The epipolar band must not cull what the reference accepts there
    (epipolar_band returns +inf below the normal-float range)."""
    F0 = oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    rng = np.random.default_rng(4242)
    for scale, thresh in ((1e-25, 1.0), (1e-30, 1.0), (1e-21, 1.0), (1e-19, 1.0), (1e-17, 1.0), (1.0, 0.0), (1e-25, 0.0)):
        for it in range(2):
            kp1, kp2, d1, d2, mp = _medium_case(rng, F0 * scale)
            mp = MatchParams.stereo(F0 * scale)
            mp.sampson_thresh = thresh
            mp.max_neighbors = 200 if it else 1000
            want = oracle.match_desc(kp1, kp2, d1, d2, mp)
            got = libviso_amd.match_desc(kp1, kp2, d1, d2, mp)
            assert np.array_equal(got, want), (scale, thresh, it, len(kp1), len(kp2))


# the other plain calls (tests/api_fuzz.py): descriptor extraction at and beyond the image borders, the circle join on lists
# with repeated / out-of-range indices and tiny capacities, collect / triangulate with zero disparities, the solver calls on
# ill-conditioned point sets
def test_plain_calls_randomised(viso, oracle):
    import api_fuzz
    bad = api_fuzz.run(1, 150, libviso_amd, oracle)
    assert not bad, bad[:5]


# whole sequences through the batch family (tests/batch_fuzz.py): features in with every matcher variant, ragged / empty /
# repeated frames, odd capacities; images in with keypoints on the borders; images in with the detector on random bin grids
def test_batch_family_randomised(viso, oracle):
    import batch_fuzz
    bad = batch_fuzz.run(3, 30, libviso_amd, oracle)
    assert not bad, bad[:5]
