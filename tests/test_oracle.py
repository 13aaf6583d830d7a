"""Pins the CPU oracle (no GPU): analytic known answers taken from the
reference's own (disabled) tests, hand-built quirk cases Q1-Q8, and an
independent numpy restatement.  The reference holds no golden vectors for this
path, so these are what 'pinned' means here."""
import numpy as np
import pytest

from libviso_amd import synth
from libviso_amd.abi import MatchParams, Param

import npref


def _rand_problem(rng, n1, n2, width=200, height=100, dlen=121, planted=0.5, lo=-300, hi=300):
    kp2 = np.stack([rng.integers(0, width, n2), rng.integers(0, height, n2)], 1).astype(np.float32)
    d2 = rng.integers(lo, hi, (n2, dlen)).astype(np.float32)
    kp1 = np.stack([rng.integers(0, width, n1), rng.integers(0, height, n1)], 1).astype(np.float32)
    d1 = rng.integers(lo, hi, (n1, dlen)).astype(np.float32)
    k = int(planted * min(n1, n2))
    if k:
        src = rng.choice(n2, k, replace=False)
        kp1[:k] = kp2[src] + rng.integers(-6, 7, (k, 2))
        d1[:k] = d2[src] + rng.integers(-5, 6, (k, dlen))
    return kp1, kp2, d1, d2


# ------------------------------------------------------------ known answers
def test_F_from_P_known_answer(oracle):
    # reference src/mvg.cpp:73-89 (test_F_from_P, never called there)
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = np.hstack([np.eye(3), np.array([[1.0], [0], [0]])])
    F = oracle.F_from_P(P1, P2)
    assert np.array_equal(F, np.array([[0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]]))
    from libviso_amd import hostmath
    assert np.array_equal(hostmath.F_from_P(P1, P2), F)


def test_F_kitti_is_rectified_epipolar(oracle):
    F = oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    # for rectified pairs sampson == dy^2/2
    for dy, want in ((0, 0.0), (1, 0.5), (2, 2.0), (3, 4.5)):
        s = oracle.sampson_distance(F, (300.0, 100.0), (280.0, 100.0 + dy))
        assert abs(s - want) < 1e-5
        assert abs(npref.sampson(F, (300.0, 100.0), (280.0, 100.0 + dy)) - s) <= 1e-12 * max(1, s)


def test_tr2mat_convention(oracle):
    tr = np.array([0.1, -0.2, 0.3, 1.0, 2.0, 3.0])
    T = oracle.tr2mat(tr)
    R, t = synth.rot_from_tr(tr)
    assert np.allclose(T[:3, :3], R, atol=1e-15) and np.allclose(T[:3, 3], t)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
    # R = Rx*Ry*Rz
    cx, sx, cy, sy, cz, sz = np.cos(.1), np.sin(.1), np.cos(-.2), np.sin(-.2), np.cos(.3), np.sin(.3)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    assert np.allclose(R, Rx @ Ry @ Rz, atol=1e-15)
    P = oracle.pose_update(np.eye(4), tr)
    assert np.allclose(P @ T, np.eye(4), atol=1e-13)


def test_gn_recovers_known_motion(oracle):
    # recipe of reference test/test.cpp:51-114: tr0 = (0,0,0,1,0,0), all points
    # active, start from zero, sum |tr - tr0| < 1e-4 (with a correct baseline)
    rng = np.random.default_rng(0)
    param = Param.kitti00()
    X = rng.uniform(0, 1000, (3, 10))
    X[2] += 5
    tr0 = np.array([0, 0, 0, 1.0, 0, 0])
    obs = npref.project(X, tr0, param)
    ok, tr, iters = oracle.minimize_reproj(X, obs, np.zeros(6), param, np.arange(10))
    assert ok == 1 and np.abs(tr - tr0).sum() < 1e-4 and iters < 20


@pytest.mark.parametrize("seed", range(4))
def test_gn_recovers_random_motion(oracle, seed):
    X, obs, tr_gt, param = synth.make_solver_case(seed, m=60, outlier_frac=0.0, noise=0.0)
    ok, tr, _ = oracle.minimize_reproj(X, obs, np.zeros(6), param, np.arange(60))
    assert ok == 1 and np.abs(tr - tr_gt).max() < 2e-4  # stops once every step <= 1e-4


def test_ransac_recovers_motion_with_outliers(oracle):
    X, obs, tr_gt, param = synth.make_solver_case(3, m=300, outlier_frac=0.3, noise=0.2)
    ok, tr, inl = oracle.ransac_minimize_reproj(X, obs, param, seed=7, frame=1)
    assert ok == 1 and np.abs(tr - tr_gt).max() < 5e-3 and 150 < len(inl) <= 300
    # inliers are ascending and satisfy the threshold
    assert np.all(np.diff(inl) > 0)
    pred = npref.project(X, tr, param)
    e2 = ((obs - pred) ** 2).sum(0)
    assert np.array_equal(np.nonzero(e2 < 4.0)[0], inl)


# ----------------------------------------------------------- solver pieces
def test_compute_J_matches_finite_differences_and_q6(oracle):
    X, obs, tr_gt, param = synth.make_solver_case(1, m=30, outlier_frac=0.0, noise=0.5)
    tr = tr_gt * 0.7
    active = np.array([5, 2, 17, 9], np.int32)
    J, pred, res = oracle.compute_J(X, obs, tr, param, active)
    full = npref.project(X, tr, param)
    assert np.allclose(pred, full[:, active], rtol=1e-13, atol=1e-10)
    # Q6: weight uses observe(0, i) with i the POSITION, not active[i]
    w = 1.0 / (np.abs(obs[0, :4] - param.cu) / abs(param.cu) + 0.05)
    want_res = (w[None, :] * (obs[:, active] - pred)).T.reshape(-1)
    assert np.allclose(res, want_res, rtol=1e-13, atol=1e-12)
    eps = 1e-6
    for j in range(6):
        d = np.zeros(6); d[j] = eps
        num = (npref.project(X, tr + d, param)[:, active] - npref.project(X, tr - d, param)[:, active]) / (2 * eps)
        want = (w[None, :] * num).T.reshape(-1)
        assert np.allclose(J[:, j], want, rtol=1e-5, atol=1e-5)
    # rows 1 and 3 of each point are identical (:1479,:1481)
    assert np.array_equal(J[1::4], J[3::4])


def test_gn_step_is_normal_equations(oracle):
    X, obs, tr_gt, param = synth.make_solver_case(2, m=40, outlier_frac=0.0, noise=0.3)
    active = np.arange(40, dtype=np.int32)
    J, _, res = oracle.compute_J(X, obs, np.zeros(6), param, active)
    ok, p = oracle.lu_solve6(J.T @ J, J.T @ res)
    assert ok == 1
    assert np.allclose(p, np.linalg.lstsq(J, res, rcond=None)[0], rtol=1e-8, atol=1e-12)
    # one iteration of minimize_reproj == tr + p (when some p_j > thresh)
    p2 = Param.kitti00(); p2.thresh = 1e-4
    # emulate: run the oracle with a 1-iteration budget by checking the second J evaluation point
    ok2, tr, iters = oracle.minimize_reproj(X, obs, np.zeros(6), param, active)
    assert ok2 == 1 and iters >= 2


def test_lu_singular_and_pivoting(oracle):
    A = np.diag([1.0, 2, 3, 4, 5, 6]); A[0, 0] = 1e-17
    ok, _ = oracle.lu_solve6(A, np.ones(6))
    assert ok == 0                      # |pivot| < DBL_EPSILON -> solve() == false (:1602-1606)
    rng = np.random.default_rng(5)
    A = rng.normal(size=(6, 6)); A[0, 0] = 0.0
    b = rng.normal(size=6)
    ok, x = oracle.lu_solve6(A, b)
    assert ok == 1 and np.allclose(A @ x, b, atol=1e-10)


def test_q7_convergence_test_is_one_sided(oracle):
    # fabs(p > thresh): a large NEGATIVE first step counts as converged and is
    # not applied (:1610-1617). One point pair geometry that makes every
    # component of the first step <= 0 is easiest to get by flipping the sign
    # of the motion: recover tr0 with all-negative components.
    rng = np.random.default_rng(11)
    param = Param.kitti00()
    X = np.stack([rng.uniform(-10, 10, 30), rng.uniform(-2, 2, 30), rng.uniform(8, 40, 30)])
    tr0 = -np.array([0.01, 0.012, 0.008, 0.3, 0.2, 0.9])
    obs = npref.project(X, tr0, param)
    ok, tr, iters = oracle.minimize_reproj(X, obs, np.zeros(6), param, np.arange(30))
    if iters == 1:
        assert ok == 1 and np.array_equal(tr, np.zeros(6))   # "converged" without moving
    else:  # some component of the first step was positive; then it must be exact
        assert ok == 1


def test_get_inliers_threshold_strict_and_rms_q8(oracle):
    param = Param.kitti00()
    X = np.array([[0.0, 1.0, -1.0], [0.0, 0.5, 0.2], [10.0, 12.0, 15.0]])
    pred = npref.project(X, np.zeros(6), param)
    obs = pred.copy()
    obs[0, 1] += 2.0          # err2 == 4 exactly -> NOT an inlier (strict <)
    obs[1, 2] += 1.5
    inl, rms = oracle.get_inliers(X, obs, np.zeros(6), param)
    assert list(inl) == [0, 2]
    assert abs(rms - np.sqrt(1.5 ** 2 / 3)) < 1e-12   # last point's error only


def test_ransac_samples_are_ascending_distinct_and_partition_invariant(oracle):
    s = oracle.ransac_samples(123, 5, 50, 40)
    assert s.shape == (50, 3) and np.all(s[:, 0] < s[:, 1]) and np.all(s[:, 1] < s[:, 2])
    assert s.min() >= 0 and s.max() < 40
    assert np.array_equal(s, oracle.ransac_samples(123, 5, 50, 40))
    assert not np.array_equal(s, oracle.ransac_samples(123, 6, 50, 40))
    assert np.array_equal(oracle.ransac_samples(9, 0, 20, 3), np.tile([0, 1, 2], (20, 1)))
    assert np.array_equal(oracle.ransac_samples(9, 0, 20, 2), np.zeros((20, 3), np.int32))
    # roughly uniform
    big = oracle.ransac_samples(1, 0, 4000, 10)
    cnt = np.bincount(big.reshape(-1), minlength=10)
    assert cnt.min() > 1000 and cnt.max() < 1400


def _chi2(counts, expected):
    return float(((counts - expected) ** 2 / expected).sum())


@pytest.mark.parametrize("algorithm_s", [False, True])
def test_ransac_samples_are_uniform_three_subsets(oracle, algorithm_s):
    """randomsample(3, N, .) (src/viso.cpp:87-107) yields every 3-subset of 0..N-1 with probability 1 / C(N,3), ascending.
    The O(1) definition of round 6 (three draws through Floyd's subset sampling) must have that distribution -- and so
    must the literal algorithm S over the same stream, the definition of rounds 1-5, which is what the reference runs:
    the same test on both.  Chi-square against the uniform law over all subsets (small N), over the per-position
    marginals (larger N: P(first = a) = C(N-1-a, 2) / C(N,3), ...) and over the pair (gap1, gap2)."""
    from itertools import combinations
    from math import comb
    # all subsets, N = 5, 6, 7: 10 / 20 / 35 cells
    for N in (4, 5, 6, 7):
        n = 60000
        s = oracle.ransac_samples(77, N, n, N, algorithm_s=algorithm_s)
        assert np.all(s[:, 0] < s[:, 1]) and np.all(s[:, 1] < s[:, 2]) and s.min() >= 0 and s.max() < N
        idx = {c: i for i, c in enumerate(combinations(range(N), 3))}
        counts = np.bincount([idx[tuple(r)] for r in s.tolist()], minlength=len(idx)).astype(float)
        k = len(idx)
        # chi-square with k - 1 degrees of freedom: mean k - 1, sd sqrt(2 (k - 1)); 6 sd is a one-in-a-billion event
        assert _chi2(counts, n / k) < (k - 1) + 6 * np.sqrt(2 * (k - 1)), (N, counts)
    # marginals of every position, N = 50 and N = 1200 (binned)
    for N in (50, 1200):
        n = 200000
        s = oracle.ransac_samples(5, 3 * N, n, N, algorithm_s=algorithm_s)
        assert np.all(s[:, 0] < s[:, 1]) and np.all(s[:, 1] < s[:, 2]) and s.min() >= 0 and s.max() < N
        tot = comb(N, 3)
        a = np.arange(N)
        p_first = np.array([comb(N - 1 - int(x), 2) for x in a]) / tot
        p_mid = np.array([int(x) * (N - 1 - int(x)) for x in a]) / tot
        p_last = np.array([comb(int(x), 2) for x in a]) / tot
        for col, p in ((0, p_first), (1, p_mid), (2, p_last)):
            counts = np.bincount(s[:, col], minlength=N).astype(float)
            # merge cells into bins of expected count >= 50
            order = np.argsort(-p)
            bins_c, bins_e, c_acc, e_acc = [], [], 0.0, 0.0
            for i in order:
                c_acc += counts[i]; e_acc += n * p[i]
                if e_acc >= 50:
                    bins_c.append(c_acc); bins_e.append(e_acc); c_acc = e_acc = 0.0
            if e_acc > 0:
                bins_c[-1] += c_acc; bins_e[-1] += e_acc
            k = len(bins_c)
            assert _chi2(np.array(bins_c), np.array(bins_e)) < (k - 1) + 6 * np.sqrt(2 * (k - 1)), (N, col)
        # every index equally often over all positions: 3 / N each
        counts = np.bincount(s.reshape(-1), minlength=N).astype(float)
        assert _chi2(counts, 3 * n / N) < (N - 1) + 6 * np.sqrt(2 * (N - 1))
    # hypotheses of one frame and frames of one hypothesis are different streams
    assert len({tuple(r) for r in oracle.ransac_samples(1, 1, 50, 2000, algorithm_s=algorithm_s).tolist()}) >= 49
    assert len({tuple(oracle.ransac_samples(1, f, 1, 2000, algorithm_s=algorithm_s)[0].tolist()) for f in range(50)}) >= 49


def test_triangulate_no_clamp(oracle):
    param = Param.kitti00()
    x = np.array([[700.0, 650.0, 600.0], [200.0, 100.0, 50.0], [690.0, 650.0, 610.0], [200.0, 100.0, 50.0]])
    with np.errstate(divide="ignore", invalid="ignore"):
        X = oracle.triangulate_rectified(x, param)
    assert abs(X[2, 0] - param.f * param.base / 10.0) < 1e-12
    assert np.isinf(X[2, 1])            # d == 0 -> inf passes through (:1148-1151)
    assert X[2, 2] < 0                  # negative disparity -> negative depth


# ------------------------------------------------------------------ matcher
@pytest.mark.parametrize("seed", range(6))
def test_match_desc_vs_numpy_restatement(oracle, seed):
    rng = np.random.default_rng(seed)
    n1, n2 = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    kp1, kp2, d1, d2 = _rand_problem(rng, n1, n2)
    F = oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    for mp in (MatchParams.temporal(), MatchParams.stereo(F)):
        if seed % 2:
            mp.max_neighbors = 7      # exercise the K cap
        got, sc = oracle.match_desc(kp1, kp2, d1, d2, mp, return_scored=True)
        want, wsc = npref.match_desc(kp1, kp2, d1, d2, mp)
        assert np.array_equal(got, want) and sc == wsc


def test_radius_search_order_and_padding(oracle):
    kp2 = np.array([[10, 10], [12, 10], [10, 12], [50, 50], [11, 10], [10, 10]], np.float32)
    kp1 = np.array([[10, 10], [200, 200]], np.float32)
    nei, found = oracle.radius_search(kp1, kp2, 5.0, 4)
    # (dist, idx) ascending: 0(d0) 5(d0) 4(d1) 1(d2) 2(d2) ; K=4 keeps the first four
    assert list(nei[0]) == [0, 5, 4, 1] and found[0] == 5
    assert list(nei[1]) == [-1, -1, -1, -1] and found[1] == 0
    assert np.array_equal(nei, npref.neighbours(kp1, kp2, 5.0, 4))
    # inclusive radius
    nei, _ = oracle.radius_search(np.array([[0, 0]], np.float32), np.array([[9, 9], [3, 2], [5, 0.5]], np.float32), 5.0, 3)
    assert list(nei[0]) == [1, -1, -1]


def test_q1_target_zero_never_matches_and_truncates(oracle):
    d = np.zeros((3, 121), np.float32)
    kp2 = np.array([[10, 10], [14, 10], [30, 10]], np.float32)   # idx0 at dist 0 from query
    d2 = d.copy(); d2[1] += 5; d2[2] += 1
    kp1 = np.array([[10, 10]], np.float32)
    mp = MatchParams.temporal(); mp.enforce_2nd_best = 0
    # list = [0, 1, 2] -> stops immediately at index 0 -> no match at all
    assert len(oracle.match_desc(kp1, kp2, d[:1], d2, mp)) == 0
    # move the query so that idx 1 comes first: list = [1, 0, 2] -> only idx 1 is scored
    kp1 = np.array([[13, 10]], np.float32)
    m = oracle.match_desc(kp1, kp2, d[:1], d2, mp)
    assert m.tolist() == [[0, 1, 5 * 121]]


def test_q2_q3_ties(oracle):
    kp2 = np.array([[0, 0], [10, 10], [11, 10], [12, 10], [13, 10]], np.float32)
    d2 = np.zeros((5, 121), np.float32); d2[1] += 2; d2[2] += 2; d2[3] += 3; d2[4] += 2
    kp1 = np.array([[10, 10]], np.float32)
    d1 = np.zeros((1, 121), np.float32)
    mp = MatchParams.temporal(); mp.enforce_2nd_best = 0
    # equal SAD at idx 1,2,4: the LAST in (dist, idx) order wins (Q2) -> idx 4
    assert oracle.match_desc(kp1, kp2, d1, d2, mp).tolist() == [[0, 4, 242]]
    mp.enforce_2nd_best = 1
    # tie => best_d2 == best_d1 => ratio test fails (Q3)
    assert len(oracle.match_desc(kp1, kp2, d1, d2, mp)) == 0
    # a single candidate is accepted (best_d2 = DBL_MAX)
    assert oracle.match_desc(kp1, kp2[:2], d1, d2[:2], mp).tolist() == [[0, 1, 242]]
    # 0 < 0*0.9 is false: identical descriptors twice are rejected
    d2[:] = 0
    assert len(oracle.match_desc(kp1, kp2, d1, d2, mp)) == 0


def test_sort_order_is_dist_then_i1(oracle):
    rng = np.random.default_rng(3)
    kp1, kp2, d1, d2 = _rand_problem(rng, 60, 60, lo=-2, hi=3)   # tiny range -> many equal SADs
    mp = MatchParams.temporal(); mp.enforce_2nd_best = 0
    m = oracle.match_desc(kp1, kp2, d1, d2, mp)
    key = m[:, 2].astype(np.int64) * 100000 + m[:, 0]
    assert len(m) > 10 and np.all(np.diff(key) > 0)


def test_match_circle_literal(oracle):
    lr = np.array([[0, 5, 1], [1, 6, 2], [2, 7, 3]], np.int32)
    m11 = np.array([[1, 11, 0], [0, 10, 0], [2, 12, 0]], np.int32)
    lrp = np.array([[12, 22, 0], [10, 20, 0], [11, 21, 0]], np.int32)
    m22 = np.array([[5, 20, 0], [6, 99, 0], [7, 22, 0]], np.int32)
    r, circ, pcl, n = oracle.match_circle(lr, lrp, m11, m22)
    assert r == 1 and n == 2
    assert circ.tolist() == [[0, 5, 10, 20], [2, 7, 12, 22]] and pcl.tolist() == [[0, 1], [2, 0]]
    # duplicate keys: nested-loop order, every combination is emitted
    m11d = np.array([[0, 10, 0], [0, 10, 0]], np.int32)
    r, circ, pcl, n = oracle.match_circle(lr[:1], lrp, m11d, m22)
    assert n == 2 and pcl.tolist() == [[0, 1], [0, 1]]
    r, _, _, n = oracle.match_circle(lr, lrp, m11, m22, cap=1)
    assert r == -1 and n == 2


def test_descriptor_contract(oracle):
    from scipy import ndimage
    img = synth.make_images(4, 40, 56)
    k = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float64)
    sob = ndimage.correlate(img.astype(np.float64), k, mode="mirror")   # == BORDER_REFLECT_101
    kp = np.array([[20, 20], [0, 0], [55, 39], [3, 2], [5, 5], [50, 34]], np.float32)
    d = oracle.extract_descriptors(img, kp, 5)
    assert d.shape == (6, 121) and np.all(d == np.rint(d)) and np.abs(d).max() <= 1020
    for n, (x, y) in enumerate(kp.astype(int)):
        col = 0
        for i in range(-5, 6):
            for j in range(-5, 6):
                yy, xx = y + i, x + j
                want = sob[yy, xx] if (0 < yy < 40 and 0 < xx < 56) else 0.0   # strict > 0 (:1018)
                assert d[n, col] == want
                col += 1


def test_sequence_driver_runs_and_recovers_motion(oracle):
    seq = synth.make_sequence(5, 4, n_kp=400, width=400, height=200)
    F = seq["F"]
    out = oracle.sequence(seq["kp"], seq["desc"], seq["n"], MatchParams.stereo(F),
                          MatchParams.temporal(), seq["param"], seed=1)
    assert out["ok"][0] == 0 and np.all(out["ok"][1:] == 1)
    assert np.abs(out["tr"][1:] - seq["tr_gt"][1:]).max() < 2e-2
    assert out["scored"][1, 1:].min() > 400 and out["m_out"][0].min() > 50


def test_harris_response_vs_scipy(oracle):
    # cv::cornerHarris(blockSize 3, ksize 5, k) restated: Sobel 5x5 (deriv x smooth, reflect-101) scaled by
    # 1/(2^4*3*255), unnormalised 3x3 box of (dx^2, dxdy, dy^2) with reflect-101, det - k*trace^2
    from scipy import ndimage
    img = synth.make_images(3, 60, 96)
    r = oracle.harris_response(img)
    d = np.array([-1, -2, 0, 2, 1.0]); s = np.array([1, 4, 6, 4, 1.0])
    I = img.astype(np.float64)
    Dx = ndimage.correlate1d(ndimage.correlate1d(I, d, axis=1, mode="mirror"), s, axis=0, mode="mirror") / 12240
    Dy = ndimage.correlate1d(ndimage.correlate1d(I, s, axis=1, mode="mirror"), d, axis=0, mode="mirror") / 12240
    box = lambda A: ndimage.uniform_filter(A, 3, mode="mirror") * 9
    a, b, c = box(Dx * Dx), box(Dx * Dy), box(Dy * Dy)
    R = a * c - b * b - oracle.HARRIS_K * (a + c) ** 2
    assert np.abs(r - R).max() <= 2e-6 * np.abs(R).max()
    # the two evaluation orders (OpenCV's: scaled float smoothing kernel, symmetric grouping, row sums then column
    # sums; and the rounds 1-3 one: exact integer Sobel sums times the scale) are the same function up to float rounding
    r1 = oracle.harris_response_v1(img)
    assert np.abs(r1 - R).max() <= 2e-6 * np.abs(R).max() and np.abs(r - r1).max() <= 2e-6 * np.abs(R).max()
    assert not np.array_equal(r, r1)                      # ... and they do differ in the last bits: the order matters


def test_harris_order_by_hand(oracle):
    """oracle_harris_response against a numpy float32 transcription of OpenCV's evaluation order, written from the
    OpenCV sources' description (deriv.cpp: the scale goes into the smoothing kernel; filter.cpp: RowFilter adds left to
    right, SymmColumnFilter groups f1*(S1 + S-1); smooth.cpp: row sums, then column sums), every operation rounded to
    float32 in that order."""
    f = np.float32
    img = synth.make_images(11, 23, 31)
    rows, cols = img.shape
    pad = lambda A, n: np.pad(A, n, mode="reflect")
    scale = f(1.0 / (16.0 * 3.0 * 255.0))
    tap = [f(s) * scale for s in (1, 4, 6, 4, 1)]
    P = pad(img.astype(f), 2)
    px = lambda j: P[2:2 + rows, j:j + cols]                     # column x + j - 2
    H = f(-1) * px(0); H = H + f(-2) * px(1); H = H + f(0) * px(2); H = H + f(2) * px(3); H = H + f(1) * px(4)
    G = tap[0] * px(0)
    for j in range(1, 5):
        G = (G + tap[j] * px(j)).astype(f)
    Hp, Gp = pad(H.astype(f), ((2, 2), (0, 0))), pad(G, ((2, 2), (0, 0)))
    ry = lambda A, i: A[2 + i:2 + i + rows]
    dx = (tap[2] * ry(Hp, 0) + f(0)).astype(f)
    dx = (dx + tap[3] * (ry(Hp, 1) + ry(Hp, -1)).astype(f)).astype(f)
    dx = (dx + tap[4] * (ry(Hp, 2) + ry(Hp, -2)).astype(f)).astype(f)
    dy = (f(2) * (ry(Gp, 1) - ry(Gp, -1)).astype(f)).astype(f)
    dy = (dy + (ry(Gp, 2) - ry(Gp, -2)).astype(f)).astype(f)
    out = []
    for cv in ((dx * dx).astype(f), (dx * dy).astype(f), (dy * dy).astype(f)):
        C1 = pad(cv, ((0, 0), (1, 1)))
        rs = ((C1[:, 0:cols] + C1[:, 1:cols + 1]).astype(f) + C1[:, 2:cols + 2]).astype(f)
        R1 = pad(rs, ((1, 1), (0, 0)))
        out.append(((R1[0:rows] + R1[1:rows + 1]).astype(f) + R1[2:rows + 2]).astype(f))
    a, b, c = out
    t3 = ((a * c).astype(f) - (b * b).astype(f)).astype(f)
    tr = (a + c).astype(f)
    want = (t3.astype(np.float64) - oracle.HARRIS_K * tr.astype(np.float64) * tr.astype(np.float64)).astype(f)
    assert np.array_equal(oracle.harris_response(img), want)


def test_harris_binned_selection(oracle):
    img = synth.make_images(5, 80, 120)
    kp, resp = oracle.detect_harris_binned(img, 120, 6, 4)        # 5 per bin, 24 bins
    R = np.abs(oracle.harris_response(img))
    assert len(kp) == 120 and np.all(resp > 0)
    sx, sy = 120 // 6, 80 // 4
    for b in range(24):
        bx, by = b // 4, b % 4                                     # binx outer, biny inner (:949-951)
        blk = R[by * sy:(by + 1) * sy, bx * sx:(bx + 1) * sx]
        top = np.sort(blk.reshape(-1))[::-1][:5]
        assert np.array_equal(resp[5 * b:5 * b + 5], top)          # the 5 largest |response|, descending
        for (x, y), v in zip(kp[5 * b:5 * b + 5].astype(int), resp[5 * b:5 * b + 5]):
            assert bx * sx <= x < (bx + 1) * sx and by * sy <= y < (by + 1) * sy and R[y, x] == v
    # reference defaults: 1200 features, 24 x 5 bins, 10 per bin (src/viso.cpp:1171-1172, 915)
    big = synth.make_image_sequence(1, 1, n_kp=300, width=480, height=200)["images"][0, 0]
    assert len(oracle.detect_harris_binned(big)[0]) == 1200
