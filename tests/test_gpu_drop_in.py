"""The literal drop-in flow: the reference's sequence_odometry loop (src/viso.cpp:1205-1327) calling the plain C-ABI one
function at a time, in C++ (`viso::sequence_odometry_per_call` through libviso_host.so).  It must give what the batch
family and the oracle give on the same frames, whatever the plain family does behind the calls (image cache,
computing a frame's later calls ahead): those only move work, never a result -- every combination is run."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import drop_in, synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-5


def rel_fro(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def seq():
    return synth.make_sequence(77, 14, n_kp=500, width=620, height=188, ragged=True)


@pytest.fixture(autouse=True)
def _defaults_back():
    yield
    drop_in.plain_cache(True)
    drop_in.plain_speculate(True)


def batch_results(seq):
    nf, cap = seq["kp"].shape[0], seq["kp"].shape[2]
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, cap)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.set_params(st, tm, seq["param"], seed=5, first_frame=100)
    b.run()
    tr, ok, n_inl = b.poses()
    m = [[b.matches(w, t) for t in range(nf)] for w in range(3)]
    circ = [b.circle(t)[0] if t else np.zeros((0, 4), np.int32) for t in range(nf)]
    b.close(); ctx.close()
    return tr, ok, n_inl, m, circ


@pytest.mark.parametrize("cache,speculate", [(True, True), (True, False), (False, False), (False, True)])
def test_per_call_loop_equals_batch_family(viso, seq, cache, speculate):
    drop_in.plain_cache(cache)
    drop_in.plain_speculate(speculate)
    before = drop_in.plain_stats()
    o = drop_in.run(seq["kp"], seq["desc"], seq["n"], seq["F"], seq["param"], seed=5, first_frame=100, want_matches=True)
    after = drop_in.plain_stats()
    tr, ok, n_inl, m, circ = batch_results(seq)
    nf = seq["kp"].shape[0]
    assert o["frames"] == nf
    for t in range(nf):
        assert np.array_equal(o["matches"][0][t], m[0][t]), f"stereo matches of frame {t}"
        if t:
            assert np.array_equal(o["matches"][1][t], m[1][t]), f"temporal-left matches of frame {t}"
            assert np.array_equal(o["matches"][2][t], m[2][t]), f"temporal-right matches of frame {t}"
            assert o["n_circle"][t] == len(circ[t])
    assert np.array_equal(o["ok"], ok) and ok[1:].sum() >= nf - 3
    assert np.array_equal(o["n_inl"][ok == 1], n_inl[ok == 1])
    for t in range(1, nf):
        if ok[t]:
            assert rel_fro(libviso_amd.tr2mat(o["tr"][t]), libviso_amd.tr2mat(tr[t])) < POSE_TOL
    served = [a - b for a, b in zip(after["served"], before["served"])]
    if speculate and cache:     # the loop is recognised within a few frames, then every later call comes from the frame
        assert served[0] >= 2 * (nf - 6) and served[1] >= nf - 4 and served[3] >= nf - 8, served
    if not speculate:
        assert served == [0, 0, 0, 0]
    if cache:                   # every image crosses PCIe once
        assert after["misses"] - before["misses"] == 2 * nf


def test_per_call_loop_equals_oracle(viso, oracle, seq):
    o = drop_in.run(seq["kp"], seq["desc"], seq["n"], seq["F"], seq["param"], seed=5, first_frame=100, want_matches=True)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=5, first_frame=100)
    assert np.array_equal(o["ok"], want["ok"]) and np.array_equal(o["n_inl"], want["n_inl"])
    for t in range(1, seq["kp"].shape[0]):
        if want["ok"][t]:
            assert rel_fro(libviso_amd.tr2mat(o["tr"][t]), oracle.tr2mat(want["tr"][t])) < POSE_TOL
        n1, p1 = seq["n"][t, 0], seq["n"][t - 1, 0]
        m_cpu = oracle.match_desc(seq["kp"][t, 0, :n1], seq["kp"][t - 1, 0, :p1], seq["desc"][t, 0, :n1], seq["desc"][t - 1, 0, :p1], tm)
        assert np.array_equal(o["matches"][1][t], m_cpu)


def _loop_frame(seq, t, st, tm, state, seed=5, rs_param=None, tr0=None):
    """One iteration of the reference's loop body through the Python wrappers of the plain family; returns its products."""
    nL, nR = seq["n"][t]
    kp1, kp2 = seq["kp"][t, 0, :nL].copy(), seq["kp"][t, 1, :nR].copy()
    d1, d2 = seq["desc"][t, 0, :nL].copy(), seq["desc"][t, 1, :nR].copy()
    lr = libviso_amd.match_desc(kp1, kp2, d1, d2, st)
    x = libviso_amd.collect_matches(kp1, kp2, lr)
    X = libviso_amd.triangulate_rectified(x, seq["param"])
    out = {"kp1": kp1, "kp2": kp2, "d1": d1, "d2": d2, "lr": lr, "x": x, "X": X}
    if state is not None:
        m11 = libviso_amd.match_desc(kp1, state["kp1"].copy(), d1, state["d1"].copy(), tm)
        m22 = libviso_amd.match_desc(kp2, state["kp2"].copy(), d2, state["d2"].copy(), tm)
        _, circ, pcl, n = libviso_amd.match_circle(lr, state["lr"], m11, m22)
        out.update(m11=m11, m22=m22, circ=circ, pcl=pcl)
        if n >= 3:
            x_c, Xp_c = np.ascontiguousarray(x[:, pcl[:, 0]]), np.ascontiguousarray(state["X"][:, pcl[:, 1]])
            out["rs"] = libviso_amd.ransac_minimize_reproj(Xp_c, x_c, rs_param if rs_param is not None else seq["param"], seed=seed, frame=t, tr0=tr0)
            out.update(x_c=x_c, Xp_c=Xp_c)
    return out


def test_calls_that_are_not_the_loops_take_the_direct_path(viso, oracle, seq):
    """After the loop has been recognised, calls whose arguments differ from what a frame computed ahead -- one descriptor,
    one match row, another parameter, another stream key -- must get THEIR results (the oracle's), not the frame's."""
    drop_in.plain_cache(True)
    drop_in.plain_speculate(True)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    state = None
    for t in range(8):
        state = _loop_frame(seq, t, st, tm, state)
    s0 = drop_in.plain_stats()
    assert s0["served"][0] > 0 and s0["served"][1] > 0 and s0["served"][3] > 0, s0   # the loop IS being answered from frames
    prev = state
    t = 8
    nL, nR = seq["n"][t]
    kp1, kp2 = seq["kp"][t, 0, :nL].copy(), seq["kp"][t, 1, :nR].copy()
    d1, d2 = seq["desc"][t, 0, :nL].copy(), seq["desc"][t, 1, :nR].copy()
    lr = libviso_amd.match_desc(kp1, kp2, d1, d2, st)            # opens the frame, computes everything ahead
    assert np.array_equal(lr, oracle.match_desc(kp1, kp2, d1, d2, st))
    # collect_matches with two rows of the list exchanged
    lr2 = lr.copy(); lr2[[0, 1]] = lr2[[1, 0]]
    assert np.array_equal(libviso_amd.collect_matches(kp1, kp2, lr2), oracle.collect_matches(kp1, kp2, lr2))
    x = libviso_amd.collect_matches(kp1, kp2, lr)
    assert np.array_equal(x, oracle.collect_matches(kp1, kp2, lr))
    # triangulate_rectified with another baseline, and with one coordinate moved
    p2 = type(seq["param"]).from_buffer_copy(seq["param"]); p2.base = seq["param"].base * 1.25
    assert np.array_equal(libviso_amd.triangulate_rectified(x, p2), oracle.triangulate_rectified(x, p2))
    x2 = x.copy(); x2[0, 3] += 1.0
    assert np.array_equal(libviso_amd.triangulate_rectified(x2, seq["param"]), oracle.triangulate_rectified(x2, seq["param"]))
    X = libviso_amd.triangulate_rectified(x, seq["param"])
    assert np.array_equal(X, oracle.triangulate_rectified(x, seq["param"]))
    # temporal call with one descriptor element changed / another ratio
    d1b = d1.copy(); d1b[5, 7] += 3
    assert np.array_equal(libviso_amd.match_desc(kp1, prev["kp1"], d1b, prev["d1"], tm), oracle.match_desc(kp1, prev["kp1"], d1b, prev["d1"], tm))
    tm2 = MatchParams.temporal(); tm2.ratio_2nd_best = 0.8
    assert np.array_equal(libviso_amd.match_desc(kp1, prev["kp1"], d1, prev["d1"], tm2), oracle.match_desc(kp1, prev["kp1"], d1, prev["d1"], tm2))
    m11 = libviso_amd.match_desc(kp1, prev["kp1"], d1, prev["d1"], tm)
    m22 = libviso_amd.match_desc(kp2, prev["kp2"], d2, prev["d2"], tm)
    assert np.array_equal(m11, oracle.match_desc(kp1, prev["kp1"], d1, prev["d1"], tm))
    assert np.array_equal(m22, oracle.match_desc(kp2, prev["kp2"], d2, prev["d2"], tm))
    # match_circle with a row of match22 removed
    m22b = m22[1:].copy()
    _, c_a, p_a, n_a = libviso_amd.match_circle(lr, prev["lr"], m11, m22b)
    _, c_o, p_o, _ = oracle.match_circle(lr, prev["lr"], m11, m22b)
    assert n_a == len(c_o) and np.array_equal(c_a, c_o) and np.array_equal(p_a, p_o)
    _, circ, pcl, n = libviso_amd.match_circle(lr, prev["lr"], m11, m22)
    _, c_o, p_o, _ = oracle.match_circle(lr, prev["lr"], m11, m22)
    assert n == len(c_o) and np.array_equal(circ, c_o) and np.array_equal(pcl, p_o)
    assert n >= 3
    x_c, Xp_c = np.ascontiguousarray(x[:, pcl[:, 0]]), np.ascontiguousarray(prev["X"][:, pcl[:, 1]])
    # RANSAC with another stream key, another seed, one observation moved, another threshold
    for kw, xc in (({"seed": 5, "frame": t + 40}, x_c), ({"seed": 6, "frame": t}, x_c)):
        r_a, tr_a, inl_a = libviso_amd.ransac_minimize_reproj(Xp_c, xc, seq["param"], **kw)
        r_o, tr_o, inl_o = oracle.ransac_minimize_reproj(Xp_c, xc, seq["param"], **kw)
        assert r_a == r_o and np.array_equal(inl_a, inl_o)
        if r_o:
            assert rel_fro(libviso_amd.tr2mat(tr_a), oracle.tr2mat(tr_o)) < POSE_TOL
    xcb = x_c.copy(); xcb[1, 2] += 0.5
    p3 = type(seq["param"]).from_buffer_copy(seq["param"]); p3.inlier_threshold = 1.5
    for prm, xc in ((seq["param"], xcb), (p3, x_c)):
        r_a, tr_a, inl_a = libviso_amd.ransac_minimize_reproj(Xp_c, xc, prm, seed=5, frame=t)
        r_o, tr_o, inl_o = oracle.ransac_minimize_reproj(Xp_c, xc, prm, seed=5, frame=t)
        assert r_a == r_o and np.array_equal(inl_a, inl_o)
    r_a, tr_a, inl_a = libviso_amd.ransac_minimize_reproj(Xp_c, x_c, seq["param"], seed=5, frame=t)
    r_o, tr_o, inl_o = oracle.ransac_minimize_reproj(Xp_c, x_c, seq["param"], seed=5, frame=t)
    assert r_a == r_o and np.array_equal(inl_a, inl_o)


@pytest.mark.parametrize("speculate", [True, False])
def test_best_tr_survives_a_frame_without_support_in_both_modes(viso, oracle, seq, speculate):
    """ransac_minimize_reproj leaves the caller's best_tr alone when no hypothesis finds support (src/viso.cpp:1564-1568).  The
    stage that was computed ahead inside the stereo call cannot know the caller's array: it must say "nothing assigned"
    and the call must then leave the array as it came -- same as the direct path, same as the oracle.  A threshold of 0
    (err2 < 0 never holds) makes every frame of the loop such a frame."""
    drop_in.plain_cache(True)
    drop_in.plain_speculate(speculate)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    p0 = type(seq["param"]).from_buffer_copy(seq["param"]); p0.inlier_threshold = 0.0
    tr0 = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    before = drop_in.plain_stats()
    state, seen = None, 0
    for t in range(12):
        state = _loop_frame(seq, t, st, tm, state, rs_param=p0, tr0=tr0)
        if "rs" in state:
            r_a, tr_a, inl_a = state["rs"]
            r_o, tr_o, inl_o = oracle.ransac_minimize_reproj(state["Xp_c"], state["x_c"], p0, seed=5, frame=t, tr0=tr0)
            assert (r_a, len(inl_a)) == (r_o, len(inl_o)) == (0, 0), t
            assert np.array_equal(tr_o, tr0) and np.array_equal(tr_a.view(np.int64), tr0.view(np.int64)), (t, tr_a)
            seen += 1
    assert seen >= 10
    served = [a - b for a, b in zip(drop_in.plain_stats()["served"], before["served"])]
    assert (served[3] >= 4) if speculate else (served[3] == 0), served   # the answered-ahead path WAS the one under test


def test_an_image_that_does_not_fit_the_u16_rows_where_none_was_expected(viso, oracle, seq):
    """Integer-valued images for a while (the launches leave the general kernels out), then fractional descriptors: the
    call is repeated with the general kernels and gives the oracle's matches."""
    drop_in.plain_cache(True)
    drop_in.plain_speculate(True)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    state = None
    for t in range(5):
        state = _loop_frame(seq, t, st, tm, state)
    before = drop_in.plain_stats()["general_reruns"]
    t = 5
    nL, nR = seq["n"][t]
    kp1, kp2 = seq["kp"][t, 0, :nL].copy(), seq["kp"][t, 1, :nR].copy()
    d1, d2 = seq["desc"][t, 0, :nL].copy(), seq["desc"][t, 1, :nR].copy()
    d2[3, 4] += 0.5                                              # one fractional value in the right image
    lr = libviso_amd.match_desc(kp1, kp2, d1, d2, st)
    assert np.array_equal(lr, oracle.match_desc(kp1, kp2, d1, d2, st))
    assert drop_in.plain_stats()["general_reruns"] == before + 1
    m11 = libviso_amd.match_desc(kp1, state["kp1"], d1, state["d1"], tm)
    m22 = libviso_amd.match_desc(kp2, state["kp2"], d2, state["d2"], tm)
    assert np.array_equal(m11, oracle.match_desc(kp1, state["kp1"], d1, state["d1"], tm))
    assert np.array_equal(m22, oracle.match_desc(kp2, state["kp2"], d2, state["d2"], tm))
    # and the loop goes on with integer images
    state = None
    for t in range(6, 12):
        state = _loop_frame(seq, t, st, tm, state)
        if t > 6:
            want = oracle.match_desc(state["kp1"], seq["kp"][t - 1, 0, :seq["n"][t - 1, 0]], state["d1"], seq["desc"][t - 1, 0, :seq["n"][t - 1, 0]], tm)
            assert np.array_equal(state["m11"], want)


def test_a_stereo_pair_that_is_not_rectified_where_rectified_ones_were_expected(viso, oracle, seq):
    """Rectified pairs for a while (the launches leave the wide-band stereo kernel out), then a fundamental matrix whose
    epipolar band match_stereo_kernel declines: the call is repeated with the kernel and gives the oracle's matches."""
    drop_in.plain_cache(True)
    drop_in.plain_speculate(True)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    state = None
    for t in range(5):
        state = _loop_frame(seq, t, st, tm, state)
    before = drop_in.plain_stats()["general_reruns"]
    R, _ = synth.rot_from_tr(np.r_[0.2, -0.25, 0.3, 0, 0, 0])
    P2 = synth.KITTI_P1[:, :3] @ np.c_[R, np.array([-0.54, 0.03, 0.02])]
    wide = MatchParams.stereo(oracle.F_from_P(synth.KITTI_P1, P2))
    for t in (5, 6):
        nL, nR = seq["n"][t]
        kp1, kp2 = seq["kp"][t, 0, :nL], seq["kp"][t, 1, :nR]
        d1, d2 = seq["desc"][t, 0, :nL], seq["desc"][t, 1, :nR]
        lr = libviso_amd.match_desc(kp1, kp2, d1, d2, wide)
        assert np.array_equal(lr, oracle.match_desc(kp1, kp2, d1, d2, wide))
        # the first surprises the launcher, the second does not (the kernel is back in the launch)
        assert drop_in.plain_stats()["general_reruns"] == before + 1
    # and the loop goes on with rectified pairs
    state = None
    for t in range(7, 13):
        state = _loop_frame(seq, t, st, tm, state)
        nL, nR = seq["n"][t]
        want = oracle.match_desc(seq["kp"][t, 0, :nL], seq["kp"][t, 1, :nR], seq["desc"][t, 0, :nL], seq["desc"][t, 1, :nR], st)
        assert np.array_equal(state["lr"], want)


def test_an_image_against_itself_and_against_a_copy_of_itself(viso, oracle, seq):
    """Both sides of a call the same arrays (one upload serves both), then the same bytes in other arrays (found resident),
    with and without the image cache: the oracle's matches every time."""
    tm = MatchParams.temporal()
    for cache in (True, False):
        drop_in.plain_cache(cache)
        for t in (2, 3):
            n = seq["n"][t, 0]
            kp, d = np.ascontiguousarray(seq["kp"][t, 0, :n]), np.ascontiguousarray(seq["desc"][t, 0, :n])
            want = oracle.match_desc(kp, kp, d, d, tm)
            assert np.array_equal(libviso_amd.match_desc(kp, kp, d, d, tm), want)
            assert np.array_equal(libviso_amd.match_desc(kp, kp.copy(), d, d.copy(), tm), want)
            assert np.array_equal(libviso_amd.match_desc(kp.copy(), kp, d.copy(), d, tm), want)


def test_frames_of_changing_size_and_empty_images(viso, oracle):
    """Keypoint counts that change from frame to frame (the frame's blocks are re-laid out), an image without keypoints."""
    s = synth.make_sequence(3, 9, n_kp=300, width=500, height=200, ragged=True)
    s["n"][4, 1] = 0                                              # frame 4: no keypoints in the right image
    s["n"][6, 0] = 40
    st, tm = MatchParams.stereo(s["F"]), MatchParams.temporal()
    o = drop_in.run(s["kp"], s["desc"], s["n"], s["F"], s["param"], seed=2, want_matches=True)
    want = oracle.sequence(s["kp"], s["desc"], s["n"], st, tm, s["param"], seed=2)
    assert np.array_equal(o["ok"], want["ok"]) and np.array_equal(o["n_inl"], want["n_inl"])
    for t in range(9):
        nL, nR = s["n"][t]
        assert np.array_equal(o["matches"][0][t], oracle.match_desc(s["kp"][t, 0, :nL], s["kp"][t, 1, :nR], s["desc"][t, 0, :nL], s["desc"][t, 1, :nR], st))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_shuffled_and_perturbed_call_sequences_always_get_their_own_results(viso, oracle, seed):
    """The frame logic guesses which call comes next; this feeds it sequences that are ALMOST the loop's -- calls left out,
    repeated, out of order, with perturbed arguments, scene cuts, frames of other sizes -- and checks every single result
    against the oracle.  Whatever was computed ahead may only be handed out for byte-identical arguments."""
    drop_in.plain_cache(True)
    drop_in.plain_speculate(True)
    rng = np.random.default_rng(seed)
    seqs = [synth.make_sequence(40 + seed, 26, n_kp=360, width=500, height=180, ragged=True),
            synth.make_sequence(90 + seed, 26, n_kp=300, width=500, height=180)]
    F, prm = seqs[0]["F"], seqs[0]["param"]
    st, tm = MatchParams.stereo(F), MatchParams.temporal()
    prev = None
    which = 0
    for t in range(26):
        if rng.random() < 0.12:
            which ^= 1                                         # scene cut: the other sequence's frame t
        s = seqs[which]
        nL, nR = s["n"][t]
        kp1, kp2 = s["kp"][t, 0, :nL].copy(), s["kp"][t, 1, :nR].copy()
        d1, d2 = s["desc"][t, 0, :nL].copy(), s["desc"][t, 1, :nR].copy()
        if rng.random() < 0.1 and prev is not None:            # a temporal call BEFORE the frame's stereo call
            got = libviso_amd.match_desc(kp1, prev["kp1"], d1, prev["d1"], tm)
            assert np.array_equal(got, oracle.match_desc(kp1, prev["kp1"], d1, prev["d1"], tm))
        lr = libviso_amd.match_desc(kp1, kp2, d1, d2, st)
        assert np.array_equal(lr, oracle.match_desc(kp1, kp2, d1, d2, st)), t
        cur = {"kp1": kp1, "kp2": kp2, "d1": d1, "d2": d2, "lr": lr}
        lr_used = lr
        if rng.random() < 0.2 and len(lr) > 4:                 # collect on a shortened list
            lr_used = lr[:-2].copy()
        if rng.random() < 0.85:
            x = libviso_amd.collect_matches(kp1, kp2, lr_used)
            assert np.array_equal(x, oracle.collect_matches(kp1, kp2, lr_used)), t
        else:
            x = oracle.collect_matches(kp1, kp2, lr_used)       # the call left out
        p = prm
        if rng.random() < 0.15:
            p = type(prm).from_buffer_copy(prm); p.f = prm.f * 1.01
        X = libviso_amd.triangulate_rectified(x, p)
        assert np.array_equal(X, oracle.triangulate_rectified(x, p)), t
        if rng.random() < 0.2:                                  # and once more
            assert np.array_equal(libviso_amd.triangulate_rectified(x, prm), oracle.triangulate_rectified(x, prm)), t
        cur["X"] = oracle.triangulate_rectified(oracle.collect_matches(kp1, kp2, lr), prm)
        if prev is not None:
            order = [0, 1] if rng.random() < 0.8 else [1, 0]     # right before left
            res = {}
            for w in order:
                a, b = (("kp1", "d1") if w == 0 else ("kp2", "d2"))
                q_kp, q_d = cur[a], cur[b]
                if rng.random() < 0.1:
                    q_d = q_d.copy(); q_d[rng.integers(len(q_d)), rng.integers(121)] += 2
                res[w] = libviso_amd.match_desc(q_kp, prev[a], q_d, prev[b], tm)
                assert np.array_equal(res[w], oracle.match_desc(q_kp, prev[a], q_d, prev[b], tm)), (t, w)
            m11, m22 = res[0], res[1]
            lrp = prev["lr"]
            if rng.random() < 0.15 and len(lrp) > 3:
                lrp = lrp[1:].copy()
            _, circ, pcl, n = libviso_amd.match_circle(lr, lrp, m11, m22)
            _, c_o, p_o, n_o = oracle.match_circle(lr, lrp, m11, m22)
            assert n == n_o and np.array_equal(circ, c_o) and np.array_equal(pcl, p_o), t
            if n >= 3 and lrp is prev["lr"] and lr_used is lr:
                x_full = oracle.collect_matches(kp1, kp2, lr)
                x_c, Xp_c = np.ascontiguousarray(x_full[:, pcl[:, 0]]), np.ascontiguousarray(prev["X"][:, pcl[:, 1]])
                frame = t if rng.random() < 0.85 else t + 7
                r_a, tr_a, inl_a = libviso_amd.ransac_minimize_reproj(Xp_c, x_c, prm, seed=9, frame=frame)
                r_o, tr_o, inl_o = oracle.ransac_minimize_reproj(Xp_c, x_c, prm, seed=9, frame=frame)
                assert r_a == r_o and np.array_equal(inl_a, inl_o), t
                if r_o:
                    assert rel_fro(libviso_amd.tr2mat(tr_a), oracle.tr2mat(tr_o)) < POSE_TOL
        prev = cur
    st_ = drop_in.plain_stats()
    assert st_["served"][0] + st_["served"][1] > 0              # the frame logic did take part
