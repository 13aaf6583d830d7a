"""BASELINE.json sizes: configs[1]/[2] (1241x376, 2000 kp) and configs[4]
(2048x1024, 8000 kp) against the oracle on a few frames, plus
size-independent properties over a whole bench-sized batch."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu


def _run_batch(seq, seed=3, full=True, temporal_k=None):
    nf, _, cap, _ = seq["kp"].shape
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    if temporal_k is not None:
        tm.max_neighbors = temporal_k
    ctx = libviso_amd.Context(0)
    if _VARIANT[0] is not None:
        libviso_amd.set_matcher_variant(_VARIANT[0], ctx)
    b = libviso_amd.Batch(ctx, nf, cap)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.set_params(st, tm, seq["param"], seed=seed)
    (b.run if full else b.run_matcher)()
    return ctx, b, st, tm


_VARIANT = [None]   # matcher variant of the contexts _run_batch creates (None: the build's default)


def _per_call(oracle, seq, which, t, st, tm):
    n = seq["n"]
    q = (0, t) if which < 2 else (1, t)
    tg = (1, t) if which == 0 else ((0, t - 1) if which == 1 else (1, t - 1))
    nq, nt = n[q[1], q[0]], n[tg[1], tg[0]]
    return oracle.match_desc(seq["kp"][q[1], q[0], :nq], seq["kp"][tg[1], tg[0], :nt],
                             seq["desc"][q[1], q[0], :nq], seq["desc"][tg[1], tg[0], :nt],
                             st if which == 0 else tm, return_scored=True)


def test_config2_3_full_size_vs_oracle(viso, oracle):
    seq = synth.make_sequence(101, 3, n_kp=2000)              # 1241x376, 2000 features/image
    ctx, b, st, tm = _run_batch(seq)
    sc, mo = b.counters()
    for t in range(3):
        for which in range(3 if t else 1):
            want, wsc = _per_call(oracle, seq, which, t, st, tm)
            assert np.array_equal(b.matches(which, t), want) and sc[which, t] == wsc
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=3)
    tr, ok, n_inl = b.poses()
    assert np.array_equal(ok, want["ok"]) and np.array_equal(n_inl, want["n_inl"]) and ok[1:].all()
    for t in (1, 2):
        a, r = libviso_amd.tr2mat(tr[t]), oracle.tr2mat(want["tr"][t])
        assert np.linalg.norm(a - r) / np.linalg.norm(r) < 1e-5
    # ~50 candidates per temporal query, ~3 per stereo query
    assert 35 < sc[1, 1] / 2000 < 65 and 1.5 < sc[0, 1] / 2000 < 5
    b.close(); ctx.close()


@pytest.mark.parametrize("k", [45, 60, 93])
def test_k_cap_between_a_querys_count_and_the_unions_length(viso, oracle, k):
    """max_neighbors below the length of a round's union list (~94 rows) but around / above a single query's ~50
    in-radius targets: the temporal kernel cannot tell from the list length that nobody exceeds K and takes the per-query
    counts — some queries leave for the overflow kernel (K-cap selection), the others stand and must add exactly their
    own candidates to the scored-pair counter."""
    seq = synth.make_sequence(107, 2, n_kp=2000)
    ctx, b, st, tm = _run_batch(seq, full=False, temporal_k=k)
    sc, _ = b.counters()
    kq, kt = seq["kp"][1, 0, :seq["n"][1, 0]], seq["kp"][0, 0, :seq["n"][0, 0]]
    cnt = (np.abs(kq[:, None, :] - kt[None, :, :]).sum(2) <= 80).sum(1)
    assert (cnt <= k).any() and ((cnt > k).any() or k == 93)   # both kinds of query exist (K = 93: nobody exceeds it,
                                                               # but many a union list is longer)
    for which in (1, 2):
        want, wsc = _per_call(oracle, seq, which, 1, st, tm)
        assert np.array_equal(b.matches(which, 1), want) and sc[which, 1] == wsc, (which, k)
    b.close(); ctx.close()


def test_clustered_keypoints_vs_oracle(viso, oracle):
    """2000 features per image of which 70 % sit in a few blobs (what a corner detector finds on real images): many
    queries have more than K in-radius candidates, windows outgrow the LDS staging, union lists overflow — the
    overflow kernel and the K-cap selection carry a real share of the work.  Everything still equals the oracle."""
    seq = synth.make_sequence(106, 3, n_kp=2000, cluster_frac=0.7)
    ctx, b, st, tm = _run_batch(seq)
    sc, _ = b.counters()
    over_k = 0
    for t in range(3):
        for which in range(3 if t else 1):
            want, wsc = _per_call(oracle, seq, which, t, st, tm)
            assert np.array_equal(b.matches(which, t), want) and sc[which, t] == wsc, (which, t)
    # the data really is dense: some query has more in-radius targets than the temporal K (250)
    kp = seq["kp"][1, 0, :seq["n"][1, 0]]
    kq = seq["kp"][0, 0, :seq["n"][0, 0]]
    d = np.abs(kp[:, None, :] - kq[None, :, :]).sum(2)
    over_k = int(((d <= 80).sum(1) > 250).sum())
    assert over_k > 20
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=3)
    tr, ok, n_inl = b.poses()
    assert np.array_equal(ok, want["ok"]) and np.array_equal(n_inl, want["n_inl"])
    b.close(); ctx.close()


def test_config5_stress_size_vs_oracle(viso, oracle):
    """configs[4] geometry (2048x1024, 8000 features/image), whole pipeline: matcher, circle join, RANSAC/GN."""
    seq = synth.make_sequence(102, 3, n_kp=8000, width=2048, height=1024)
    ctx, b, st, tm = _run_batch(seq, full=True)
    sc, _ = b.counters()
    for which, t in ((0, 0), (0, 1), (1, 1), (2, 1), (0, 2), (1, 2), (2, 2)):
        want, wsc = _per_call(oracle, seq, which, t, st, tm)
        assert np.array_equal(b.matches(which, t), want) and sc[which, t] == wsc
    want = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=3)
    tr, ok, n_inl = b.poses()
    assert np.array_equal(ok, want["ok"]) and np.array_equal(n_inl, want["n_inl"]) and ok[1:].all()
    for t in (1, 2):
        circ, pcl = b.circle(t)       # match_circle (:207-243) over the four (already verified) match lists
        _, wc, wp, _ = oracle.match_circle(b.matches(0, t), b.matches(0, t - 1), b.matches(1, t), b.matches(2, t))
        assert len(circ) > 100 and np.array_equal(circ, wc) and np.array_equal(pcl, wp)
        a, r = libviso_amd.tr2mat(tr[t]), oracle.tr2mat(want["tr"][t])
        assert np.linalg.norm(a - r) / np.linalg.norm(r) < 1e-5
    b.close(); ctx.close()


def test_bench_sized_batch_properties(viso):
    """64 frame pairs of the bench workload: properties that need no oracle."""
    seq = synth.make_sequence(103, 65, n_kp=2000)
    ctx, b, st, tm = _run_batch(seq, full=True)
    tr1, ok1, ni1 = b.poses()
    first = [b.matches(w, t) for t in (1, 17, 64) for w in range(3)]
    b.run()                                                     # idempotent / deterministic
    tr2, ok2, ni2 = b.poses()
    assert np.array_equal(tr1, tr2) and np.array_equal(ok1, ok2) and np.array_equal(ni1, ni2)
    again = [b.matches(w, t) for t in (1, 17, 64) for w in range(3)]
    assert all(np.array_equal(x, y) for x, y in zip(first, again))
    assert ok1[1:].all() and np.abs(tr1[1:] - seq["tr_gt"][1:]).max() < 3e-2
    kp, desc, n = seq["kp"], seq["desc"], seq["n"]
    for t in (1, 17, 64):
        for which in range(3):
            m = b.matches(which, t)
            q = (0, t) if which < 2 else (1, t)
            tg = (1, t) if which == 0 else ((0, t - 1) if which == 1 else (1, t - 1))
            # sorted by (dist, i1); one match per query; indices in range; target 0 never matched (Q1)
            key = m[:, 2].astype(np.int64) * 10000 + m[:, 0]
            assert np.all(np.diff(key) > 0) and len(np.unique(m[:, 0])) == len(m)
            assert m[:, 0].max() < n[q[1], q[0]] and 0 < m[:, 1].min() and m[:, 1].max() < n[tg[1], tg[0]]
            # reported distance is the SAD of the reported pair, within the L1 radius
            dq, dt = desc[q[1], q[0]][m[:, 0]], desc[tg[1], tg[0]][m[:, 1]]
            assert np.array_equal(np.abs(dq - dt).sum(1).astype(np.int32), m[:, 2])
            kq, kt = kp[q[1], q[0]][m[:, 0]], kp[tg[1], tg[0]][m[:, 1]]
            assert np.abs(kq - kt).sum(1).max() <= 80
            if which == 0:
                assert np.abs(kq[:, 1] - kt[:, 1]).max() <= 1      # rectified epipolar gate == |dy| <= 1
    b.close(); ctx.close()


@pytest.mark.parametrize("variant", libviso_amd.MATCHER_VARIANTS)
def test_both_matcher_kernels_agree_with_oracle(viso, oracle, variant):
    """Every matcher variant of the build must be bit-exact: ragged counts, duplicated patches (exact SAD ties ->
    overflow kernel), dense clusters (K cap), and the 8000-keypoint window."""
    libviso_amd.set_matcher_variant(variant)        # plain family
    _VARIANT[0] = variant                           # batches: per context (see _run_batch)
    try:
        seq = synth.make_sequence(104, 4, n_kp=900, width=300, height=120, ragged=True, dup_frac=0.1)
        ctx, b, st, tm = _run_batch(seq, full=False)
        sc, _ = b.counters()
        for t in range(4):
            for which in range(3 if t else 1):
                want, wsc = _per_call(oracle, seq, which, t, st, tm)     # 900 kp in 300x120: > K in radius
                assert np.array_equal(b.matches(which, t), want) and sc[which, t] == wsc, (variant, which, t)
        b.close(); ctx.close()
        seq = synth.make_sequence(105, 2, n_kp=8000, width=2048, height=1024)
        ctx, b, st, tm = _run_batch(seq, full=False)
        for which, t in ((0, 1), (1, 1), (2, 1)):
            want, _ = _per_call(oracle, seq, which, t, st, tm)
            assert np.array_equal(b.matches(which, t), want), (variant, which, t)
        b.close(); ctx.close()
    finally:
        _VARIANT[0] = None
        libviso_amd.set_matcher_variant(libviso_amd.DEFAULT_MATCHER)
