"""match_union8_kernel (matcher variant 6): candidates ranked on the rows' 8-bit planes, the two best scored exactly, the
third key's lower bound deciding whether that settles match_desc (src/viso.cpp:703-716).  The ordinary suites run every
variant on ordinary data; here the data sit ON the verdict's edges: SADs graded around the 896-unit slack of the bound,
second / third candidates close enough to force the overflow path, values outside the plane's range (clamped), the
ratio test on and off — matches and the scored-pair counters must equal the oracle's whichever path a query takes."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu

V8 = 6


def _spread(rng, total, dlen=121, step=None):
    """An integer vector of `dlen` entries whose absolute values sum to `total` (random signs, random positions)."""
    v = np.zeros(dlen, np.int64)
    left = int(total)
    while left > 0:
        s = min(left, int(rng.integers(1, (step or 40) + 1)))
        v[rng.integers(0, dlen)] += s * (1 if rng.random() < 0.5 else -1)
        left -= s
    return v


def _graded_case(rng, n1, n2, vrange, second, ratio):
    """Every target is within the radius of every query; query i has planted candidates at chosen SADs d1 <= d2 <= d3 ..."""
    kp1 = rng.integers(0, 30, (n1, 2)).astype(np.float32)
    kp2 = rng.integers(0, 30, (n2, 2)).astype(np.float32)
    d1 = rng.integers(-vrange, vrange + 1, (n1, 121)).astype(np.int64)
    d2 = rng.integers(-vrange, vrange + 1, (n2, 121)).astype(np.int64)
    gaps = np.array([0, 0, 1, 2, 7, 60, 300, 700, 840, 847, 890, 895, 896, 897, 905, 1000, 1790, 1800, 2500, 6000])
    free = list(rng.permutation(np.arange(1, n2)))   # target 0 stays random (Q1: its distance cuts the list)
    for i in range(n1):
        k = int(rng.integers(0, 6))
        if len(free) < k:
            break
        d = int(rng.integers(0, 5000))
        for _ in range(k):
            t = free.pop()
            # perturbations that cancel in SAD terms would change the planted distance: each one touches its own entries
            d2[t] = d1[i] + _spread(rng, d)
            d += int(rng.choice(gaps))
    lim = 32767
    d1, d2 = np.clip(d1, -lim, lim), np.clip(d2, -lim, lim)
    mp = MatchParams.temporal()
    mp.enforce_2nd_best = int(second)
    mp.ratio_2nd_best = float(ratio)
    mp.radius = 200.0
    return kp1, kp2, d1.astype(np.float32), d2.astype(np.float32), mp


@pytest.mark.parametrize("shift", [3, 2, 1, 0])
@pytest.mark.parametrize("vrange", [1020, 40, 6000])
def test_graded_sads_around_the_bounds_slack(viso, oracle, vrange, shift):
    """Every shift of the planes must give the oracle's matches on every data range: a shift too small for the data clamps
    (the bound only gets looser), a shift too large leaves a slack wider than the SADs."""
    if V8 not in libviso_amd.MATCHER_VARIANTS:
        pytest.skip("this build has no variant 6")
    rng = np.random.default_rng(8800 + vrange)
    libviso_amd.set_matcher_variant(V8)
    libviso_amd.set_row8_shift(shift)
    try:
        n_acc = 0
        for it in range(40):
            second = it % 4 != 3
            ratio = [0.9, 0.8, 1.0, 0.5, 0.9, 1.5][it % 6]
            kp1, kp2, d1, d2, mp = _graded_case(rng, int(rng.integers(8, 70)), int(rng.integers(4, 240)), vrange, second, ratio)
            want = oracle.match_desc(kp1, kp2, d1, d2, mp)
            got = libviso_amd.match_desc(kp1, kp2, d1, d2, mp)
            assert np.array_equal(got, want), (vrange, it, len(kp1), len(kp2), second, ratio)
            n_acc += len(want)
        assert n_acc > 100
    finally:
        libviso_amd.set_matcher_variant(libviso_amd.DEFAULT_MATCHER)
        libviso_amd.set_row8_shift(-1)


def _batch(seq, variant, tm=None):
    st = MatchParams.stereo(seq["F"])
    tm = tm or MatchParams.temporal()
    ctx = libviso_amd.Context(0)
    libviso_amd.set_matcher_variant(variant, ctx)
    nf, _, cap, _ = seq["kp"].shape
    b = libviso_amd.Batch(ctx, nf, cap)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.set_params(st, tm, seq["param"], seed=5)
    b.run_matcher()
    return ctx, b, st, tm


def _oracle_call(oracle, seq, which, t, st, tm):
    n = seq["n"]
    q = (0, t) if which < 2 else (1, t)
    tg = (1, t) if which == 0 else ((0, t - 1) if which == 1 else (1, t - 1))
    nq, nt = n[q[1], q[0]], n[tg[1], tg[0]]
    return oracle.match_desc(seq["kp"][q[1], q[0], :nq], seq["kp"][tg[1], tg[0], :nt],
                             seq["desc"][q[1], q[0], :nq], seq["desc"][tg[1], tg[0], :nt],
                             st if which == 0 else tm, return_scored=True)


@pytest.mark.parametrize("scale", [1.0, 0.04])
def test_counters_and_matches_whichever_path_a_query_takes(viso, oracle, scale):
    """Image-sized frames through the batch pipeline.  scale 1: the bench's data, nearly every query is settled by its two
    exact SADs.  scale 0.04: descriptors squeezed into +-40, every SAD of a query within the slack of the others — the
    third key never clears the bound and the kernel's rescue loop has to score most members of (nearly) every query
    exactly (or, where the round's list is longer than its SAD8 store, leaves the query to the overflow kernel): same
    matches, same counters."""
    if V8 not in libviso_amd.MATCHER_VARIANTS:
        pytest.skip("this build has no variant 6")
    seq = synth.make_sequence(611, 3, n_kp=1500, width=900, height=300, ragged=True, dup_frac=0.03)
    seq["desc"] = np.rint(seq["desc"] * scale).astype(np.float32)
    ctx, b, st, tm = _batch(seq, V8)
    sc, _ = b.counters()
    novf = b.overflow_count()
    for t in range(3):
        for which in range(3 if t else 1):
            want, wsc = _oracle_call(oracle, seq, which, t, st, tm)
            assert np.array_equal(b.matches(which, t), want) and sc[which, t] == wsc, (scale, which, t)
    n_temporal = int(seq["n"][1:].sum())
    if scale == 1.0:
        assert novf < n_temporal // 10, (novf, n_temporal)
    b.close(); ctx.close()


def test_same_results_as_the_u16_kernel_int16_rows_and_images(viso):
    """The other two producers of the 8-bit planes (pack_desc_i16_kernel, extract_pack_kernel) against match_union_kernel."""
    if V8 not in libviso_amd.MATCHER_VARIANTS:
        pytest.skip("this build has no variant 6")
    seq = synth.make_sequence(612, 3, n_kp=1200, width=800, height=300)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    out = {}
    for v in (3, V8):
        ctx = libviso_amd.Context(0)
        libviso_amd.set_matcher_variant(v, ctx)
        b = libviso_amd.Batch(ctx, 3, 1200)
        b.upload_i16(seq["kp"], seq["desc"].astype(np.int16), seq["n"])
        b.set_params(st, tm, seq["param"], seed=5)
        b.run_matcher()
        out[v] = ([b.matches(w, t) for t in range(3) for w in range(3 if t else 1)], b.counters()[0].copy())
        b.close(); ctx.close()
    assert all(np.array_equal(a, c) for a, c in zip(out[3][0], out[V8][0])) and np.array_equal(out[3][1], out[V8][1])
    img = synth.make_image_sequence(613, 3, n_kp=900, width=640, height=240)
    out = {}
    for v in (3, V8):
        ctx = libviso_amd.Context(0)
        libviso_amd.set_matcher_variant(v, ctx)
        b = libviso_amd.Batch(ctx, 3, 900)
        b.upload_images(img["images"], img["kp"], img["n"])
        b.set_params(MatchParams.stereo(img["F"]), MatchParams.temporal(), img["param"], seed=5)
        b.run_images(matcher_only=True)
        out[v] = ([b.matches(w, t) for t in range(3) for w in range(3 if t else 1)], b.counters()[0].copy())
        b.close(); ctx.close()
    assert all(np.array_equal(a, c) for a, c in zip(out[3][0], out[V8][0])) and np.array_equal(out[3][1], out[V8][1])
    assert sum(len(m) for m in out[V8][0]) > 1000


@pytest.mark.parametrize("scale, want_shift", [(1.0, 3), (0.5, 2), (0.25, 1), (0.1, 0)])
def test_the_planes_shift_follows_the_data(viso, oracle, scale, want_shift):
    """The first run of a batch uses the default shift (3: the whole range of a Sobel of uint8), the pack kernels count
    magnitudes on the way, and the next run takes the smallest shift that clamps at most one element pair in 256: both runs
    give the oracle's matches and counters."""
    if V8 not in libviso_amd.MATCHER_VARIANTS:
        pytest.skip("this build has no variant 6")
    seq = synth.make_sequence(615, 3, n_kp=1200, width=800, height=300)
    seq["desc"] = np.rint(seq["desc"] * scale).astype(np.float32)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    ctx = libviso_amd.Context(0)
    libviso_amd.set_matcher_variant(V8, ctx)
    libviso_amd.set_row8_shift(-1, ctx)
    b = libviso_amd.Batch(ctx, 3, 1200)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.set_params(st, tm, seq["param"], seed=5)
    shifts = []
    for run in range(2):
        b.run_matcher()
        shifts.append(b.row8_shift())
        sc, _ = b.counters()
        for t in range(3):
            for which in range(3 if t else 1):
                want, wsc = _oracle_call(oracle, seq, which, t, st, tm)
                assert np.array_equal(b.matches(which, t), want) and sc[which, t] == wsc, (scale, run, which, t)
    assert shifts == [3, want_shift], shifts
    b.close(); ctx.close()
