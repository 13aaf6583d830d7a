"""viso_batch_upload_i16*: the descriptors as int16 rows (the lossless encoding of the reference's CV_32F Sobel windows,
src/viso.cpp:995-1024) must give exactly what the f32 uploads of the same values give, through the synchronous and the
asynchronous (pinned, stream ordered) entry points, for ragged keypoint counts and a partial frame range."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu


def _results(b, nf):
    out = [b.poses()]
    for t in range(nf):
        for which in range(3):
            if which == 0 or t > 0:
                out.append(b.matches(which, t))
    out.append(b.counters())
    return out


def _same(a, b):
    for x, y in zip(a, b):
        if isinstance(x, tuple):
            assert all(np.array_equal(u, v) for u, v in zip(x, y))
        else:
            assert np.array_equal(x, y)


def test_i16_rows_give_the_f32_results(viso, oracle):
    nf = 7
    seq = synth.make_sequence(91, nf, n_kp=700, width=800, height=300, ragged=True, dup_frac=0.05)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    d16 = seq["desc"].astype(np.int16)
    assert np.array_equal(d16.astype(np.float32), seq["desc"])
    ctx = libviso_amd.Context(0)
    cap = seq["kp"].shape[2]
    b = libviso_amd.Batch(ctx, nf, cap)
    b.set_params(st, tm, seq["param"], seed=3)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.run()
    want = _results(b, nf)
    # and those are the oracle's
    ref = oracle.sequence(seq["kp"], seq["desc"], seq["n"], st, tm, seq["param"], seed=3)
    assert np.array_equal(want[0][1], ref["ok"]) and np.array_equal(want[0][2], ref["n_inl"])
    # synchronous int16 upload
    b.upload_i16(seq["kp"], d16, seq["n"])
    b.run()
    _same(_results(b, nf), want)
    assert not b.general_path_flags().any()
    # asynchronous, from pinned memory, in two pieces (frames 0..2 then 3..6)
    pk = libviso_amd.PinnedArray(seq["kp"].shape, np.float32)
    pd = libviso_amd.PinnedArray(d16.shape, np.int16)
    pk.a[...] = seq["kp"]; pd.a[...] = d16
    b.upload(seq["kp"][::-1].copy(), seq["desc"][::-1].copy(), seq["n"][::-1].copy())     # other data in between
    b.upload_i16(pk.a[:3], pd.a[:3], seq["n"][:3], f0=0, asynchronous=True)
    b.upload_i16(pk.a[3:], pd.a[3:], seq["n"][3:], f0=3, asynchronous=True)
    b.run()
    _same(_results(b, nf), want)
    # back to f32: the pack kernel follows the last upload
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.run()
    _same(_results(b, nf), want)
    pk.close(); pd.close(); b.close(); ctx.close()


def test_mixed_families_in_one_run_are_refused(viso):
    """int16 rows live in the f32 rows' device buffer; a run over frames of both families would reinterpret
    one of them silently.  The batch remembers the family per frame and viso_batch_run* refuses the mix."""
    nf = 5
    seq = synth.make_sequence(92, nf, n_kp=300, width=600, height=200)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    d16 = seq["desc"].astype(np.int16)
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, seq["kp"].shape[2])
    b.set_params(st, tm, seq["param"], seed=3)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.run()
    want = b.poses()
    b.upload_i16(seq["kp"][2:4], d16[2:4], seq["n"][2:4], f0=2)          # frames 2..3 as int16, the rest still f32
    with pytest.raises(RuntimeError, match="both"):
        b.run()
    with pytest.raises(RuntimeError, match="both"):
        b.run_matcher()
    b.upload_i16(seq["kp"], d16, seq["n"])                               # all frames through one family again
    b.run()
    assert all(np.array_equal(x, y) for x, y in zip(b.poses(), want))
    # the hypotheses getter sizes its arrays from set_params' ransac_iter and refuses another count
    tr_h, ok_h, cnt_h, _ = b.hypotheses()
    assert tr_h.shape == (nf, seq["param"].ransac_iter, 6)
    with pytest.raises(ValueError):
        b.hypotheses(seq["param"].ransac_iter - 10)
    b.close(); ctx.close()


def test_i16_extreme_values(viso, oracle):
    """The whole int16 range (beyond what Sobel of uint8 produces): still exact, never the general path."""
    rng = np.random.default_rng(5)
    nf, cap = 3, 256
    seq = synth.make_sequence(7, nf, n_kp=cap, width=300, height=120)
    d16 = rng.integers(-32768, 32768, seq["desc"].shape).astype(np.int16)
    d16[1:] = np.clip(d16[:-1].astype(np.int32) + rng.integers(-3, 4, d16[1:].shape), -32768, 32767).astype(np.int16)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, cap)
    b.set_params(st, tm, seq["param"], seed=1)
    b.upload_i16(seq["kp"], d16, seq["n"])
    b.run_matcher()
    for t in range(1, nf):
        for which, (qs, qt, ts, tt) in ((1, (0, t, 0, t - 1)), (2, (1, t, 1, t - 1))):
            n1, n2 = seq["n"][qt, qs], seq["n"][tt, ts]
            want = oracle.match_desc(seq["kp"][qt, qs, :n1], seq["kp"][tt, ts, :n2], d16[qt, qs, :n1].astype(np.float32),
                                     d16[tt, ts, :n2].astype(np.float32), tm)
            assert np.array_equal(b.matches(which, t), want), (which, t)
    assert not b.general_path_flags().any()
    b.close(); ctx.close()
