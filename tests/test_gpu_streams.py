"""Several batches in flight on several contexts (one HIP stream each), the way bench.py and
viso::sequence_odometry drive the library: every batch must produce exactly what a lone batch produces."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu


def _run_alone(seq, st, tm, nf, kp):
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, kp)
    b.upload(seq["kp"], seq["desc"], seq["n"])
    b.set_params(st, tm, seq["param"], seed=5, first_frame=0)
    b.run()
    tr, ok, ninl = b.poses()
    m = [b.matches(w, t) for w in range(3) for t in range(nf) if not (w > 0 and t == 0)]
    b.close()
    return tr, ok, ninl, m


def test_three_batches_in_flight_match_a_lone_batch(viso):
    nf, kp = 9, 600
    seqs = [synth.make_sequence(300 + i, nf, n_kp=kp) for i in range(3)]
    st = [MatchParams.stereo(s["F"]) for s in seqs]
    tm = MatchParams.temporal()
    want = [_run_alone(seqs[i], st[i], tm, nf, kp) for i in range(3)]
    lanes = []
    for i in range(3):
        ctx = libviso_amd.Context(0)
        b = libviso_amd.Batch(ctx, nf, kp)
        b.upload(seqs[i]["kp"], seqs[i]["desc"], seqs[i]["n"])
        b.set_params(st[i], tm, seqs[i]["param"], seed=5, first_frame=0)
        lanes.append((ctx, b))
    for _ in range(4):                      # several rounds, nothing synchronised in between
        for _, b in lanes:
            b.run()
    for i, (ctx, b) in enumerate(lanes):
        tr, ok, ninl = b.poses()
        wtr, wok, wninl, wm = want[i]
        assert np.array_equal(ok, wok) and np.array_equal(ninl, wninl)
        assert np.array_equal(tr, wtr)      # same kernels, same inputs: bit identical
        m = [b.matches(w, t) for w in range(3) for t in range(nf) if not (w > 0 and t == 0)]
        assert all(np.array_equal(a, c) for a, c in zip(m, wm))
        b.close()
