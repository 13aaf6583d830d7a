"""Image-in mode: descriptors extracted on the device
from uint8 images straight into the matcher's row format must give exactly the
pipeline results the oracle gets from its own extractor
(MyFeatureExtractor::computeImpl, reference src/viso.cpp:1004-1024)."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu


def _oracle_desc(oracle, seq):
    nf, _, cap, _ = seq["kp"].shape
    desc = np.zeros((nf, 2, cap, 121), np.float32)
    for t in range(nf):
        for side in range(2):
            k = seq["n"][t, side]
            desc[t, side, :k] = oracle.extract_descriptors(seq["images"][t, side], seq["kp"][t, side, :k])
    return desc


def test_batch_from_images_equals_oracle(viso, oracle):
    seq = synth.make_image_sequence(7, 5, n_kp=700, width=640, height=200)
    seq["kp"][1, 0, :3] = [[0, 0], [639, 199], [2, 1]]          # border windows (strict > 0 rule, reflect-101)
    desc = _oracle_desc(oracle, seq)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    want = oracle.sequence(seq["kp"], desc, seq["n"], st, tm, seq["param"], seed=2)
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, 5, 700)
    b.upload_images(seq["images"], seq["kp"], seq["n"])
    b.set_params(st, tm, seq["param"], seed=2)
    b.run_images()
    tr, ok, n_inl = b.poses()
    sc, mo = b.counters()
    assert np.array_equal(ok, want["ok"]) and np.array_equal(n_inl, want["n_inl"]) and ok[1:].all()
    assert np.array_equal(sc, want["scored"]) and np.array_equal(mo, want["m_out"])
    for t in range(1, 5):
        a, r = libviso_amd.tr2mat(tr[t]), oracle.tr2mat(want["tr"][t])
        assert np.linalg.norm(a - r) / np.linalg.norm(r) < 1e-5
        assert np.abs(tr[t] - seq["tr_gt"][t]).max() < 3e-2
    # match lists of one frame against per-call oracle on the extracted descriptors
    n = seq["n"]
    m = oracle.match_desc(seq["kp"][2, 0, :n[2, 0]], seq["kp"][1, 0, :n[1, 0]], desc[2, 0, :n[2, 0]], desc[1, 0, :n[1, 0]], tm)
    assert np.array_equal(b.matches(1, 2), m)
    # the feature-in path on the same (host-extracted) descriptors agrees too
    b2 = libviso_amd.Batch(ctx, 5, 700)
    b2.upload(seq["kp"], desc, seq["n"]); b2.set_params(st, tm, seq["param"], seed=2); b2.run()
    tr2, ok2, ni2 = b2.poses()
    assert np.array_equal(tr, tr2) and np.array_equal(ok, ok2) and np.array_equal(n_inl, ni2)
    b.close(); b2.close(); ctx.close()
