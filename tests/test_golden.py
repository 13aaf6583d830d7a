"""Committed golden vectors (tests/golden/*.npz, made by make_golden.py): the
oracle must keep reproducing them (CPU suite) and the HIP path must reproduce
them bit-exactly / within 1e-5 (GPU suite)."""
import os

import numpy as np
import pytest

from libviso_amd.abi import MatchParams, Param

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _param(c):
    return Param.default(base=float(c[0]), f=float(c[1]), cu=float(c[2]), cv=float(c[3]))


def _check_matcher(impl):
    g = np.load(os.path.join(G, "matcher.npz"))
    f = lambda k: g[k].astype(np.float32)
    st, tm = MatchParams.stereo(g["F"]), MatchParams.temporal()
    assert np.array_equal(impl.match_desc(f("kL1"), f("kR1"), f("dL1"), f("dR1"), st), g["m_stereo"])
    assert np.array_equal(impl.match_desc(f("kL1"), f("kL0"), f("dL1"), f("dL0"), tm), g["m_temporal"])
    assert len(g["m_stereo"]) > 40 and len(g["m_temporal"]) > 40


def _check_solver(impl, tol):
    g = np.load(os.path.join(G, "solver.npz"))
    p = _param(g["calib"])
    ok, tr, inl = impl.ransac_minimize_reproj(g["X"], g["obs"], p, samples=g["samples"])
    assert ok == int(g["ok"]) == 1 and np.array_equal(inl, g["inl"])
    assert np.abs(tr - g["tr"]).max() <= tol * max(1.0, np.abs(g["tr"]).max())
    r = impl.minimize_reproj(g["X"], g["obs"], np.zeros(6), p, np.arange(0, 120, 3))
    assert r[0] == int(g["ok_gn"]) and np.abs(r[1] - g["tr_gn"]).max() <= tol


def test_oracle_reproduces_golden(oracle):
    _check_matcher(oracle)
    _check_solver(oracle, 0.0)
    g = np.load(os.path.join(G, "sequence.npz"))
    st, tm = MatchParams.stereo(g["F"]), MatchParams.temporal()
    out = oracle.sequence(g["kp"], g["desc"].astype(np.float32), g["n"], st, tm, _param(g["calib"]), seed=4, first_frame=10)
    assert np.array_equal(out["tr"], g["tr"]) and np.array_equal(out["ok"], g["ok"])
    assert np.array_equal(out["scored"], g["scored"]) and np.array_equal(out["m_out"], g["m_out"])


@pytest.mark.gpu
def test_hip_reproduces_golden(viso):
    import libviso_amd
    _check_matcher(libviso_amd)
    _check_solver(libviso_amd, 1e-7)
    g = np.load(os.path.join(G, "sequence.npz"))
    st, tm = MatchParams.stereo(g["F"]), MatchParams.temporal()
    nf, _, cap, _ = g["kp"].shape
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, cap)
    b.upload(g["kp"], g["desc"].astype(np.float32), g["n"])
    b.set_params(st, tm, _param(g["calib"]), seed=4, first_frame=10)
    b.run()
    tr, ok, n_inl = b.poses()
    sc, mo = b.counters()
    assert np.array_equal(ok, g["ok"]) and np.array_equal(n_inl, g["n_inl"])
    assert np.array_equal(sc, g["scored"]) and np.array_equal(mo, g["m_out"])
    for t in range(1, nf):
        a, r = libviso_amd.tr2mat(tr[t]), libviso_amd.tr2mat(g["tr"][t])
        assert np.linalg.norm(a - r) / np.linalg.norm(r) < 1e-5
    b.close(); ctx.close()
