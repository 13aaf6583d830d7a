"""Why tests/dropin_stress.py saw two unsolved frames report another support than the oracle after the sampler changed (round 6):\nhypothesis by hypothesis for the frames that differ; dumps their solver inputs to gpurun_out/stress_frame_<t>.npz.\n    gpurun -- 'PYTHONPATH=. python tools/experiments/stress_diag.py'"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import libviso_amd
from libviso_amd import synth, drop_in
from libviso_amd.abi import MatchParams
from oracle import pyoracle as O
rng = np.random.default_rng(5)
nf = 240
s = synth.make_sequence(321, nf, n_kp=1400, width=900, height=300, ragged=True)
for t in range(nf):
    r = rng.random()
    if r < 0.1: s["n"][t] = rng.integers(3, 60, 2)
    elif r < 0.2: s["n"][t] = rng.integers(60, 400, 2)
    elif r < 0.22: s["n"][t, rng.integers(2)] = 0
st, tm = MatchParams.stereo(s["F"]), MatchParams.temporal()
o = drop_in.run(s["kp"], s["desc"], s["n"], s["F"], s["param"], seed=4, first_frame=10)
want = O.sequence(s["kp"], s["desc"], s["n"], st, tm, s["param"], seed=4, first_frame=10)
d = np.nonzero(o["n_inl"] != want["n_inl"])[0]
print("ok equal", np.array_equal(o["ok"], want["ok"]), "frames differing in n_inl", d, "ok there", want["ok"][d], "dev", o["n_inl"][d], "oracle", want["n_inl"][d], "n_circle", o["n_circle"][d])
# the batch family on the same frames
ctx = libviso_amd.Context(0); b = libviso_amd.Batch(ctx, nf, s["kp"].shape[2])
b.upload(s["kp"], s["desc"], s["n"]); b.set_params(st, tm, s["param"], seed=4, first_frame=10); b.run()
tr, ok, n_inl = b.poses()
print("batch vs oracle n_inl differ at", np.nonzero(n_inl != want["n_inl"])[0], "batch vs dropin differ at", np.nonzero(n_inl != o["n_inl"])[0])
tr_h, ok_h, cnt_h, nu = b.hypotheses()
for t in d:
    lr, lrp = b.matches(0, t), b.matches(0, t - 1)
    circ, pcl = b.circle(t); m = len(circ)
    x = O.collect_matches(s["kp"][t, 0, :s["n"][t, 0]], s["kp"][t, 1, :s["n"][t, 1]], lr)
    xp = O.collect_matches(s["kp"][t - 1, 0, :s["n"][t - 1, 0]], s["kp"][t - 1, 1, :s["n"][t - 1, 1]], lrp)
    Xp = O.triangulate_rectified(xp, s["param"])
    obs, X = np.ascontiguousarray(x[:, pcl[:, 0]]), np.ascontiguousarray(Xp[:, pcl[:, 1]])
    S = O.ransac_samples(4, 10 + t, s["param"].ransac_iter, m)
    np.savez(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "stress_frame_%d.npz" % t), X=X, obs=obs, S=S, tr_h=tr_h[t], ok_h=ok_h[t], cnt_h=cnt_h[t])
    print("frame", t, "finite X", np.isfinite(X).all(), "finite obs", np.isfinite(obs).all(), "X range", np.nanmin(X), np.nanmax(X))
    for h in range(len(S)):
        ok0, tr0, it0 = O.minimize_reproj(X, obs, np.zeros(6), s["param"], S[h].astype(np.int32))
        c0 = len(O.get_inliers(X, obs, tr0, s["param"])[0]) if ok0 else 0
        c1 = int(cnt_h[t, h]) if ok_h[t, h] else 0
        if ok0 != ok_h[t, h] or c0 != c1:
            print("frame", t, "m", m, "hyp", h, S[h], "oracle ok", ok0, "it", it0, "cnt", c0, "tr", tr0, "| device ok", ok_h[t, h], "cnt", c1, "tr", tr_h[t, h])
