"""A sequence dumped by tests/batch_fuzz.py ($VISO_FUZZ_DUMP) whose solver stage differs from the oracle's: hypothesis by
hypothesis (a script, not collected):  python3 tests/solver_case.py dump.npz"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libviso_amd
from libviso_amd.abi import MatchParams, Param
from oracle import pyoracle as O

O.lib()
libviso_amd.load()
d = np.load(sys.argv[1])
kp, desc, n = d["kp"], d["desc"], d["n"]
param = Param.from_buffer_copy(d["param"].tobytes())
seed, first = int(d["seed"]), int(d["first"])
nf, _, cap, _ = kp.shape
st, tm = MatchParams.stereo(d["F"]), MatchParams.temporal()
ctx = libviso_amd.Context(0)
b = libviso_amd.Batch(ctx, nf, cap)
b.upload(kp, desc, n)
b.set_params(st, tm, param, seed=seed, first_frame=first)
b.run()
tr, ok, n_inl = b.poses()
tr_h, ok_h, cnt_h, nu = b.hypotheses()
want = O.sequence(kp, desc, n, st, tm, param, seed=seed, first_frame=first)
print("oracle ok", want["ok"], "inliers", want["n_inl"], "| device ok", ok, "inliers", n_inl)
for t in range(1, nf):
    lr, lrp, m11, m22 = b.matches(0, t), b.matches(0, t - 1), b.matches(1, t), b.matches(2, t)
    circ, pcl = b.circle(t)
    m = len(circ)
    if m < 3:
        continue
    # what sequence_odometry hands the solver (src/viso.cpp:1292-1305)
    x = O.collect_matches(kp[t, 0, :n[t, 0]], kp[t, 1, :n[t, 1]], lr)
    xp = O.collect_matches(kp[t - 1, 0, :n[t - 1, 0]], kp[t - 1, 1, :n[t - 1, 1]], lrp)
    Xp = O.triangulate_rectified(xp, param)
    obs = np.ascontiguousarray(x[:, pcl[:, 0]]); X = np.ascontiguousarray(Xp[:, pcl[:, 1]])
    S = np.asarray(O.ransac_samples(seed, first + t, param.ransac_iter, m)).reshape(-1, 3)
    nd = 0
    for h, s3 in enumerate(S):
        ok0, tr0, it0 = O.minimize_reproj(X, obs, np.zeros(6), param, s3.astype(np.int32))
        c0 = len(O.get_inliers(X, obs, tr0, param)[0]) if ok0 else 0
        same = ok0 == ok_h[t, h] and (not ok0 or (np.array_equal(tr0, tr_h[t, h]) and c0 == cnt_h[t, h]))
        if not same:
            nd += 1
            if nd <= 6:
                print("frame %d hypothesis %d sample %s: oracle ok %d it %d count %d tr %s | device ok %d count %d tr %s" % (
                    t, h, s3, ok0, it0, c0, tr0, ok_h[t, h], cnt_h[t, h], tr_h[t, h]))
    print("frame %d: %d circle matches, %d of %d hypotheses differ" % (t, m, nd, len(S)))
b.close(); ctx.close()
