"""latency of small batches (upload + run + get_poses) against the GN split of the RANSAC stage: python3 tools/experiments/latency_split.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams
s = synth.make_sequence(7, 17, n_kp=2000)
st, tm = MatchParams.stereo(s["F"]), MatchParams.temporal()
ctx = libviso_amd.Context(0)
for pairs in (1, 2, 4, 8, 16):
    nf = pairs + 1
    b = libviso_amd.Batch(ctx, nf, 2000)
    b.set_params(st, tm, s["param"], seed=1)
    for split in (0, 1, 2, 4):
        libviso_amd.set_gn_split(split, ctx)
        lat = []
        for i in range(30):
            t0 = time.perf_counter()
            b.upload(s["kp"][:nf], s["desc"][:nf], s["n"][:nf])
            b.run()
            b.poses()
            lat.append(time.perf_counter() - t0)
        print("pairs %2d split %d: median %.1f us  min %.1f us" % (pairs, split, np.median(lat[5:]) * 1e6, min(lat) * 1e6), flush=True)
    b.close()
