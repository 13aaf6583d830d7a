"""One batch, viso_batch_run + synchronize per step: every kernel of the chain runs ALONE on the GPU (no other batch in
flight, no next run behind it).  For rocprofv3 --kernel-trace --stats / --pmc passes of the RANSAC stage.
Usage: python tools/run_alone.py [frame pairs=512] [runs=12]"""
import sys
import torch  # noqa: F401
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 512
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seq = synth.make_sequence(1000, nf + 1, n_kp=2000, width=1241, height=376)
ctx = libviso_amd.Context(0)
b = libviso_amd.Batch(ctx, nf + 1, 2000)
b.upload(seq["kp"], seq["desc"], seq["n"])
b.set_params(MatchParams.stereo(seq["F"]), MatchParams.temporal(), seq["param"], seed=1, first_frame=0)
for _ in range(runs):
    b.run()
    ctx.synchronize()
tr, ok, n_inl = b.poses()
import numpy as np
print("poses ok", int(ok[1:].sum()), "mean inliers", float(n_inl[1:].mean()), "circle matches (frame 5):", len(b.circle(5)[0]) if hasattr(b, "circle") else "?")
