"""where the 1-pair batch's latency goes: upload / run / poses, wall time: python3 tools/experiments/latency_parts.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams
s = synth.make_sequence(7, 30, n_kp=2000)
st, tm = MatchParams.stereo(s["F"]), MatchParams.temporal()
ctx = libviso_amd.Context(0)
b = libviso_amd.Batch(ctx, 2, 2000)
b.set_params(st, tm, s["param"], seed=1)
parts = []
for i in range(28):
    t0 = time.perf_counter()
    b.upload(s["kp"][i:i + 2], s["desc"][i:i + 2], s["n"][i:i + 2])
    t1 = time.perf_counter()
    b.run()
    ctx.synchronize()
    t2 = time.perf_counter()
    b.poses()
    t3 = time.perf_counter()
    parts.append((t1 - t0, t2 - t1, t3 - t2))
p = np.median(np.array(parts[5:]), 0) * 1e6
print("upload %.1f us, run + synchronize %.1f us, poses %.1f us" % tuple(p))
b.kernel_timing(True)
for i in range(6):
    b.upload(s["kp"][i:i + 2], s["desc"][i:i + 2], s["n"][i:i + 2]); b.run(); b.poses()
print(b.kernel_ms())
