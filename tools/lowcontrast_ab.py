#!/usr/bin/env python3
"""tools/lowcontrast_ab.py: the temporal kernels on descriptors squeezed towards zero (scale s: values * s, rounded) —
the regime where match_union8_kernel's 8-bit bound stops settling queries and its rescue path scores more and more
candidates exactly.  Prints the timed kernel's average per 256-pair batch for variants 3 and 6 and the overflow count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

nf = 257
seq = synth.make_sequence(5, nf, n_kp=2000)
for scale in (1.0, 0.25, 0.1, 0.04):
    desc = np.rint(seq["desc"] * scale).astype(np.float32)
    out = []
    for v in (3, 6):
        ctx = libviso_amd.Context(0)
        libviso_amd.set_matcher_variant(v, ctx)
        b = libviso_amd.Batch(ctx, nf, 2000)
        b.upload(seq["kp"], desc, seq["n"])
        b.set_params(MatchParams.stereo(seq["F"]), MatchParams.temporal(), seq["param"], seed=1)
        for _ in range(3):
            b.run_matcher()
        ctx.synchronize()
        b.kernel_timing(True)
        for _ in range(10):
            b.run_matcher()
        ctx.synchronize()
        ms, n = b.kernel_ms()
        out.append((libviso_amd.matcher_kernel_name(ctx), ms, b.overflow_count(), int(b.counters()[1].sum())))
        b.close(); ctx.close()
    print(f"scale {scale}: " + "; ".join(f"{k} {ms:.3f} ms, {ov} overflow queries, {m} matches" for k, ms, ov, m in out), flush=True)
