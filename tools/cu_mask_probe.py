"""One 512-pair batch, matcher only, on a stream restricted to K of every 4 compute units (hipExtStreamCreateWithCUMask):
how the step's kernels scale with the CUs they may use (under rocprofv3 --kernel-trace --stats).
Usage: python tools/cu_mask_probe.py K [runs=12] [pattern=mod4|low]"""
import ctypes as C
import sys
import torch  # noqa: F401
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pattern = sys.argv[3] if len(sys.argv) > 3 else "mod4"
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
stream = C.c_void_p()
words = 8   # 256 CUs
mask = (C.c_uint32 * words)()
for i in range(256):
    on = (i % 4) < k if pattern == "mod4" else i < 64 * k
    if on:
        mask[i // 32] |= (1 << (i % 32))
r = hip.hipExtStreamCreateWithCUMask(C.byref(stream), C.c_uint32(words), mask)
assert r == 0, r
nf = 512
seq = synth.make_sequence(1000, nf + 1, n_kp=2000, width=1241, height=376)
ctx = libviso_amd.Context(0, stream=stream)
b = libviso_amd.Batch(ctx, nf + 1, 2000)
b.upload(seq["kp"], seq["desc"], seq["n"])
b.set_params(MatchParams.stereo(seq["F"]), MatchParams.temporal(), seq["param"], seed=1, first_frame=0)
for _ in range(runs):
    b.run_matcher()
    ctx.synchronize()
print("mask", k, "of 4", pattern, "done")
