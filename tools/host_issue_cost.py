"""Host-side cost of issuing one step (all launches of viso_batch_run*) against the GPU time of the step.
Usage: python tools/host_issue_cost.py [frames] [images:0|1] [batches in flight] [init torch first:0|1] [idle batch first:0|1] [run|matcher] [timing events 0|1|2]"""
import sys, time
import numpy as np
import torch  # noqa: F401  (one HIP runtime in the process)
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
images = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(sys.argv) > 4 and int(sys.argv[4]):   # initialise torch's HIP state first, as bench.py does
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
ctxs, runs = [], []
for _ in range(ns):
  ctx = libviso_amd.Context(0)
  ctxs.append(ctx)
  if len(sys.argv) > 5 and int(sys.argv[5]):   # an idle batch per context first: its RANSAC stream exists and stays unused
      dummies = globals().setdefault('dummies', [])
      dummies.append(libviso_amd.Batch(ctx, 8, 2000))
  if images:
      seq = synth.make_image_sequence(2000, nf + 1, n_kp=2000, width=1241, height=376)
      b = libviso_amd.Batch(ctx, nf + 1, 2000)
      b.upload_images(seq["images"], seq["kp"], seq["n"])
      runs.append(lambda b=b: b.run_images(False))
  else:
      seq = synth.make_sequence(1000, nf + 1, n_kp=2000, width=1241, height=376)
      b = libviso_amd.Batch(ctx, nf + 1, 2000)
      b.upload(seq["kp"], seq["desc"], seq["n"])
      runs.append(lambda b=b: b.run())
  st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
  b.set_params(st, tm, seq["param"], seed=1, first_frame=0)
mode = sys.argv[6] if len(sys.argv) > 6 else "run"          # "matcher": viso_batch_run_matcher only
timing = int(sys.argv[7]) if len(sys.argv) > 7 else 0         # 1: a few runs with kernel timing (HIP events) first, then off; 2: left on
batches = [r.__defaults__[0] for r in runs]
if mode == "matcher":
    runs = [(lambda b=b: b.run_matcher()) for b in batches]
if timing:
    for b in batches:
        b.kernel_timing(True)
    for i in range(3 * ns):
        runs[i % ns]()
    for c in ctxs:
        c.synchronize()
    if timing == 1:
        for b in batches:
            b.kernel_timing(False)
for i in range(3 * ns):
    runs[i % ns]()
for c in ctxs:
    c.synchronize()
n = 60
marks = []
t0 = time.perf_counter()
for i in range(n):
    runs[i % ns]()
    marks.append(time.perf_counter())
t1 = time.perf_counter()
for c in ctxs:
    c.synchronize()
t2 = time.perf_counter()
gaps = np.diff(np.array([t0] + marks)) * 1e3
print(f"frames {nf} images {images} batches {ns} mode {mode} timing {timing}: host issue {1e3 * (t1 - t0) / n:.3f} ms/step (max {gaps.max():.3f}), total {1e3 * (t2 - t0) / n:.3f} ms/step")
