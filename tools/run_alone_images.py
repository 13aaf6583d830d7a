"""One image batch (128 frame pairs), binned Harris on the device -> descriptors -> matcher -> solver, synchronize per run:
every kernel of the image-in pipeline ALONE on the GPU.  For rocprofv3 --kernel-trace --stats.
Usage: python tools/run_alone_images.py [frame pairs=128] [runs=12]"""
import sys
import torch  # noqa: F401
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 128
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
iseq = synth.make_image_sequence(2000, nf + 1, n_kp=2000, width=1241, height=376)
ctx = libviso_amd.Context(0)
b = libviso_amd.Batch(ctx, nf + 1, 1200)
b.upload_images_only(iseq["images"])
b.set_params(MatchParams.stereo(iseq["F"]), MatchParams.temporal(), iseq["param"], seed=1, first_frame=0)
for _ in range(runs):
    b.detect()
    b.run_images(False)
    ctx.synchronize()
tr, ok, n_inl = b.poses()
print("poses ok", int(ok[1:].sum()), "mean inliers", float(n_inl[1:].mean()))
