"""tools/experiments/sortkp_phases.py (debug build: tools/build_dbg.sh, VISO_HIP_SO=build_ab/dbg.so): where sort_kp_kernel's time goes for ONE image
(workgroup 0 of the per-call loop's launch) -- 100 MHz time stamps the kernel leaves (viso_debug_sortkp_clocks)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams
seq = synth.make_sequence(1000, 30, n_kp=2000)
L = libviso_amd.load()
st = MatchParams.stereo(seq["F"])
rows = []
for t in range(30):
    nL, nR = seq["n"][t]
    libviso_amd.match_desc(seq["kp"][t, 0, :nL].copy(), seq["kp"][t, 1, :nR].copy(), seq["desc"][t, 0, :nL].copy(), seq["desc"][t, 1, :nR].copy(), st)
    clk = (C.c_uint64 * 8)()
    assert L.viso_debug_sortkp_clocks(clk) >= 0
    c = [int(v) for v in clk]
    if t > 3: rows.append([(c[i + 1] - c[i]) / 100 for i in range(5)])
r = np.array(rows)
names = ["view, keypoints (over PCIe here), extent walk", "extent reduction", "scale (one thread), barrier", "histogram, scan", "scatter (+ global stores)", "y rank inside the 64-blocks"]
print("sort_kp_kernel, one image of 2000 keypoints (us): " + " | ".join("%s %.1f" % (n, v) for n, v in zip(names, r.mean(0))) + " | total %.1f" % r.sum(1).mean())
