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
mrows = []
tm = MatchParams.temporal()
prev = None
for t in range(30):
    nL, nR = seq["n"][t]
    kp1, kp2 = seq["kp"][t, 0, :nL].copy(), seq["kp"][t, 1, :nR].copy()
    d1, d2 = seq["desc"][t, 0, :nL].copy(), seq["desc"][t, 1, :nR].copy()
    libviso_amd.match_desc(kp1, kp2, d1, d2, st)   # from the second frame on: the three-problem launch of the loop
    if prev is not None:
        libviso_amd.match_desc(kp1, prev[0], d1, prev[2], tm)
        libviso_amd.match_desc(kp2, prev[1], d2, prev[3], tm)
    prev = (kp1, kp2, d1, d2)
    clk = (C.c_uint64 * 8)()
    assert L.viso_debug_sortkp_clocks(clk) >= 0
    c = [int(v) for v in clk]
    if t > 3: rows.append([(c[i + 1] - c[i]) / 100 for i in range(5)])
    cm = (C.c_uint64 * 12)()
    if hasattr(L, "viso_debug_sortm_clocks") and L.viso_debug_sortm_clocks(cm) >= 0:
        m = [int(v) for v in cm]
        if t > 3 and m[7] > m[0]: mrows.append([(m[i + 1] - m[i]) / 100 for i in range(7)])
r = np.array(rows)
names = ["view, keypoints (over PCIe here), extent walk", "extent reduction + scale", "histogram, scan", "scatter (+ global stores)", "y rank inside the 64-blocks"]
print("sort_kp_kernel, one image of 2000 keypoints (us): " + " | ".join("%s %.1f" % (n, v) for n, v in zip(names, r.mean(0))) + " | total %.1f" % r.sum(1).mean())

if mrows:
    mr = np.array(mrows)
    mn = ["problem, results, compaction", "min / max / sum", "two trims of the mean", "bucket map + histogram", "scan (wave 0)", "scatter", "rank in bucket + rows out (+ collect / triangulate)"]
    print("sort_matches_kernel, the frame's left temporal problem (us): " + " | ".join("%s %.1f" % (n, v) for n, v in zip(mn, mr.mean(0))) + " | total %.1f" % mr.sum(1).mean())
