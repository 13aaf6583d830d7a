"""tools/experiments/refit_phases.py (debug build: make DEBUG_VARIANTS=1 OUT=... , VISO_HIP_SO=that): where ransac_refit_kernel's time goes on ONE
frame of the per-call loop -- 100 MHz time stamps the kernel leaves (viso_debug_refit_clocks): best-hypothesis reduction, support
set, Gauss-Newton on the support set (with its iteration count), final support set, results + copy-out."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libviso_amd
from libviso_amd import synth, drop_in
from libviso_amd.abi import MatchParams
seq = synth.make_sequence(1000, 40, n_kp=2000)
L = libviso_amd.load()
st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
drop_in.plain_speculate(False)
state = None
rows = []
for t in range(40):
    nL, nR = seq["n"][t]
    kp1, kp2 = seq["kp"][t, 0, :nL].copy(), seq["kp"][t, 1, :nR].copy()
    d1, d2 = seq["desc"][t, 0, :nL].copy(), seq["desc"][t, 1, :nR].copy()
    lr = libviso_amd.match_desc(kp1, kp2, d1, d2, st)
    x = libviso_amd.collect_matches(kp1, kp2, lr)
    X = libviso_amd.triangulate_rectified(x, seq["param"])
    if state is not None:
        m11 = libviso_amd.match_desc(kp1, state["kp1"], d1, state["d1"], tm)
        m22 = libviso_amd.match_desc(kp2, state["kp2"], d2, state["d2"], tm)
        _, circ, pcl, n = libviso_amd.match_circle(lr, state["lr"], m11, m22)
        if n >= 3:
            x_c, Xp_c = np.ascontiguousarray(x[:, pcl[:, 0]]), np.ascontiguousarray(state["X"][:, pcl[:, 1]])
            libviso_amd.ransac_minimize_reproj(Xp_c, x_c, seq["param"], seed=1, frame=t)
            clk = (C.c_uint64 * 16)()
            assert L.viso_debug_refit_clocks(clk) >= 0
            c = [int(v) for v in clk]
            rows.append([(c[1] - c[0]) / 100, (c[2] - c[1]) / 100, (c[3] - c[2]) / 100, (c[4] - c[3]) / 100, (c[5] - c[4]) / 100, c[6], c[7], n])
    state = {"kp1": kp1, "kp2": kp2, "d1": d1, "d2": d2, "lr": lr, "X": X}
r = np.array(rows, float)
print("frames", len(r))
print("us: best-hypothesis reduction %.1f | support set %.1f | Gauss-Newton %.1f (%.1f iterations, %.2f us each) | final support set %.1f | results + copy-out %.1f"
      % (r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 6].mean(), (r[:, 2] / r[:, 6]).mean(), r[:, 3].mean(), r[:, 4].mean()))
print("inliers of the best hypothesis %.0f of %.0f points" % (r[:, 5].mean(), r[:, 7].mean()))
