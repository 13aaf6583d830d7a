"""tools/experiments/frame_phases.py (debug build: tools/build_dbg.sh, VISO_HIP_SO=build_ab/dbg.so): where match_frame_kernel's time goes on ONE frame of
the per-call loop -- 100 MHz time stamps the stereo part's first tile leaves (viso_debug_frame_clocks)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libviso_amd
from libviso_amd import synth, drop_in
from libviso_amd.abi import MatchParams
seq = synth.make_sequence(1000, 40, n_kp=2000)
L = libviso_amd.load()
o = drop_in.run(seq["kp"][:4], seq["desc"][:4], seq["n"][:4], seq["F"], seq["param"], seed=1)
rows = []
blocks = []
urows = []
st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
state = None
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
    assert L.viso_debug_frame_clocks(clk) >= 0
    c = [int(v) for v in clk]
    uc = (C.c_uint64 * 8)()
    assert L.viso_debug_frame_uclocks(uc) >= 0
    u = [int(v) for v in uc]
    if t > 5 and u[3] > u[0]: urows.append([(u[1] - u[0]) / 100, (u[2] - u[1]) / 100, (u[3] - u[2]) / 100])
    blk = (C.c_uint64 * 512)()
    assert L.viso_debug_frame_blocks(blk) >= 0
    b = np.array(list(blk), np.int64).reshape(2, 256)
    used = b[1] > 0
    if t > 5 and used.any():
        t0 = b[0][used].min()
        blocks.append(((b[0][used] - t0) / 100.0, (b[1][used] - t0) / 100.0, np.nonzero(used)[0]))
    if t > 5 and c[8] > c[0]: rows.append([(c[i + 1] - c[i]) / 100 for i in range(8)] + [(c[9] - c[0]) / 100])
    state = {"kp1": kp1, "kp2": kp2, "d1": d1, "d2": d2, "lr": lr, "X": X}
r = np.array(rows, float)
names = ["problem + query loads, band", "window bounds, Q1", "window loads + y bucket sort", "walk", "exact gate", "flat list + scoring", "reduce", "K cap", ]
print("frames", len(r))
print("stereo part, first tile (us): " + " | ".join("%s %.1f" % (n, v) for n, v in zip(names, r[:, :8].mean(0))) + " | then results: total %.1f" % r[:, :8].sum(1).mean())
print("first union8 tile ends %.1f us after the stereo tile started" % r[:, 8].mean())

# per workgroup: when it started and ended, relative to the launch's first start
ends = np.array([e.max() for _, e, _ in blocks]); print("kernel span by the stamps: %.1f us on average" % ends.mean())
st0, en0, ids = blocks[len(blocks) // 2]
order = np.argsort(en0)
print("one frame: workgroups", len(ids), "; last to end:", [(int(ids[i]), round(float(st0[i]), 1), round(float(en0[i]), 1)) for i in order[-6:]])
ns = int((ids < 8).sum())
dur = en0 - st0
print("durations: workgroups 0-7 (stereo part) mean %.1f max %.1f ; the rest (union8 part) mean %.1f max %.1f ; latest start %.1f" % (dur[ids < 8].mean(), dur[ids < 8].max(), dur[ids >= 8].mean(), dur[ids >= 8].max(), st0.max()))
real = dur[(ids >= 8) & (dur > 1.0)]
print("union8 workgroups with work: %d, durations sorted: %s" % (len(real), " ".join("%.0f" % v for v in np.sort(real))))
sreal = dur[(ids < 8)]
print("stereo workgroups: %s" % " ".join("%.0f" % v for v in np.sort(sreal)))

if urows:
    ur = np.array(urows)
    print("union8 part, an interior tile of the left temporal problem, wave 0 (us): staging (window, y index) %.1f | round 0 %.1f | round 1 %.1f" % tuple(ur.mean(0)))
