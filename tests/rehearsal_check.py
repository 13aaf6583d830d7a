#!/usr/bin/env python3
"""Oracle check of a full-length KITTI rehearsal (tools/kitti_rehearsal.py).  Test infrastructure: this
is the only part of the rehearsal that touches oracle/, which is why it lives under tests/.

    python tests/rehearsal_check.py KITTI_HOME SEQ RESULT_SHA WORLD [--every 50] [--procs N]

reads the rank files `results/SEQ/RESULT_SHA/shards/SEQ.<r>of<WORLD>.rec` the GPU runner left ({tr[6], ok, n_inl, frame}
per frame pair), then runs the CPU oracle -- binned Harris detector, descriptor extractor, the three match_desc calls,
match_circle, triangulation, RANSAC / Gauss-Newton, all from the PNG files -- on every `--every`-th pair AND on every pair
the GPU runner reports as unsolved, and compares: `ok` and `n_inl` exactly, `tr` within 1e-5 relative Frobenius of
tr2mat.  The unsolved pairs are tabulated by the exit of the reference's loop they take:
    src/viso.cpp:1283  fewer than 3 circle matches
    src/viso.cpp:1571  the best hypothesis has fewer than 6 inliers (or no hypothesis converged)
    src/viso.cpp:1573  the refit on the inliers fails (:1605 singular system / :1622 100 iterations without convergence)
Exit code 0 iff every checked pair agrees."""
import argparse
import multiprocessing as mp
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
POSE_TOL = 1e-5


def read_records(path):
    b = open(path, "rb").read()
    magic, first, last, n = struct.unpack("<4i", b[:16])
    rec = np.frombuffer(b, dtype=np.dtype([("tr", "<f8", 6), ("ok", "<i4"), ("n_inl", "<i4"), ("frame", "<i4"), ("res", "<i4")]), count=n, offset=16)
    return first, last, rec


def load_calib(path):
    P = {}
    for line in open(path):
        k, v = line.split(":", 1)
        P[k.strip()] = np.array([float(x) for x in v.split()]).reshape(3, 4)
    return P["P0"], P["P1"]


def _gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), dtype=np.uint8)


def _check(job):
    base, t, seed, P1, P2, gpu = job
    from oracle import pyoracle as O
    from libviso_amd.abi import MatchParams, Param
    F = O.F_from_P(P1, P2)
    st, tm = MatchParams.stereo(F), MatchParams.temporal()
    prm = Param.default(base=abs(P2[0, 3] / P2[0, 0]), f=P1[0, 0], cu=P1[0, 2], cv=P1[1, 2])
    cap = 1200
    kp = np.zeros((2, 2, cap, 2), np.float32); desc = np.zeros((2, 2, cap, 121), np.float32); n = np.zeros((2, 2), np.int32)
    for j, fr in enumerate((t - 1, t)):
        for side in (0, 1):
            img = _gray(os.path.join(base, f"image_{side}", "%06d.png" % fr))
            k, _ = O.detect_harris_binned(img, 1200, 24, 5, O.HARRIS_K)
            n[j, side] = len(k)
            kp[j, side, :len(k)] = k
            desc[j, side, :len(k)] = O.extract_descriptors(img, k, 5)
    o = O.sequence(kp, desc, n, st, tm, prm, seed=seed, first_frame=t - 1)
    ok, n_inl, tr = int(o["ok"][1]), int(o["n_inl"][1]), o["tr"][1]
    same = ok == gpu[0] and (n_inl == gpu[1] or not ok)
    err = 0.0
    if same and ok:
        a, b = O.tr2mat(np.asarray(gpu[2])), O.tr2mat(tr)
        err = float(np.linalg.norm(a - b) / np.linalg.norm(b))
        same = err < POSE_TOL
    exit_ = ""
    if not ok:   # which exit of the loop body
        def m(j, a, b, p):
            return O.match_desc(kp[j[0], a, :n[j[0], a]], kp[j[1], b, :n[j[1], b]], desc[j[0], a, :n[j[0], a]], desc[j[1], b, :n[j[1], b]], p)
        lr, lrp = m((1, 1), 0, 1, st), m((0, 0), 0, 1, st)
        m11, m22 = m((1, 0), 0, 0, tm), m((1, 0), 1, 1, tm)
        _, circ, pcl, nc = O.match_circle(lr, lrp, m11, m22)
        if nc < 3:
            exit_ = ":1283 fewer than 3 circle matches"
        else:
            x = O.collect_matches(kp[1, 0, :n[1, 0]], kp[1, 1, :n[1, 1]], lr)
            xp = O.collect_matches(kp[0, 0, :n[0, 0]], kp[0, 1, :n[0, 1]], lrp)
            Xp = O.triangulate_rectified(xp, prm)
            r, _tr, inl = O.ransac_minimize_reproj(np.ascontiguousarray(Xp[:, pcl[:, 1]]), np.ascontiguousarray(x[:, pcl[:, 0]]), prm, seed=seed, frame=t)
            assert not r
            exit_ = ":1571 fewer than 6 inliers" if len(inl) < 6 else ":1573 refit failed (:1605 / :1622)"
        exit_ += f" [{nc} circle matches]"
    return t, same, ok, n_inl, gpu[0], gpu[1], err, exit_


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("home"); ap.add_argument("seq"); ap.add_argument("sha"); ap.add_argument("world", type=int)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--procs", type=int, default=min(16, len(os.sched_getaffinity(0))))
    a = ap.parse_args()
    base = os.path.join(a.home, "sequences", a.seq)
    P1, P2 = load_calib(os.path.join(base, "calib.txt"))
    recs = {}
    for r in range(a.world):
        _f, _l, rec = read_records(os.path.join(a.home, "results", a.seq, a.sha, "shards", f"{a.seq}.{r}of{a.world}.rec"))
        for x in rec:
            recs[int(x["frame"])] = (int(x["ok"]), int(x["n_inl"]), [float(v) for v in x["tr"]])
    frames = sorted(recs)
    unsolved = [t for t in frames if not recs[t][0]]
    sample = sorted(set(frames[::a.every]) | set(unsolved))
    from oracle import pyoracle
    pyoracle.lib()
    jobs = [(base, t, a.seed, P1, P2, recs[t]) for t in sample]
    with mp.get_context("fork").Pool(a.procs) as pool:
        out = pool.map(_check, jobs, chunksize=2)
    bad = [o for o in out if not o[1]]
    worst = max([o[6] for o in out] + [0.0])
    print(f"oracle check of {len(frames)} frame pairs of the GPU runner ({a.sha}): {len(sample)} pairs through the CPU oracle "
          f"(every {a.every}th: {len(frames[::a.every])}, every unsolved pair: {len(unsolved)}), {len(sample) - len(bad)} agree "
          f"(ok and n_inl exact, tr within {POSE_TOL:g} relative Frobenius; worst {worst:.2e}), {len(bad)} differ")
    exits = {}
    for o in out:
        if o[7]:
            exits.setdefault(o[7].split(" [")[0], []).append(o[0])
    print(f"the {len(unsolved)} unsolved pairs by the exit of the reference's loop body they take (oracle):")
    for k in sorted(exits):
        v = exits[k]
        print(f"  src/viso.cpp{k}: {len(v)} pairs  (frames {', '.join(str(t) for t in v[:12])}{' ...' if len(v) > 12 else ''})")
    seams = [t for t in unsolved if t % 71 == 0]
    print(f"  {len(seams)} of them are seams of the synthetic tree (frames that start an independent 71-frame block: no real motion joins them)")
    for o in bad[:20]:
        print(f"  DIFFERS frame {o[0]}: oracle ok {o[2]} n_inl {o[3]}, GPU ok {o[4]} n_inl {o[5]}, tr err {o[6]:.2e} {o[7]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
