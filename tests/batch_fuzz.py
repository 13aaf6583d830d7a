"""Random sequences through the batch family against oracle_sequence: image sizes, keypoint counts (ragged, empty frames,
frames of two keypoints), capacities that are no multiple of anything, duplicated descriptors (SAD ties), clustered
keypoints, every matcher variant, seeds and first frames.  Test infrastructure (imports oracle/): lives under tests/;
`run(seed, n, L, O)` feeds tests/test_gpu_fuzz.py, as a script (not collected):  python3 tests/batch_fuzz.py SEED N"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
POSE_TOL = 1e-5


def _solved_alike(ok, n_inl, want):
    """the same frames solved, and the same support where a pose was found.  (A frame WITHOUT a pose reports the support
    of its best failed hypothesis; on a handful of circle matches that can be a 3-point Gauss-Newton run that wanders
    through rotations of thousands of radians, where one ulp of sincos -- the device's differs from libm's in 3 % of
    arguments, tools/experiments/sincos_parity.hip -- decides whether it comes back: not compared here.  tests/test_gpu_solver_edges.py::test_every_hypothesis_against_the_oracle
    compares every hypothesis of a sequence with the oracle -- exactly for the ones the oracle decides within 20
    iterations -- and bounds the share of the wanderers that differ.)"""
    ok, w = np.asarray(ok), np.asarray(want["ok"])
    return np.array_equal(ok, w) and np.array_equal(np.asarray(n_inl)[ok != 0], np.asarray(want["n_inl"])[w != 0])


def one(rng, L, O, ctx):
    from libviso_amd import synth
    from libviso_amd.abi import MatchParams
    nf = int(rng.integers(2, 7))
    n_kp = int(rng.choice([40, 150, 333, 700, 1100]))
    width, height = int(rng.choice([300, 640, 1241])), int(rng.choice([120, 376, 480]))
    cap = n_kp + int(rng.integers(0, 70))
    s = synth.make_sequence(int(rng.integers(1 << 30)), nf, n_kp=n_kp, width=width, height=height, cap=cap,
                            ragged=bool(rng.integers(0, 2)), dup_frac=float(rng.choice([0.0, 0.0, 0.1])),
                            cluster_frac=float(rng.choice([0.0, 0.0, 0.7])), outlier_frac=float(rng.choice([0.1, 0.3])))
    kind = int(rng.integers(0, 5))
    if kind == 1:
        s["n"][int(rng.integers(0, nf))] = [0, 0]                           # a frame without keypoints
    elif kind == 2:
        s["n"][int(rng.integers(0, nf))] = [2, int(rng.integers(0, 3))]
    elif kind == 3:
        t = int(rng.integers(1, nf))                                       # the same frame twice: zero motion
        for k in ("kp", "desc", "n"):
            s[k][t] = s[k][t - 1]
    variant = int(rng.choice(L.MATCHER_VARIANTS))
    seed, first = int(rng.integers(0, 100)), int(rng.integers(0, 5000))
    st, tm = MatchParams.stereo(s["F"]), MatchParams.temporal()
    want = O.sequence(s["kp"], s["desc"], s["n"], st, tm, s["param"], seed=seed, first_frame=first)
    L.set_matcher_variant(variant, ctx)
    b = L.Batch(ctx, nf, cap)
    try:
        b.upload(s["kp"], s["desc"], s["n"])
        b.set_params(st, tm, s["param"], seed=seed, first_frame=first)
        b.run()
        tr, ok, n_inl = b.poses()
        sc, mo = b.counters()
    finally:
        b.close()
    what = "nf=%d n_kp=%d cap=%d %dx%d kind=%d variant=%d seed=%d first=%d" % (nf, n_kp, cap, width, height, kind, variant, seed, first)
    if not (_solved_alike(ok, n_inl, want)):
        return False, what + " | ok %s / %s, inliers %s / %s" % (want["ok"], ok, want["n_inl"], n_inl)
    if not (np.array_equal(sc, want["scored"]) and np.array_equal(mo, want["m_out"])):
        return False, what + " | scored / matches differ"
    for t in range(1, nf):
        if ok[t]:
            a, r = L.tr2mat(tr[t]), O.tr2mat(want["tr"][t])
            if not np.linalg.norm(a - r) <= POSE_TOL * np.linalg.norm(r):
                return False, what + " | pose %d: %s / %s" % (t, want["tr"][t], tr[t])
    return True, what


def one_images(rng, L, O, ctx):
    """images in: keypoints given (extract_pack_kernel) or found on the device (the binned Harris detector, any bin geometry)"""
    from libviso_amd import synth
    from libviso_amd.abi import MatchParams
    nf = int(rng.integers(2, 5))
    width, height = int(rng.choice([200, 333, 640])), int(rng.choice([96, 150, 240]))
    n_kp = int(rng.choice([60, 200, 400]))
    s = synth.make_image_sequence(int(rng.integers(1 << 30)), nf, n_kp=n_kp, width=width, height=height)
    detect = bool(rng.integers(0, 2))
    seed, first = int(rng.integers(0, 100)), int(rng.integers(0, 5000))
    st, tm = MatchParams.stereo(s["F"]), MatchParams.temporal()
    if detect:
        bx, by = int(rng.integers(1, 14)), int(rng.integers(1, 6))
        per = int(rng.choice([3, 10, 20, 32, 35]))
        nfeat = per * bx * by
        cap = nfeat
        kp = np.zeros((nf, 2, cap, 2), np.float32)
        n = np.zeros((nf, 2), np.int32)
        for t in range(nf):
            for side in range(2):
                k, _ = O.detect_harris_binned(s["images"][t, side], nfeat, bx, by)
                n[t, side] = len(k)
                kp[t, side, :len(k)] = k
        what = "images %dx%d nf=%d detect %dx%d bins, %d per bin" % (height, width, nf, bx, by, per)
    else:
        cap = s["kp"].shape[2]
        kp, n = s["kp"], s["n"]
        if rng.integers(0, 2):                                             # keypoints on and around the border
            for t in range(nf):
                for side in range(2):
                    m = min(int(n[t, side]), 12)
                    kp[t, side, :m, 0] = rng.choice([0, 1, 5, 6, width - 7, width - 6, width - 1], m)
                    kp[t, side, :m, 1] = rng.choice([0, 1, 5, 6, height - 7, height - 6, height - 1], m)
        what = "images %dx%d nf=%d keypoints given (%d)" % (height, width, nf, cap)
    desc = np.zeros((nf, 2, cap, 121), np.float32)
    for t in range(nf):
        for side in range(2):
            desc[t, side, :n[t, side]] = O.extract_descriptors(s["images"][t, side], kp[t, side, :n[t, side]])
    want = O.sequence(kp, desc, n, st, tm, s["param"], seed=seed, first_frame=first)
    b = L.Batch(ctx, nf, cap)
    try:
        b.set_params(st, tm, s["param"], seed=seed, first_frame=first)
        if detect:
            b.upload_images_only(s["images"])
            b.detect(nfeat, bx, by)
        else:
            b.upload_images(s["images"], kp, n)
        b.run_images()
        if detect:
            for t in range(nf):
                for side in range(2):
                    if not np.array_equal(b.keypoints(t, side), kp[t, side, :n[t, side]]):
                        return False, what + " | keypoints of frame %d side %d differ" % (t, side)
        tr, ok, n_inl = b.poses()
        sc, mo = b.counters()
        detail = ""
        if not (_solved_alike(ok, n_inl, want)):
            # what the solver was given: the join of the device's own match lists, by the oracle's nested loops
            for t in range(1, nf):
                r_, circ, pcl, nc = O.match_circle(b.matches(0, t), b.matches(0, t - 1), b.matches(1, t), b.matches(2, t))
                c1, p1 = b.circle(t)
                detail += " [frame %d: matches %s, join %d (oracle's loops on the same lists: %d)]" % (t, [len(b.matches(w, t)) for w in range(3)], len(c1), nc)
    finally:
        b.close()
    if not (np.array_equal(sc, want["scored"]) and np.array_equal(mo, want["m_out"])):
        return False, what + " | scored / matches differ: %s / %s" % (want["m_out"].tolist(), mo.tolist())
    if not (_solved_alike(ok, n_inl, want)):
        if os.environ.get("VISO_FUZZ_DUMP"):   # the sequence as the solver stage saw it, for tests/solver_case.py
            p = s["param"]
            np.savez(os.environ["VISO_FUZZ_DUMP"], kp=kp, desc=desc, n=n, F=s["F"], seed=seed, first=first,
                     param=np.frombuffer(bytes(p), np.uint8))
        return False, what + " | ok %s / %s, inliers %s / %s, seed %d first %d%s" % (want["ok"], ok, want["n_inl"], n_inl, seed, first, detail)
    for t in range(1, nf):
        if ok[t]:
            a, r = L.tr2mat(tr[t]), O.tr2mat(want["tr"][t])
            if not np.linalg.norm(a - r) <= POSE_TOL * np.linalg.norm(r):
                return False, what + " | pose %d: %s / %s" % (t, want["tr"][t], tr[t])
    return True, what


def run(seed, n, L, O, images=True):
    rng = np.random.default_rng(seed)
    ctx = L.Context(0)
    default = L.DEFAULT_MATCHER
    bad = []
    try:
        for _ in range(n):
            ok, what = one(rng, L, O, ctx)
            if not ok:
                bad.append(what)
            if images:
                L.set_matcher_variant(default, ctx)
                ok, what = one_images(rng, L, O, ctx)
                if not ok:
                    bad.append(what)
    finally:
        L.set_matcher_variant(default, ctx)
        ctx.close()
    return bad


if __name__ == "__main__":
    import libviso_amd
    from oracle import pyoracle as O
    O.lib()
    libviso_amd.load()
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40, libviso_amd, O)
    for b in bad[:30]:
        print("MISMATCH", b)
    print("done, mismatches:", len(bad))
    sys.exit(1 if bad else 0)
