"""Random image sizes, bin grids and corner counts: the binned Harris detector and the response image against the oracle.
Test infrastructure (imports oracle/): lives under tests/; `cases()` feeds tests/test_gpu_harris.py, and as a script
(not collected by pytest) it runs longer sweeps and prints what differs:  python3 tests/harris_fuzz.py SEED N"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cases(seed, n):
    """n random (image, n_features, nbinx, nbiny): smooth textures, noise, black / white noise, one odd pixel in a flat image;
    1..139 rows, 1..259 columns, 1..29 x 1..11 bins, 1..39 corners per bin (the fused detector keeps up to 32)."""
    from libviso_amd import synth
    rng = np.random.default_rng(seed)
    for _ in range(n):
        rows, cols = int(rng.integers(1, 140)), int(rng.integers(1, 260))
        bx, by = int(rng.integers(1, max(2, min(cols, 30)))), int(rng.integers(1, max(2, min(rows, 12))))
        per = int(rng.integers(1, 40))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            img = synth.make_images(int(rng.integers(1 << 30)), rows, cols)
        elif kind == 1:
            img = rng.integers(0, 256, (rows, cols)).astype(np.uint8)
        elif kind == 2:
            img = (rng.integers(0, 2, (rows, cols)) * 255).astype(np.uint8)
        else:
            img = np.full((rows, cols), int(rng.integers(0, 256)), np.uint8)
            img[rng.integers(0, rows), rng.integers(0, cols)] ^= 0x80
        yield img, per * bx * by, bx, by


def describe(img, nf, bx, by, k0, k1):
    rows, cols = img.shape
    sx, sy = cols // bx, rows // by
    s1, s0 = set(map(tuple, k1.astype(int).tolist())), set(map(tuple, k0.astype(int).tolist()))
    miss = [p for p in s0 if p not in s1]
    return "%dx%d, %dx%d bins of %dx%d, %d per bin: oracle %d corners, device %d; device lacks %s, has extra %s" % (
        rows, cols, bx, by, sx, sy, nf // (bx * by), len(k0), len(k1),
        [(x, y, (y % sy) * sx + x % sx) for x, y in miss[:8]], [p for p in s1 if p not in s0][:8])


if __name__ == "__main__":
    import libviso_amd
    from oracle import pyoracle as O
    O.lib()
    libviso_amd.load()
    bad = 0
    for img, nf, bx, by in cases(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300):
        k0, r0 = O.detect_harris_binned(img, nf, bx, by)
        k1, r1 = libviso_amd.detect_harris_binned(img, nf, bx, by)
        ok = np.array_equal(k0, k1) and np.array_equal(r0, r1)
        ok2 = np.array_equal(libviso_amd.harris_response(img), O.harris_response(img))
        if not (ok and ok2):
            bad += 1
            print("MISMATCH (detector ok: %s, response ok: %s)" % (ok, ok2), describe(img, nf, bx, by, k0, k1), flush=True)
    print("done, mismatches:", bad)
    sys.exit(1 if bad else 0)
