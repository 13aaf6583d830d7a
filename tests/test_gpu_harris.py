"""Binned Harris detector on device against the oracle's
restatement of HarrisBinnedFeatureDetector::detectImpl (reference
src/viso.cpp:926-975; k explicit, per-bin order defined), and the complete
image -> keypoints -> descriptors -> matches -> pose path without any host
round trip."""
import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(60, 96), (37, 131), (376, 1241)])
def test_harris_response_bit_exact(viso, oracle, shape):
    img = synth.make_images(shape[0], shape[0], shape[1])
    a, b = libviso_amd.harris_response(img), oracle.harris_response(img)
    assert np.array_equal(a, b)
    assert np.abs(a).max() > 0


# the response image is walked in bands whose height the launcher picks by the amount of work (and VISO_HARRIS_BAND
# overrides): every height gives the same image
@pytest.mark.parametrize("band", ["1", "7", "19", "76", "500"])
def test_harris_response_band_heights(viso, oracle, band, monkeypatch):
    monkeypatch.setenv("VISO_HARRIS_BAND", band)
    for shape in ((37, 131), (200, 59), (3, 3), (2, 70), (1, 9), (90, 1)):
        img = synth.make_images(11, shape[0], shape[1])
        assert np.array_equal(libviso_amd.harris_response(img), oracle.harris_response(img)), shape


def test_detect_binned_bit_exact(viso, oracle):
    img = synth.make_image_sequence(3, 1, n_kp=1500)["images"][0, 0]          # 376 x 1241
    for nf, bx, by in ((1200, 24, 5), (120, 6, 3), (64, 2, 2)):
        k0, r0 = oracle.detect_harris_binned(img, nf, bx, by)
        k1, r1 = libviso_amd.detect_harris_binned(img, nf, bx, by)
        assert np.array_equal(k0, k1) and np.array_equal(r0, r1)
        assert len(k0) == (nf // (bx * by)) * bx * by
    flat = np.full((40, 60), 7, np.uint8)                                      # zero response everywhere: no corners
    assert len(libviso_amd.detect_harris_binned(flat, 120, 6, 2)[0]) == 0 == len(oracle.detect_harris_binned(flat, 120, 6, 2)[0])


# bin widths either side of the detector's two pixel sources (direct image walk up to 58 columns, LDS tile for 59..62),
# bins that touch both image borders at once, bins a few rows / columns small, images narrower than a wave
@pytest.mark.parametrize("shape,bx,by,per", [
    ((376, 1241), 21, 5, 10),     # 59 columns: LDS tile
    ((376, 1241), 20, 4, 12),     # 62 columns: LDS tile, the widest
    ((376, 1241), 22, 5, 10),     # 56 columns
    ((100, 58), 1, 2, 16),        # 58 columns: the widest direct bin, left and right border in one tile
    ((64, 116), 2, 1, 20),
    ((50, 40), 1, 1, 30),
    ((30, 7), 1, 1, 5),
    ((90, 200), 25, 3, 4),        # 8 columns
    ((30, 64), 2, 10, 3),         # 3 rows per bin
    ((33, 61), 1, 11, 2),         # 61 columns, both borders, LDS tile
    ((47, 130), 3, 3, 32),        # per at the fused detector's cap
    ((47, 130), 3, 3, 33),        # ... and past it (response image + per-bin selection)
])
def test_detect_binned_geometries_bit_exact(viso, oracle, shape, bx, by, per):
    nf = per * bx * by
    for seed in (3, 4, 5):
        img = synth.make_images(seed, shape[0], shape[1])
        k0, r0 = oracle.detect_harris_binned(img, nf, bx, by)
        k1, r1 = libviso_amd.detect_harris_binned(img, nf, bx, by)
        assert np.array_equal(k0, k1) and np.array_equal(r0, r1)
        assert len(k0) > 0


# random geometries (tests/harris_fuzz.py).  Seed 1 holds the two cases that found the selection from a response image
# (more than 32 corners per bin, or bins wider than 62 columns) dropping corners: its second walk appended to the list
# with a per-lane count, and the lanes whose share of a bin is one pixel shorter never saw the last appends.
def test_detect_random_geometries(viso, oracle):
    import harris_fuzz
    for img, nf, bx, by in harris_fuzz.cases(1, 400):
        k0, r0 = oracle.detect_harris_binned(img, nf, bx, by)
        k1, r1 = libviso_amd.detect_harris_binned(img, nf, bx, by)
        assert np.array_equal(k0, k1) and np.array_equal(r0, r1), harris_fuzz.describe(img, nf, bx, by, k0, k1)
        assert np.array_equal(libviso_amd.harris_response(img), oracle.harris_response(img)), img.shape


# The strip kernel (waves over 58-column strips of a bin row, a candidate list per bin column, parts merged per bin; round 6:
# measured 11 % slower than the wave-per-bin kernel, so it is compiled into -DVISO_DEBUG_VARIANTS libraries only -- in the
# product build $VISO_HARRIS_STRIPS changes nothing and these tests run the one kernel twice) must give
# what the wave-per-bin kernel gives: forced by $VISO_HARRIS_STRIPS=1 for single images -- bin widths from 29 (the narrowest
# it takes: a strip then touches three bin columns) to 58, strips that end inside the last bin, bins in one part and in
# two, both image borders -- and chosen by the launcher itself for a batch large enough.
@pytest.mark.parametrize("shape,bx,by,per", [
    ((376, 1241), 24, 5, 10),     # the reference's geometry: 51-column bins, 22 strips, columns 1224..1240 in no bin
    ((376, 1241), 22, 5, 10),     # 56 columns
    ((376, 1241), 21, 5, 10),     # 59 columns: not the strip kernel's (LDS tile), the switch must not break it
    ((120, 400), 13, 3, 7),       # 30 columns: three bin columns per strip
    ((120, 400), 10, 2, 32),      # 40 columns, per at the cap
    ((90, 300), 6, 4, 5),         # 50 columns
    ((100, 58), 1, 2, 16),        # one bin = one strip, both borders
    ((64, 116), 2, 1, 20),        # 58 columns: bins and strips coincide
    ((50, 95), 3, 1, 9),          # 31 columns, 93 of 95 in bins: the second strip is 35 wide
    ((30, 64), 2, 10, 3),         # 3 rows per bin
])
def test_detect_strip_kernel_bit_exact(viso, oracle, shape, bx, by, per, monkeypatch):
    nf = per * bx * by
    for seed in (3, 4):
        img = synth.make_images(seed, shape[0], shape[1])
        k0, r0 = oracle.detect_harris_binned(img, nf, bx, by)
        monkeypatch.setenv("VISO_HARRIS_STRIPS", "0")
        ka, ra = libviso_amd.detect_harris_binned(img, nf, bx, by)
        monkeypatch.setenv("VISO_HARRIS_STRIPS", "1")
        kb, rb = libviso_amd.detect_harris_binned(img, nf, bx, by)
        assert np.array_equal(k0, ka) and np.array_equal(r0, ra)
        assert np.array_equal(k0, kb) and np.array_equal(r0, rb), (shape, bx, by, per)
        assert len(k0) > 0


def test_detect_strip_kernel_random_geometries(viso, oracle, monkeypatch):
    import harris_fuzz
    monkeypatch.setenv("VISO_HARRIS_STRIPS", "1")
    n_strip = 0
    for img, nf, bx, by in harris_fuzz.cases(7, 500):
        sx = img.shape[1] // bx
        if not (29 <= sx <= 58):      # the geometries the strip kernel takes
            continue
        n_strip += 1
        k0, r0 = oracle.detect_harris_binned(img, nf, bx, by)
        k1, r1 = libviso_amd.detect_harris_binned(img, nf, bx, by)
        assert np.array_equal(k0, k1) and np.array_equal(r0, r1), harris_fuzz.describe(img, nf, bx, by, k0, k1)
    assert n_strip >= 40


def test_batch_detect_takes_the_strip_kernel_when_it_pays(viso, oracle, monkeypatch):
    """A batch of enough images for 16384 strip waves: the launcher picks the strip kernel by itself; keypoints of sampled
    images against the oracle, and the whole batch against the wave-per-bin kernel."""
    monkeypatch.delenv("VISO_HARRIS_STRIPS", raising=False)
    rows, cols, nf = 120, 400, 98          # 13 x 12 bins of 30 x 10: 98 frames x 2 images x 7 strips x 12 bin rows = 16464 waves (>= 16384)
    rng = np.random.default_rng(5)
    base = [synth.make_images(100 + i, rows, cols) for i in range(6)]
    images = np.stack([np.stack([np.roll(base[(2 * t + s) % 6], (t + 3 * s) % 17, axis=1) for s in range(2)]) for t in range(nf)])
    got = {}
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("VISO_HARRIS_STRIPS", raising=False)
        else:
            monkeypatch.setenv("VISO_HARRIS_STRIPS", mode)
        ctx = libviso_amd.Context(0)
        b = libviso_amd.Batch(ctx, nf, 13 * 12 * 3)
        b.upload_images_only(images)
        b.detect(n_features=13 * 12 * 3, nbinx=13, nbiny=12)
        got[mode] = [b.keypoints(t, s) for t in range(nf) for s in range(2)]
        b.close(); ctx.close()
    for a, c in zip(got[None], got["0"]):
        assert np.array_equal(a, c)
    for t, s in ((0, 0), (17, 1), (97, 1)):
        k0, _ = oracle.detect_harris_binned(images[t, s], 13 * 12 * 3, 13, 12)
        assert np.array_equal(got[None][2 * t + s], k0)


def test_image_to_pose_pipeline(viso, oracle):
    seq = synth.make_image_sequence(9, 4, n_kp=1500)
    nf = 4
    # oracle: detect -> extract -> sequence, all on the CPU
    cap = 1200
    kp = np.zeros((nf, 2, cap, 2), np.float32)
    n = np.zeros((nf, 2), np.int32)
    desc = np.zeros((nf, 2, cap, 121), np.float32)
    for t in range(nf):
        for side in range(2):
            k, _ = oracle.detect_harris_binned(seq["images"][t, side])
            n[t, side] = len(k)
            kp[t, side, :len(k)] = k
            desc[t, side, :len(k)] = oracle.extract_descriptors(seq["images"][t, side], k)
    st, tm = MatchParams.stereo(seq["F"]), MatchParams.temporal()
    want = oracle.sequence(kp, desc, n, st, tm, seq["param"], seed=6)
    ctx = libviso_amd.Context(0)
    b = libviso_amd.Batch(ctx, nf, cap)
    b.upload_images_only(seq["images"])
    b.set_params(st, tm, seq["param"], seed=6)
    b.detect()
    b.run_images()
    for t in range(nf):
        for side in range(2):
            assert np.array_equal(b.keypoints(t, side), kp[t, side, :n[t, side]])
    tr, ok, n_inl = b.poses()
    sc, mo = b.counters()
    assert np.array_equal(ok, want["ok"]) and np.array_equal(n_inl, want["n_inl"])
    assert np.array_equal(sc, want["scored"]) and np.array_equal(mo, want["m_out"])
    assert ok[1:].all() and n_inl[1:].min() > 30
    for t in range(1, nf):
        a, r = libviso_amd.tr2mat(tr[t]), oracle.tr2mat(want["tr"][t])
        assert np.linalg.norm(a - r) / np.linalg.norm(r) < 1e-5
        assert np.abs(tr[t] - seq["tr_gt"][t]).max() < 5e-2
    b.close(); ctx.close()
