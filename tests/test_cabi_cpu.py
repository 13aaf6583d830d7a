"""No-GPU checks of the boundary: the C-ABI library loads, exports every
symbol include/viso_hip.h declares, host-only entry points compute the
reference's values, and device entry points fail loudly (never fall back to
CPU) when no HIP device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import libviso_amd
from libviso_amd import synth
from libviso_amd.abi import MatchParams, Param

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(libviso_amd.SO_PATH):
        libviso_amd.build()
    return libviso_amd.load()


def test_every_declared_symbol_is_exported(L):
    hdr = open(os.path.join(ROOT, "include", "viso_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(viso_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layout_matches_header(L):
    # sizes the C compiler produces for include/viso_hip.h
    assert C.sizeof(MatchParams) == 16 + 9 * 8 + 3 * 8
    assert C.sizeof(Param) == 8 + 8 + 5 * 8
    mp = MatchParams()
    F = np.arange(9, dtype=np.float64)
    L.viso_match_params_stereo(C.byref(mp), F.ctypes.data_as(C.POINTER(C.c_double)))
    ref = MatchParams.stereo(F)
    assert bytes(mp) == bytes(ref)          # MatchParams(F), reference src/viso.cpp:62-71
    L.viso_match_params_temporal(C.byref(mp))
    assert bytes(mp) == bytes(MatchParams.temporal())
    p = Param()
    L.viso_param_default(C.byref(p))
    assert bytes(p) == bytes(Param.default())


def test_host_entry_points_match_oracle(L, oracle):
    tr = np.array([0.01, -0.02, 0.03, 0.1, -0.2, 1.1])
    assert np.array_equal(libviso_amd.tr2mat(tr), oracle.tr2mat(tr))
    pose = oracle.pose_update(np.eye(4), tr * 0.5)
    assert np.allclose(libviso_amd.pose_update(pose, tr), oracle.pose_update(pose, tr), rtol=0, atol=1e-14)
    F = libviso_amd.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    Fo = oracle.F_from_P(synth.KITTI_P1, synth.KITTI_P2)
    assert np.allclose(F, Fo, rtol=1e-9, atol=1e-9)
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))]); P2 = P1.copy(); P2[0, 3] = 1
    assert np.array_equal(libviso_amd.F_from_P(P1, P2), np.array([[0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]]))
    # the host twin of the device's sampler (viso_sample3, csrc/solver_dev.h: the code ransac_hyp_kernel runs per lane) against
    # the oracle's: the same integers for every point count, seed and stream key (64-bit keys included)
    for m in (0, 2, 3, 4, 5, 10, 63, 64, 65, 500, 2047, 16384):
        for seed, frame in ((42, 7), (0, 0), (2**63 + 11, 2**40 + 3), (2**64 - 1, 2**64 - 1)):
            assert np.array_equal(libviso_amd.ransac_samples(seed, frame, 50, m), oracle.ransac_samples(seed, frame, 50, m)), (m, seed, frame)
    s = libviso_amd.ransac_samples(1, 2, 2000, 7)
    assert (s[:, 0] < s[:, 1]).all() and (s[:, 1] < s[:, 2]).all() and s.min() == 0 and s.max() == 6


def test_argument_errors_do_not_abort(L):
    n = C.c_int(0)
    mp = MatchParams.temporal()
    r = L.viso_match_desc(None, -1, None, 0, None, None, 121, C.byref(mp), None, C.byref(n))
    assert r == -1 and b"bad argument" in L.viso_last_error()


def test_device_calls_fail_loudly_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rng = np.random.default_rng(0)
    kp = rng.integers(0, 50, (8, 2)).astype(np.float32)
    d = rng.integers(-5, 5, (8, 121)).astype(np.float32)
    with pytest.raises(libviso_amd.VisoError):
        libviso_amd.match_desc(kp, kp, d, d, MatchParams.temporal())
    with pytest.raises(libviso_amd.VisoError):
        libviso_amd.Context(0)


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure; the shipped path must not reach it
    for base, _, files in os.walk(os.path.join(ROOT, "libviso_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="replace").read()
                bad = re.findall(r"^\s*(?:from|import)\s+oracle|#include\s+[\"<][^\n]*oracle|libviso_oracle|oracle_[a-z_0-9]+\s*\(",
                                 txt, flags=re.M)
                assert not bad, (os.path.join(base, f), bad)


def test_cpp_host_mirror_selftest(tmp_path):
    """C++ host mirror (libviso_amd/host): Procrustes known answer of the
    reference's test/test.cpp:171-205, F_from_P (src/mvg.cpp:73-89), tr2mat and
    the KITTI calib/pose file formats (src/kitti.cpp:23-64).  Host-only code."""
    import subprocess
    exe = os.path.join(ROOT, "libviso_amd", "viso_host_selftest")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "libviso_amd", "host"), "-s"])
    # PNG decoder (KITTI images are 8-bit grayscale PNG, src/kitti.cpp:108-110): every filter type,
    # stored / fixed / dynamic deflate blocks, and an RGB image, each against the same pixels as PGM
    import struct
    import zlib
    rng = np.random.default_rng(0)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)

    def write_png(path, img, level, ftypes):
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        flat = img.reshape(h, w * ch).astype(np.int32)
        raw = bytearray()
        for y in range(h):
            ft = ftypes[y % len(ftypes)]
            cur, up = flat[y], (flat[y - 1] if y else np.zeros(w * ch, np.int32))
            a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
            c = np.concatenate([np.zeros(ch, np.int32), up[:-ch]])
            if ft == 0: f = cur
            elif ft == 1: f = cur - a
            elif ft == 2: f = cur - up
            elif ft == 3: f = cur - ((a + up) >> 1)
            else:
                pp = a + up - c
                pa, pb, pc = np.abs(pp - a), np.abs(pp - up), np.abs(pp - c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
                f = cur - pred
            raw += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
        comp = zlib.compressobj(level, zlib.DEFLATED, 15, 9, zlib.Z_FIXED if level == 1 else zlib.Z_DEFAULT_STRATEGY)
        z = comp.compress(bytes(raw)) + comp.flush()
        ihdr = struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2}[ch], 0, 0, 0)
        half = len(z) // 2                                     # two IDAT chunks
        with open(path, "wb") as fo:
            fo.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b""))

    def write_pgm(path, gray):
        with open(path, "wb") as fo:
            fo.write(b"P5\n%d %d\n255\n" % (gray.shape[1], gray.shape[0]) + gray.tobytes())

    args = []
    smooth = np.clip(np.cumsum(rng.normal(0, 3, (61, 97)), 1) + 128, 0, 255).astype(np.uint8)
    for i, (level, ft) in enumerate(((0, [0]), (1, [1, 2]), (6, [0, 1, 2, 3, 4]), (9, [4]), (6, [3]))):
        write_png(str(tmp_path / f"g{i}.png"), smooth, level, ft)
        write_pgm(str(tmp_path / f"g{i}.pgm"), smooth)
        args += [str(tmp_path / f"g{i}.png"), str(tmp_path / f"g{i}.pgm")]
    rgb = rng.integers(0, 256, (20, 33, 3)).astype(np.uint8)
    write_png(str(tmp_path / "c.png"), rgb, 6, [0, 4])
    r64 = rgb.astype(np.int64)
    g = ((r64[..., 0] * 4899 + r64[..., 1] * 9617 + r64[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)
    write_pgm(str(tmp_path / "c.pgm"), g)
    args += [str(tmp_path / "c.png"), str(tmp_path / "c.pgm")]
    # hostile files: every one must be refused, quickly and without a large allocation
    sig = b"\x89PNG\r\n\x1a\n"
    good_ihdr = struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 0)
    z44 = zlib.compress(bytes(4 * 5))
    bad = {
        "short_ihdr.png": sig + chunk(b"IHDR", good_ihdr[:5]) + chunk(b"IDAT", z44) + chunk(b"IEND", b""),
        "ihdr_not_first.png": sig + chunk(b"tEXt", b"x") + chunk(b"IHDR", good_ihdr) + chunk(b"IDAT", z44) + chunk(b"IEND", b""),
        "two_ihdr.png": sig + chunk(b"IHDR", good_ihdr) + chunk(b"IHDR", good_ihdr) + chunk(b"IDAT", z44) + chunk(b"IEND", b""),
        "huge_header.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 65535, 65535, 8, 6, 0, 0, 0)) + chunk(b"IDAT", z44) + chunk(b"IEND", b""),
        "bomb.png": sig + chunk(b"IHDR", good_ihdr) + chunk(b"IDAT", zlib.compress(bytes(32 << 20), 9)) + chunk(b"IEND", b""),
        "bad_nlen.png": sig + chunk(b"IHDR", good_ihdr) + chunk(b"IDAT", b"\x78\x01\x01\x14\x00\x00\x00" + bytes(20)) + chunk(b"IEND", b""),
        "truncated.png": (sig + chunk(b"IHDR", good_ihdr) + chunk(b"IDAT", z44))[:-6],
        "len_overflow.png": sig + chunk(b"IHDR", good_ihdr) + struct.pack(">I", 0xfffffff0) + b"IDAT" + z44,
    }
    args.append("--bad")
    for name, data in bad.items():
        with open(str(tmp_path / name), "wb") as fo:
            fo.write(data)
        args.append(str(tmp_path / name))
    # a VALID 4x4 file built the same way, so that the refusals above are not an artefact of the construction
    with open(str(tmp_path / "ok44.png"), "wb") as fo:
        fo.write(sig + chunk(b"IHDR", good_ihdr) + chunk(b"IDAT", z44) + chunk(b"IEND", b""))
    write_pgm(str(tmp_path / "ok44.pgm"), np.zeros((4, 4), np.uint8))
    args = [str(tmp_path / "ok44.png"), str(tmp_path / "ok44.pgm")] + args
    r = subprocess.run([exe, str(tmp_path)] + args, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stdout + r.stderr


def test_png_decoder_survives_hostile_files_under_sanitizers(tmp_path):
    """The KITTI runner's own PNG decoder (libviso_amd/host/png_read.hpp: table-driven inflate on raw pointers) reads
    files it does not trust.  tools/png_fuzz.cpp, built with AddressSanitizer + UBSan for the CPU, damages a valid
    8-bit grayscale PNG 1500 times (bit flips, truncation, rewritten chunk lengths, runs of 0x00 / 0xff and random
    bytes inside the IDAT payload) and decodes every variant: refused or an image of the header's size, never a
    report, never a byte written past a caller's buffer."""
    import shutil
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pngutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "png_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(ROOT, "tools", "png_fuzz.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(3)
    img = np.clip(np.cumsum(rng.normal(0, 3, (96, 160)), 1) + 128, 0, 255).astype(np.uint8)
    seed = str(tmp_path / "seed.png")
    pngutil.write_gray_png(seed, img)
    r = subprocess.run([exe, seed, "1500"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "png_fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
