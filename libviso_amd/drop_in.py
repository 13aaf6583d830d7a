"""The literal drop-in flow, driven from numpy arrays (bench.py's `drop_in_per_call`, tests/test_gpu_drop_in.py).

The loop is C++ (libviso_host.so `viso_host_drop_in_run` -> `viso::sequence_odometry_per_call`, host/viso.hpp): the
patched reference's sequence_odometry (src/viso.cpp:1205-1327) calling the plain C-ABI of include/viso_hip.h once per
reference function per frame - match_desc x3, collect_matches, triangulate_rectified, match_circle,
ransac_minimize_reproj - host pointers in, host results out, one frame at a time.  This module is ctypes plumbing.
"""
import ctypes as C

import numpy as np

from . import load
from .abi import Param, f32p, f64p, i32p, ptr
from .kitti_shard import load_host

PLAIN_N = 7


class PlainTimes(C.Structure):
    _fields_ = [("calls", C.c_int64), ("host_us", C.c_double), ("h2d_us", C.c_double), ("kernel_us", C.c_double),
                ("d2h_us", C.c_double), ("wait_us", C.c_double)]


def _host():
    H = load_host()
    H.viso_host_drop_in_run.restype = C.c_int
    H.viso_host_drop_in_run.argtypes = [f32p, f32p, i32p, C.c_int, C.c_int, C.c_int, f64p, C.POINTER(Param), C.c_uint64,
                                        C.c_uint64, f64p, i32p, i32p, i32p, f64p, f64p]
    return H


def plain_profile(enable):
    """viso_plain_profile: bracket every plain-family call's phases with hipEvents (zeroes the sums when switched on)."""
    L = load()
    L.viso_plain_profile.argtypes = [C.c_int]
    L.viso_plain_profile(1 if enable else 0)


def plain_profile_rows():
    """{function name: {calls, host_us, h2d_us, kernel_us, d2h_us, wait_us}} - sums since profiling was switched on."""
    L = load()
    L.viso_plain_profile_get.argtypes = [C.c_int, C.POINTER(PlainTimes)]
    L.viso_plain_profile_name.restype = C.c_char_p
    L.viso_plain_profile_name.argtypes = [C.c_int]
    out = {}
    for fn in range(PLAIN_N):
        t = PlainTimes()
        L.viso_plain_profile_get(fn, C.byref(t))
        if t.calls:
            out[L.viso_plain_profile_name(fn).decode()] = {k: float(getattr(t, k)) for k in ("host_us", "h2d_us", "kernel_us", "d2h_us", "wait_us")} | {"calls": int(t.calls)}
    return out


def run(kp, desc, n, F, param, seed=0, first_frame=0, want_matches=False):
    """kp [nf][2][cap][2] f32, desc [nf][2][cap][dlen] f32, n [nf][2] i32 (the layout of Batch.upload).
    Returns dict(tr [nf][6], ok [nf], n_inl [nf], n_circle [nf], calls {name: (calls, wall us)}, loop_s, carry_s
    [, matches [3][nf] lists of (i1, i2, dist) arrays])."""
    H = _host()
    L = load()
    L.viso_plain_profile_name.restype = C.c_char_p
    L.viso_plain_profile_name.argtypes = [C.c_int]
    kp = np.ascontiguousarray(kp, np.float32)
    desc = np.ascontiguousarray(desc, np.float32)
    n = np.ascontiguousarray(n, np.int32)
    nf, _, cap, dlen = desc.shape
    Fa = np.ascontiguousarray(np.asarray(F, np.float64).reshape(9))
    rec = np.zeros((nf, 8), np.float64)
    m = np.zeros((3, nf, cap, 3), np.int32) if want_matches else None
    mn = np.zeros((3, nf), np.int32)
    nc = np.zeros(nf, np.int32)
    cu = np.zeros((PLAIN_N, 2), np.float64)
    ls = np.zeros(3, np.float64)
    r = H.viso_host_drop_in_run(ptr(kp, C.c_float), ptr(desc, C.c_float), ptr(n, C.c_int32), nf, cap, dlen,
                                ptr(Fa, C.c_double), C.byref(param), int(seed), int(first_frame), ptr(rec, C.c_double),
                                ptr(m, C.c_int32) if m is not None else None, ptr(mn, C.c_int32), ptr(nc, C.c_int32),
                                ptr(cu, C.c_double), ptr(ls, C.c_double))
    if r < 0:
        raise RuntimeError(f"viso_host_drop_in_run failed with {r}: {H.viso_host_last_error().decode()}")
    out = {"frames": int(r), "tr": rec[:, :6].copy(), "ok": rec[:, 6].astype(np.int32), "n_inl": rec[:, 7].astype(np.int32),
           "n_circle": nc, "match_n": mn, "loop_s": float(ls[0]), "carry_s": float(ls[1]),
           "calls": {L.viso_plain_profile_name(f).decode(): (int(cu[f, 0]), float(cu[f, 1])) for f in range(PLAIN_N) if cu[f, 0]}}
    if want_matches:
        out["matches"] = [[m[w, t, :mn[w, t]].copy() for t in range(nf)] for w in range(3)]
    return out


def plain_cache(enable):
    """viso_plain_cache: the plain family's image cache on / off (off: every image uploaded, sorted and packed again)."""
    r = load().viso_plain_cache(1 if enable else 0)
    if r != 1:
        raise RuntimeError(f"viso_plain_cache failed with {r}")


def plain_speculate(enable):
    """viso_plain_speculate: a frame's stereo call also runs what the loop asks for next (on), or every call direct (off)."""
    r = load().viso_plain_speculate(1 if enable else 0)
    if r != 1:
        raise RuntimeError(f"viso_plain_speculate failed with {r}")


def plain_stats():
    """dict(hits, misses, served [4], wasted [4], general_reruns) of the plain family's cache / frames."""
    L = load()
    L.viso_plain_cache_stats.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.viso_plain_general_reruns.restype = C.c_int64
    h, m = C.c_int64(0), C.c_int64(0)
    L.viso_plain_cache_stats(C.byref(h), C.byref(m))
    st = (C.c_int64 * 8)()
    L.viso_plain_speculate_stats(st)
    return {"hits": h.value, "misses": m.value, "served": list(st)[:4], "wasted": list(st)[4:],
            "general_reruns": int(L.viso_plain_general_reruns())}
