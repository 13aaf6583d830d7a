"""ctypes wrapper of oracle/libviso_oracle.so — TEST INFRASTRUCTURE ONLY.

Builds the library with `make -C oracle` on first use when gcc is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from libviso_amd.abi import (MatchParams, Param, declare_common, f32p, f64p, i32p, i64p, intp,
                             ptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libviso_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "viso_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        MP, PP = declare_common(L, "oracle_")
        L.oracle_radius_search.restype = C.c_int
        L.oracle_radius_search.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_float, C.c_int, i32p, i32p]
        L.oracle_sampson_distance.restype = C.c_double
        L.oracle_sampson_distance.argtypes = [f64p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.oracle_match_desc.restype = C.c_int
        L.oracle_match_desc.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p, f32p, C.c_int, MP, i32p,
                                        intp, i64p]
        L.oracle_compute_J.restype = None
        L.oracle_compute_J.argtypes = [f64p, f64p, C.c_int, f64p, PP, i32p, C.c_int, f64p, f64p, f64p]
        L.oracle_minimize_reproj.restype = C.c_int
        L.oracle_minimize_reproj.argtypes = [f64p, f64p, C.c_int, f64p, PP, i32p, C.c_int, intp]
        L.oracle_lu_solve6.restype = C.c_int
        L.oracle_lu_solve6.argtypes = [f64p, f64p]
        L.oracle_ransac_samples_algorithm_s.restype = None
        L.oracle_ransac_samples_algorithm_s.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, i32p]
        L.oracle_harris_response.restype = C.c_int
        L.oracle_harris_response.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, f32p]
        L.oracle_harris_response_v1.restype = C.c_int
        L.oracle_harris_response_v1.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, f32p]
        L.oracle_detect_harris_binned.restype = C.c_int
        L.oracle_detect_harris_binned.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_double, f32p, f32p, intp]
        L.oracle_sequence.restype = C.c_int
        L.oracle_sequence.argtypes = [f32p, f32p, i32p, C.c_int, C.c_int, C.c_int, MP, MP, PP,
                                      C.c_uint64, C.c_uint64, C.c_int, f64p, i32p, i32p, i64p, i64p,
                                      f64p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def radius_search(kp1, kp2, radius, K):
    kp1, kp2 = _f32(kp1).reshape(-1, 2), _f32(kp2).reshape(-1, 2)
    nei = np.empty((len(kp1), K), np.int32)
    found = np.empty(len(kp1), np.int32)
    lib().oracle_radius_search(ptr(kp1, C.c_float), len(kp1), ptr(kp2, C.c_float), len(kp2),
                               C.c_float(radius), K, ptr(nei, C.c_int32), ptr(found, C.c_int32))
    return nei, found


def sampson_distance(F, p1, p2):
    F = _f64(F).reshape(9)
    return lib().oracle_sampson_distance(ptr(F, C.c_double), p1[0], p1[1], p2[0], p2[1])


def match_desc(kp1, kp2, d1, d2, mp, return_scored=False):
    kp1, kp2 = _f32(kp1).reshape(-1, 2), _f32(kp2).reshape(-1, 2)
    d1, d2 = _f32(d1), _f32(d2)
    n1, n2 = len(kp1), len(kp2)
    dlen = d1.shape[1] if d1.ndim == 2 else d2.shape[1]
    out = np.empty((max(n1, 1), 3), np.int32)
    n = C.c_int(0)
    sc = C.c_int64(0)
    r = lib().oracle_match_desc(ptr(kp1, C.c_float), n1, ptr(kp2, C.c_float), n2,
                                ptr(d1, C.c_float), ptr(d2, C.c_float), dlen, C.byref(mp),
                                ptr(out, C.c_int32), C.byref(n), C.byref(sc))
    assert r == 1, r
    m = out[:n.value].copy()
    return (m, sc.value) if return_scored else m


def match_circle(lr, lr_prev, m11, m22, cap=None):
    lr, lr_prev, m11, m22 = (_i32(a).reshape(-1, 3) for a in (lr, lr_prev, m11, m22))
    cap = cap if cap is not None else max(1, len(lr) * 4)
    circ = np.empty((cap, 4), np.int32)
    pcl = np.empty((cap, 2), np.int32)
    n = C.c_int(0)
    r = lib().oracle_match_circle(ptr(lr, C.c_int32), len(lr), ptr(lr_prev, C.c_int32), len(lr_prev),
                                  ptr(m11, C.c_int32), len(m11), ptr(m22, C.c_int32), len(m22),
                                  ptr(circ, C.c_int32), ptr(pcl, C.c_int32), cap, C.byref(n))
    return r, circ[:min(n.value, cap)].copy(), pcl[:min(n.value, cap)].copy(), n.value


def collect_matches(kp1, kp2, match):
    kp1, kp2 = _f32(kp1).reshape(-1, 2), _f32(kp2).reshape(-1, 2)
    match = _i32(match).reshape(-1, 3)
    x = np.empty((4, len(match)), np.float64)
    r = lib().oracle_collect_matches(ptr(kp1, C.c_float), len(kp1), ptr(kp2, C.c_float), len(kp2),
                                     ptr(match, C.c_int32), len(match), ptr(x, C.c_double))
    assert r == 1
    return x


def triangulate_rectified(x, param):
    x = _f64(x)
    X = np.empty((3, x.shape[1]), np.float64)
    lib().oracle_triangulate_rectified(ptr(x, C.c_double), x.shape[1], C.byref(param), ptr(X, C.c_double))
    return X


def compute_J(X, obs, tr, param, active):
    X, obs, tr, active = _f64(X), _f64(obs), _f64(tr), _i32(active)
    n = len(active)
    J = np.empty((4 * n, 6)); pred = np.empty((4, n)); res = np.empty(4 * n)
    lib().oracle_compute_J(ptr(X, C.c_double), ptr(obs, C.c_double), X.shape[1], ptr(tr, C.c_double),
                           C.byref(param), ptr(active, C.c_int32), n, ptr(J, C.c_double),
                           ptr(pred, C.c_double), ptr(res, C.c_double))
    return J, pred, res


def minimize_reproj(X, obs, tr, param, active):
    X, obs, active = _f64(X), _f64(obs), _i32(active)
    tr = _f64(tr).copy()
    it = C.c_int(0)
    ok = lib().oracle_minimize_reproj(ptr(X, C.c_double), ptr(obs, C.c_double), X.shape[1],
                                      ptr(tr, C.c_double), C.byref(param), ptr(active, C.c_int32),
                                      len(active), C.byref(it))
    return ok, tr, it.value


def get_inliers(X, obs, tr, param):
    X, obs, tr = _f64(X), _f64(obs), _f64(tr)
    m = X.shape[1]
    inl = np.empty(max(m, 1), np.int32)
    n = C.c_int(0)
    rms = C.c_double(0)
    lib().oracle_get_inliers(ptr(X, C.c_double), ptr(obs, C.c_double), m, ptr(tr, C.c_double),
                             C.byref(param), ptr(inl, C.c_int32), C.byref(n), C.byref(rms))
    return inl[:n.value].copy(), rms.value


def ransac_samples(seed, frame, iters, m, algorithm_s=False):
    out = np.empty((iters, 3), np.int32)
    (lib().oracle_ransac_samples_algorithm_s if algorithm_s else lib().oracle_ransac_samples)(seed, frame, iters, m, ptr(out, C.c_int32))
    return out


def ransac_minimize_reproj(X, obs, param, samples=None, seed=0, frame=0, tr0=None):
    X, obs = _f64(X), _f64(obs)
    m = X.shape[1]
    tr = np.zeros(6) if tr0 is None else _f64(tr0).copy()
    inl = np.empty(max(m, 1), np.int32)
    n = C.c_int(0)
    s = None if samples is None else _i32(samples)
    ok = lib().oracle_ransac_minimize_reproj(ptr(X, C.c_double), ptr(obs, C.c_double), m,
                                             ptr(tr, C.c_double), ptr(inl, C.c_int32), C.byref(n),
                                             C.byref(param), ptr(s, C.c_int32), seed, frame)
    return ok, tr, inl[:n.value].copy()


def lu_solve6(A, b):
    A, b = _f64(A).copy().reshape(36), _f64(b).copy().reshape(6)
    ok = lib().oracle_lu_solve6(ptr(A, C.c_double), ptr(b, C.c_double))
    return ok, b


def tr2mat(tr):
    tr = _f64(tr)
    T = np.empty((4, 4))
    lib().oracle_tr2mat(ptr(tr, C.c_double), ptr(T, C.c_double))
    return T


def pose_update(pose, tr):
    pose, tr = _f64(pose), _f64(tr)
    out = np.empty((4, 4))
    lib().oracle_pose_update(ptr(pose, C.c_double), ptr(tr, C.c_double), ptr(out, C.c_double))
    return out


def F_from_P(P1, P2):
    P1, P2 = _f64(P1), _f64(P2)
    F = np.empty((3, 3))
    lib().oracle_F_from_P(ptr(P1, C.c_double), ptr(P2, C.c_double), ptr(F, C.c_double))
    return F


def extract_descriptors(img, kp, radius=5):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    kp = _f32(kp).reshape(-1, 2)
    d = np.empty((len(kp), (2 * radius + 1) ** 2), np.float32)
    r = lib().oracle_extract_descriptors(ptr(img, C.c_uint8), img.shape[0], img.shape[1],
                                         ptr(kp, C.c_float), len(kp), radius, ptr(d, C.c_float))
    assert r == 1
    return d


HARRIS_K = float(np.float32(0.04))   # the reference's intended default (float k = .04, src/viso.cpp:915)


def harris_response(img, k=HARRIS_K):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    r = np.empty(img.shape, np.float32)
    assert lib().oracle_harris_response(ptr(img, C.c_uint8), img.shape[0], img.shape[1], k, ptr(r, C.c_float)) == 1
    return r


def harris_response_v1(img, k=HARRIS_K):
    """The rounds 1-3 restatement (exact integer Sobel sums times the scale): kept for comparison."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    r = np.empty(img.shape, np.float32)
    assert lib().oracle_harris_response_v1(ptr(img, C.c_uint8), img.shape[0], img.shape[1], k, ptr(r, C.c_float)) == 1
    return r


def detect_harris_binned(img, n_features=1200, nbinx=24, nbiny=5, k=HARRIS_K):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    kp = np.empty((max(1, n_features), 2), np.float32)
    resp = np.empty(max(1, n_features), np.float32)
    n = C.c_int(0)
    r = lib().oracle_detect_harris_binned(ptr(img, C.c_uint8), img.shape[0], img.shape[1], n_features, nbinx, nbiny,
                                          k, ptr(kp, C.c_float), ptr(resp, C.c_float), C.byref(n))
    assert r == 1, r
    return kp[:n.value].copy(), resp[:n.value].copy()


def sequence(kp, desc, n, stereo, temporal, param, seed=0, first_frame=0, matcher_only=False):
    """oracle_sequence over frames laid out like viso_batch. Returns dict."""
    kp, desc, n = _f32(kp), _f32(desc), _i32(n)
    nf, _, cap, _ = kp.shape
    dlen = desc.shape[-1]
    tr = np.zeros((nf, 6)); ok = np.zeros(nf, np.int32); ninl = np.zeros(nf, np.int32)
    scored = np.zeros((3, nf), np.int64); mout = np.zeros((3, nf), np.int64)
    st = np.zeros(5)   # match_desc total, circle, gather + triangulate, RANSAC/GN, neighbour-search share of [0]
    r = lib().oracle_sequence(ptr(kp, C.c_float), ptr(desc, C.c_float), ptr(n, C.c_int32), nf, cap,
                              dlen, C.byref(stereo), C.byref(temporal), C.byref(param), seed,
                              first_frame, int(matcher_only), ptr(tr, C.c_double), ptr(ok, C.c_int32),
                              ptr(ninl, C.c_int32), ptr(scored, C.c_int64), ptr(mout, C.c_int64),
                              ptr(st, C.c_double))
    assert r == 1
    return dict(tr=tr, ok=ok, n_inl=ninl, scored=scored, m_out=mout, stage_s=st)
