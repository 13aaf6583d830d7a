"""libviso_amd — MI355X (gfx950) implementation of libviso's per-frame hot path.

The product is the C-ABI shared library built from libviso_amd/csrc
(libviso_hip.so, declared in include/viso_hip.h) plus the C++ host mirror of
the reference interface in libviso_amd/host.  This Python package is plumbing
for tests and bench.py: a ctypes loader and numpy-in/numpy-out wrappers with
the reference's function names.  There is NO CPU fallback: if the library is
missing, or no HIP device is present, calls raise.
"""
import atexit
import ctypes as C
import os
import subprocess
import sys
import weakref

import numpy as np

from .abi import (DESC_LEN, MatchParams, Param, declare_common, f32p, f64p, i32p, i64p, intp, ptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VISO_HIP_SO") or os.path.join(_HERE, "libviso_hip.so")   # VISO_HIP_SO: another build of the library (A/B runs)
CSRC = os.path.join(_HERE, "csrc")


class VisoError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", "6"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise VisoError("building libviso_hip.so failed")
    return SO_PATH


_lib = None


def load():
    """ctypes handle of libviso_hip.so; raises when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise VisoError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)")
    try:
        # torch ships its own libamdhip64 (same soname): load it first so this
        # process holds exactly one HIP runtime and pointers/streams can be shared.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    L = C.CDLL(SO_PATH)
    MP, PP = declare_common(L, "viso_")
    L.viso_last_error.restype = C.c_char_p
    L.viso_version.restype = C.c_char_p
    L.viso_ctx_matcher_kernel_name.restype = C.c_char_p
    L.viso_ctx_matcher_kernel_name.argtypes = [C.c_void_p]
    L.viso_ctx_set_matcher.argtypes = [C.c_void_p, C.c_int]
    L.viso_matcher_variants.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.viso_ctx_set_gn_split.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "viso_ctx_set_row8_shift"):   # (absent from older builds of the library: VISO_HIP_SO A/B runs)
        L.viso_ctx_set_row8_shift.argtypes = [C.c_void_p, C.c_int]
        L.viso_batch_get_row8_shift.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.viso_batch_get_hypotheses.argtypes = [C.c_void_p, f64p, i32p, i32p, i32p]
    L.viso_host_alloc.restype = C.c_void_p
    L.viso_host_alloc.argtypes = [C.c_size_t]
    L.viso_host_free.argtypes = [C.c_void_p]
    L.viso_batch_upload_async.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, f32p, i32p]
    L.viso_batch_upload_images_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                                 f32p, i32p]
    L.viso_batch_upload_i16.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, C.POINTER(C.c_int16), i32p]
    L.viso_batch_upload_i16_async.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, C.POINTER(C.c_int16), i32p]
    L.viso_match_desc.restype = C.c_int
    L.viso_match_desc.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p, f32p, C.c_int, MP, i32p, intp]
    L.viso_minimize_reproj.restype = C.c_int
    L.viso_minimize_reproj.argtypes = [f64p, f64p, C.c_int, f64p, PP, i32p, C.c_int]
    L.viso_match_params_stereo.argtypes = [MP, f64p]
    L.viso_match_params_temporal.argtypes = [MP]
    L.viso_param_default.argtypes = [PP]
    L.viso_ctx_create.restype = C.c_void_p
    L.viso_ctx_create.argtypes = [C.c_int, C.c_void_p]
    L.viso_ctx_destroy.argtypes = [C.c_void_p]
    L.viso_ctx_stream.restype = C.c_void_p
    L.viso_ctx_stream.argtypes = [C.c_void_p]
    L.viso_ctx_synchronize.argtypes = [C.c_void_p]
    L.viso_batch_create.restype = C.c_void_p
    L.viso_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.viso_batch_destroy.argtypes = [C.c_void_p]
    L.viso_batch_upload.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, f32p, i32p]
    L.viso_batch_device_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p)]
    L.viso_batch_set_params.argtypes = [C.c_void_p, MP, MP, PP, C.c_uint64, C.c_uint64]
    L.viso_batch_run_matcher.argtypes = [C.c_void_p]
    L.viso_batch_run.argtypes = [C.c_void_p]
    L.viso_batch_get_matches.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, intp]
    L.viso_batch_get_circle.argtypes = [C.c_void_p, C.c_int, i32p, i32p, intp]
    L.viso_batch_get_pose.argtypes = [C.c_void_p, C.c_int, f64p, intp, i32p, intp]
    L.viso_batch_get_poses.argtypes = [C.c_void_p, f64p, i32p, i32p]
    L.viso_batch_get_counters.argtypes = [C.c_void_p, i64p, i64p]
    L.viso_batch_get_general_path_flags.argtypes = [C.c_void_p, i32p]
    L.viso_batch_get_overflow_count.argtypes = [C.c_void_p, i32p]
    L.viso_batch_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    L.viso_batch_kernel_ms.argtypes = [C.c_void_p, f64p, intp]
    L.viso_batch_upload_images.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                           f32p, i32p]
    L.viso_batch_run_images.argtypes = [C.c_void_p, C.c_int]
    L.viso_batch_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
    L.viso_batch_get_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, intp]
    L.viso_harris_response.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, f32p]
    L.viso_detect_harris_binned.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_double, f32p, f32p, intp]
    _lib = L
    return L


def _err(where, code):
    msg = load().viso_last_error().decode(errors="replace")
    raise VisoError(f"{where} failed with {code}: {msg}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)




def matcher_variants():
    """What this build of libviso_hip.so offers (viso_ctx_set_matcher): (3, 5, 6) for the product build —
    match_union_kernel, match_prune_kernel and match_union8_kernel (6, the default) — and (2, 3, 4, 5, 6) for
    `make DEBUG_VARIANTS=1`.  Asked of the library itself, so a debug build gets its variants tested."""
    out = (C.c_int * 8)()
    n = load().viso_matcher_variants(out, 8)
    return tuple(out[i] for i in range(min(n, 8)))


def __getattr__(name):   # MATCHER_VARIANTS / DEFAULT_MATCHER: resolved on first use (need the library, not a device)
    if name == "MATCHER_VARIANTS":
        return matcher_variants()
    if name == "DEFAULT_MATCHER":   # the build's default, or $VISO_MATCHER (viso_matcher_default)
        L = load()   # builds older than viso_matcher_default (VISO_HIP_SO A/B runs) started every context with variant 3
        return int(L.viso_matcher_default()) if hasattr(L, "viso_matcher_default") else 3
    raise AttributeError(name)


def set_matcher_variant(v, ctx=None):
    """Which kernel takes the temporal calls: of `ctx`, or of the plain family's default context."""
    r = load().viso_ctx_set_matcher(ctx.h if ctx is not None else None, int(v))
    if r != 1:
        _err("viso_ctx_set_matcher", r)


def set_row8_shift(shift, ctx=None):
    """viso_ctx_set_row8_shift: the shift of match_union8_kernel's 8-bit planes, -1 = chosen from the data (default), 0..3 = fixed."""
    r = load().viso_ctx_set_row8_shift(ctx.h if ctx is not None else None, int(shift))
    if r != 1:
        _err("viso_ctx_set_row8_shift", r)


def set_gn_split(split, ctx=None):
    """viso_ctx_set_gn_split: iterations of a 3-point hypothesis done by the lane-per-hypothesis kernel (0 = default)."""
    r = load().viso_ctx_set_gn_split(ctx.h if ctx is not None else None, int(split))
    if r != 1:
        _err("viso_ctx_set_gn_split", r)


def matcher_kernel_name(ctx=None):
    return load().viso_ctx_matcher_kernel_name(ctx.h if ctx is not None else None).decode()


# Handles that are still open when the interpreter exits are closed HERE, from an atexit hook: that runs before
# the HIP runtime's own teardown, whereas a __del__ during interpreter finalisation can run after it (calling
# hipStreamSynchronize then aborts the process from inside the runtime).  Batches first, then contexts.
_live = weakref.WeakSet()


def _close_all():
    objs = list(_live)
    for o in sorted(objs, key=lambda o: isinstance(o, Context)):
        try:
            o.close()
        except Exception:
            pass


atexit.register(_close_all)


class PinnedArray:
    """numpy view of hipHostMalloc memory (viso_host_alloc) for the *_async uploads."""

    def __init__(self, shape, dtype):
        self.L = load()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.p = self.L.viso_host_alloc(self.nbytes)
        if not self.p:
            raise VisoError("viso_host_alloc: " + self.L.viso_last_error().decode())
        buf = (C.c_char * self.nbytes).from_address(self.p)
        self.a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        _live.add(self)

    def close(self):
        if self.p:
            self.a = None
            self.L.viso_host_free(self.p)
            self.p = None

    def __del__(self, _finalizing=sys.is_finalizing):   # bound at definition: module globals are None late in shutdown
        if not _finalizing():
            try:
                self.close()
            except Exception:
                pass


# ------------------------------------------------------------ plain family
def match_desc(kp1, kp2, d1, d2, mp):
    """match_desc, reference src/viso.cpp:669-726 -> (M,3) int32 (i1,i2,dist)."""
    L = load()
    kp1, kp2 = _f32(kp1).reshape(-1, 2), _f32(kp2).reshape(-1, 2)
    d1, d2 = _f32(d1), _f32(d2)
    n1, n2 = len(kp1), len(kp2)
    dlen = d1.shape[1] if d1.ndim == 2 else d2.shape[1]
    out = np.empty((max(n1, 1), 3), np.int32)
    n = C.c_int(0)
    r = L.viso_match_desc(ptr(kp1, C.c_float), n1, ptr(kp2, C.c_float), n2, ptr(d1, C.c_float),
                          ptr(d2, C.c_float), dlen, C.byref(mp), ptr(out, C.c_int32), C.byref(n))
    if r != 1:
        _err("viso_match_desc", r)
    return out[:n.value].copy()


def match_circle(lr, lr_prev, m11, m22, cap=None):
    L = load()
    lr, lr_prev, m11, m22 = (_i32(a).reshape(-1, 3) for a in (lr, lr_prev, m11, m22))
    cap = cap if cap is not None else max(1, len(lr) * 4)
    circ = np.empty((cap, 4), np.int32)
    pcl = np.empty((cap, 2), np.int32)
    n = C.c_int(0)
    r = L.viso_match_circle(ptr(lr, C.c_int32), len(lr), ptr(lr_prev, C.c_int32), len(lr_prev),
                            ptr(m11, C.c_int32), len(m11), ptr(m22, C.c_int32), len(m22),
                            ptr(circ, C.c_int32), ptr(pcl, C.c_int32), cap, C.byref(n))
    if r < 0 and r != -1:
        _err("viso_match_circle", r)
    k = min(n.value, cap)
    return r, circ[:k].copy(), pcl[:k].copy(), n.value


def collect_matches(kp1, kp2, match):
    L = load()
    kp1, kp2 = _f32(kp1).reshape(-1, 2), _f32(kp2).reshape(-1, 2)
    match = _i32(match).reshape(-1, 3)
    x = np.empty((4, len(match)), np.float64)
    r = L.viso_collect_matches(ptr(kp1, C.c_float), len(kp1), ptr(kp2, C.c_float), len(kp2),
                               ptr(match, C.c_int32), len(match), ptr(x, C.c_double))
    if r != 1:
        _err("viso_collect_matches", r)
    return x


def triangulate_rectified(x, param):
    L = load()
    x = _f64(x)
    X = np.empty((3, x.shape[1]), np.float64)
    r = L.viso_triangulate_rectified(ptr(x, C.c_double), x.shape[1], C.byref(param), ptr(X, C.c_double))
    if r != 1:
        _err("viso_triangulate_rectified", r)
    return X


def minimize_reproj(X, obs, tr, param, active):
    L = load()
    X, obs, active = _f64(X), _f64(obs), _i32(active)
    tr = _f64(tr).copy()
    r = L.viso_minimize_reproj(ptr(X, C.c_double), ptr(obs, C.c_double), X.shape[1],
                               ptr(tr, C.c_double), C.byref(param), ptr(active, C.c_int32), len(active))
    if r < 0:
        _err("viso_minimize_reproj", r)
    return r, tr


def get_inliers(X, obs, tr, param):
    L = load()
    X, obs, tr = _f64(X), _f64(obs), _f64(tr)
    m = X.shape[1]
    inl = np.empty(max(m, 1), np.int32)
    n = C.c_int(0)
    rms = C.c_double(0)
    r = L.viso_get_inliers(ptr(X, C.c_double), ptr(obs, C.c_double), m, ptr(tr, C.c_double),
                           C.byref(param), ptr(inl, C.c_int32), C.byref(n), C.byref(rms))
    if r != 1:
        _err("viso_get_inliers", r)
    return inl[:n.value].copy(), rms.value


def support_sizes(X, obs, tr_h, param):
    """Support size of every motion tr_h[k] over (X, obs), through the RANSAC stage's counting kernel."""
    L = load()
    X, obs, tr_h = _f64(X), _f64(obs), _f64(np.atleast_2d(tr_h))
    cnt = np.zeros(max(len(tr_h), 1), np.int32)
    r = L.viso_support_sizes(ptr(X, C.c_double), ptr(obs, C.c_double), X.shape[1], ptr(tr_h, C.c_double), len(tr_h),
                             C.byref(param), ptr(cnt, C.c_int32))
    if r != 1:
        _err("viso_support_sizes", r)
    return cnt[:len(tr_h)].copy()


def ransac_samples(seed, frame, iters, m):
    out = np.empty((iters, 3), np.int32)
    load().viso_ransac_samples(seed, frame, iters, m, ptr(out, C.c_int32))
    return out


def ransac_minimize_reproj(X, obs, param, samples=None, seed=0, frame=0, tr0=None):
    L = load()
    X, obs = _f64(X), _f64(obs)
    m = X.shape[1]
    tr = np.zeros(6) if tr0 is None else _f64(tr0).copy()
    inl = np.empty(max(m, 1), np.int32)
    n = C.c_int(0)
    s = None if samples is None else _i32(samples)
    r = L.viso_ransac_minimize_reproj(ptr(X, C.c_double), ptr(obs, C.c_double), m, ptr(tr, C.c_double),
                                      ptr(inl, C.c_int32), C.byref(n), C.byref(param),
                                      ptr(s, C.c_int32), seed, frame)
    if r < 0:
        _err("viso_ransac_minimize_reproj", r)
    return r, tr, inl[:n.value].copy()


def tr2mat(tr):
    tr = _f64(tr)
    T = np.empty((4, 4))
    load().viso_tr2mat(ptr(tr, C.c_double), ptr(T, C.c_double))
    return T


def pose_update(pose, tr):
    pose, tr = _f64(pose), _f64(tr)
    out = np.empty((4, 4))
    load().viso_pose_update(ptr(pose, C.c_double), ptr(tr, C.c_double), ptr(out, C.c_double))
    return out


def F_from_P(P1, P2):
    P1, P2 = _f64(P1), _f64(P2)
    F = np.empty((3, 3))
    load().viso_F_from_P(ptr(P1, C.c_double), ptr(P2, C.c_double), ptr(F, C.c_double))
    return F


def extract_descriptors(img, kp, radius=5):
    """MyFeatureExtractor::computeImpl, reference src/viso.cpp:1004-1024."""
    L = load()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    kp = _f32(kp).reshape(-1, 2)
    d = np.empty((len(kp), (2 * radius + 1) ** 2), np.float32)
    r = L.viso_extract_descriptors(ptr(img, C.c_uint8), img.shape[0], img.shape[1], ptr(kp, C.c_float),
                                   len(kp), radius, ptr(d, C.c_float))
    if r != 1:
        _err("viso_extract_descriptors", r)
    return d


HARRIS_K = float(np.float32(0.04))   # the reference's intended default (float k = .04, src/viso.cpp:915)


def harris_response(img, k=HARRIS_K):
    """cv::cornerHarris(img, R, 3, 5, k, BORDER_DEFAULT) restated (reference src/viso.cpp:930)."""
    L = load()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    r = np.empty(img.shape, np.float32)
    rc = L.viso_harris_response(ptr(img, C.c_uint8), img.shape[0], img.shape[1], k, ptr(r, C.c_float))
    if rc != 1:
        _err("viso_harris_response", rc)
    return r


def detect_harris_binned(img, n_features=1200, nbinx=24, nbiny=5, k=HARRIS_K):
    """HarrisBinnedFeatureDetector::detectImpl, reference src/viso.cpp:926-975."""
    L = load()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    kp = np.empty((max(1, n_features), 2), np.float32)
    resp = np.empty(max(1, n_features), np.float32)
    n = C.c_int(0)
    rc = L.viso_detect_harris_binned(ptr(img, C.c_uint8), img.shape[0], img.shape[1], n_features, nbinx, nbiny, k,
                                     ptr(kp, C.c_float), ptr(resp, C.c_float), C.byref(n))
    if rc != 1:
        _err("viso_detect_harris_binned", rc)
    return kp[:n.value].copy(), resp[:n.value].copy()


# ------------------------------------------------------------ batched family
class Context:
    def __init__(self, device=0, stream=None):
        self.L = load()
        self.h = self.L.viso_ctx_create(device, stream)
        if not self.h:
            raise VisoError("viso_ctx_create: " + self.L.viso_last_error().decode())
        self._batches = weakref.WeakSet()   # closed with the context: a batch destroyed after its context uses freed streams
        _live.add(self)

    def synchronize(self):
        r = self.L.viso_ctx_synchronize(self.h)
        if r != 1:
            _err("viso_ctx_synchronize", r)

    def close(self):
        if self.h:
            # its batches first, whatever order the caller (or the garbage collector, after an exception skipped the
            # caller's close() calls) takes: viso_batch_destroy on a destroyed context aborts inside the HIP runtime
            for b in list(self._batches):
                try:
                    b.close()
                except Exception:
                    pass
            h, self.h = self.h, None
            r = self.L.viso_ctx_destroy(h)
            if r != 1:
                _err("viso_ctx_destroy", r)

    def __del__(self, _finalizing=sys.is_finalizing):   # bound at definition: module globals are None late in shutdown
        if _finalizing():   # the atexit hook has closed everything that was still open
            return
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """viso_batch: n_frames stereo frames resident in HBM (include/viso_hip.h)."""

    def __init__(self, ctx, n_frames, cap, dlen=DESC_LEN):
        self.ctx, self.L = ctx, ctx.L
        self.nf, self.cap, self.dlen = n_frames, cap, dlen
        self.h = self.L.viso_batch_create(ctx.h, n_frames, cap, dlen)
        if not self.h:
            raise VisoError("viso_batch_create: " + self.L.viso_last_error().decode())
        ctx._batches.add(self)
        _live.add(self)

    def _chk(self, where, r):
        if r != 1:
            _err(where, r)

    def upload(self, kp, desc, n, f0=0):
        kp, desc, n = _f32(kp), _f32(desc), _i32(n)
        nf = kp.shape[0]
        assert kp.shape == (nf, 2, self.cap, 2) and desc.shape == (nf, 2, self.cap, self.dlen)
        self._chk("viso_batch_upload", self.L.viso_batch_upload(self.h, f0, nf, ptr(kp, C.c_float),
                                                                 ptr(desc, C.c_float), ptr(n, C.c_int32)))

    def upload_async(self, kp, desc, n, f0=0):
        """Enqueue the copies on the context's stream; kp/desc should be PinnedArray views and must stay untouched
        until the stream has passed them."""
        assert kp.dtype == np.float32 and desc.dtype == np.float32 and kp.flags.c_contiguous and desc.flags.c_contiguous
        n = _i32(n)
        nf = kp.shape[0]
        self._chk("viso_batch_upload_async", self.L.viso_batch_upload_async(self.h, f0, nf, ptr(kp, C.c_float),
                                                                             ptr(desc, C.c_float), ptr(n, C.c_int32)))

    def upload_i16(self, kp, desc16, n, f0=0, asynchronous=False):
        """Descriptors as int16 [nf][2][cap][dlen] (the lossless encoding of the Sobel windows; half the bytes).
        asynchronous=True: enqueued on the context's stream; kp / desc16 should be PinnedArray views."""
        kp, n = _f32(kp), _i32(n)
        assert desc16.dtype == np.int16 and desc16.flags.c_contiguous
        nf = kp.shape[0]
        assert kp.shape == (nf, 2, self.cap, 2) and desc16.shape == (nf, 2, self.cap, self.dlen)
        fn = self.L.viso_batch_upload_i16_async if asynchronous else self.L.viso_batch_upload_i16
        self._chk("viso_batch_upload_i16", fn(self.h, f0, nf, ptr(kp, C.c_float), ptr(desc16, C.c_int16), ptr(n, C.c_int32)))

    def upload_images_async(self, images, kp, n, f0=0):
        assert images.dtype == np.uint8 and images.flags.c_contiguous and kp.dtype == np.float32 and kp.flags.c_contiguous
        n = _i32(n)
        nf, _, rows, cols = images.shape
        self._chk("viso_batch_upload_images_async", self.L.viso_batch_upload_images_async(
            self.h, f0, nf, ptr(images, C.c_uint8), rows, cols, ptr(kp, C.c_float), ptr(n, C.c_int32)))

    def upload_images(self, images, kp, n, f0=0):
        """Image-in mode: uint8 images [nf][2][rows][cols] + keypoints (descriptors are extracted on the device)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        kp, n = _f32(kp), _i32(n)
        nf, _, rows, cols = images.shape
        assert kp.shape == (nf, 2, self.cap, 2)
        self._chk("viso_batch_upload_images", self.L.viso_batch_upload_images(
            self.h, f0, nf, ptr(images, C.c_uint8), rows, cols, ptr(kp, C.c_float), ptr(n, C.c_int32)))

    def upload_images_only(self, images, f0=0):
        """Images without keypoints: follow with detect()."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        nf, _, rows, cols = images.shape
        self._chk("viso_batch_upload_images", self.L.viso_batch_upload_images(
            self.h, f0, nf, ptr(images, C.c_uint8), rows, cols, None, None))

    def detect(self, n_features=1200, nbinx=24, nbiny=5, k=HARRIS_K):
        self._chk("viso_batch_detect", self.L.viso_batch_detect(self.h, n_features, nbinx, nbiny, k))

    def keypoints(self, t, side):
        kp = np.empty((self.cap, 2), np.float32)
        n = C.c_int(0)
        self._chk("viso_batch_get_keypoints", self.L.viso_batch_get_keypoints(self.h, t, side, ptr(kp, C.c_float), C.byref(n)))
        return kp[:n.value].copy()

    def run_images(self, matcher_only=False):
        self._chk("viso_batch_run_images", self.L.viso_batch_run_images(self.h, int(matcher_only)))

    def set_params(self, stereo, temporal, param, seed=0, first_frame=0):
        self._chk("viso_batch_set_params", self.L.viso_batch_set_params(
            self.h, C.byref(stereo), C.byref(temporal), C.byref(param), seed, first_frame))
        self.ransac_iter = int(param.ransac_iter)     # what viso_batch_get_hypotheses copies per frame

    def run_matcher(self):
        self._chk("viso_batch_run_matcher", self.L.viso_batch_run_matcher(self.h))

    def run(self):
        self._chk("viso_batch_run", self.L.viso_batch_run(self.h))

    def matches(self, which, t):
        out = np.empty((self.cap, 3), np.int32)
        n = C.c_int(0)
        self._chk("viso_batch_get_matches", self.L.viso_batch_get_matches(self.h, which, t, ptr(out, C.c_int32), C.byref(n)))
        return out[:n.value].copy()

    def circle(self, t):
        circ = np.empty((self.cap, 4), np.int32)
        pcl = np.empty((self.cap, 2), np.int32)
        n = C.c_int(0)
        self._chk("viso_batch_get_circle", self.L.viso_batch_get_circle(self.h, t, ptr(circ, C.c_int32), ptr(pcl, C.c_int32), C.byref(n)))
        return circ[:n.value].copy(), pcl[:n.value].copy()

    def pose(self, t):
        tr = np.zeros(6)
        ok, n = C.c_int(0), C.c_int(0)
        inl = np.empty(self.cap, np.int32)
        self._chk("viso_batch_get_pose", self.L.viso_batch_get_pose(self.h, t, ptr(tr, C.c_double), C.byref(ok), ptr(inl, C.c_int32), C.byref(n)))
        return ok.value, tr, inl[:n.value].copy()

    def poses(self):
        tr = np.zeros((self.nf, 6))
        ok = np.zeros(self.nf, np.int32)
        n = np.zeros(self.nf, np.int32)
        self._chk("viso_batch_get_poses", self.L.viso_batch_get_poses(self.h, ptr(tr, C.c_double), ptr(ok, C.c_int32), ptr(n, C.c_int32)))
        return tr, ok, n

    def hypotheses(self, iters=None):
        """(tr_h [nf][iters][6], ok_h [nf][iters], cnt_h [nf][iters], n_undecided) of the last run's RANSAC stage.
        The buffers are sized from the ransac_iter given to set_params (the library copies nf * that many entries);
        an `iters` that disagrees is refused instead of letting the copy run past a smaller buffer."""
        have = getattr(self, "ransac_iter", None)
        if have is None:
            raise RuntimeError("hypotheses(): set_params has not been called on this batch")
        if iters is not None and iters != have:
            raise ValueError(f"hypotheses(iters={iters}): the batch was set up with ransac_iter={have}")
        iters = have
        tr = np.zeros((self.nf, iters, 6))
        ok = np.zeros((self.nf, iters), np.int32)
        cnt = np.zeros((self.nf, iters), np.int32)
        nu = np.zeros(1, np.int32)
        self._chk("viso_batch_get_hypotheses", self.L.viso_batch_get_hypotheses(
            self.h, ptr(tr, C.c_double), ptr(ok, C.c_int32), ptr(cnt, C.c_int32), ptr(nu, C.c_int32)))
        return tr, ok, cnt, int(nu[0])

    def counters(self):
        sc = np.zeros((3, self.nf), np.int64)
        mo = np.zeros((3, self.nf), np.int64)
        self._chk("viso_batch_get_counters", self.L.viso_batch_get_counters(self.h, ptr(sc, C.c_int64), ptr(mo, C.c_int64)))
        return sc, mo

    def general_path_flags(self):
        f = np.zeros((self.nf, 2), np.int32)
        self._chk("viso_batch_get_general_path_flags", self.L.viso_batch_get_general_path_flags(self.h, ptr(f, C.c_int32)))
        return f

    def overflow_count(self):
        n = C.c_int32(0)
        self._chk("viso_batch_get_overflow_count", self.L.viso_batch_get_overflow_count(self.h, C.byref(n)))
        return n.value

    def row8_shift(self):
        s = C.c_int(-1)
        self._chk("viso_batch_get_row8_shift", self.L.viso_batch_get_row8_shift(self.h, C.byref(s)))
        return s.value

    def kernel_timing(self, enable):
        self._chk("viso_batch_kernel_timing", self.L.viso_batch_kernel_timing(self.h, int(enable)))

    def kernel_ms(self):
        ms = C.c_double(0)
        n = C.c_int(0)
        self._chk("viso_batch_kernel_ms", self.L.viso_batch_kernel_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk("viso_batch_device_ptrs", self.L.viso_batch_device_ptrs(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def close(self):
        if self.h:
            h, self.h = self.h, None
            r = self.L.viso_batch_destroy(h)
            if r != 1:
                _err("viso_batch_destroy", r)

    def __del__(self, _finalizing=sys.is_finalizing):   # bound at definition: module globals are None late in shutdown
        if _finalizing():   # the atexit hook has closed everything that was still open
            return
        try:
            self.close()
        except Exception:
            pass
