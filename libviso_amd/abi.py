"""ctypes view of include/viso_hip.h (POD structs + prototypes).

Test/bench plumbing only: the product is the C-ABI library itself
(libviso_amd/csrc -> libviso_hip.so) and the C++ host mirror in
libviso_amd/host.  The oracle's ctypes wrapper (oracle/pyoracle.py) reuses the
struct definitions from here; nothing here imports the oracle.
"""
import ctypes as C

import numpy as np

VISO_OK = 1
VISO_ERR_ARG = -1
VISO_ERR_HIP = -2
VISO_ERR_UNSUPPORTED = -3
DESC_LEN = 121


class MatchParams(C.Structure):
    """struct viso_match_params <- MatchParams, reference src/viso.cpp:48-75."""
    _fields_ = [
        ("enforce_epipolar", C.c_int32),
        ("enforce_2nd_best", C.c_int32),
        ("max_neighbors", C.c_int32),
        ("_pad", C.c_int32),
        ("F", C.c_double * 9),
        ("sampson_thresh", C.c_double),
        ("ratio_2nd_best", C.c_double),
        ("radius", C.c_double),
    ]

    @classmethod
    def stereo(cls, F):
        """MatchParams(F), reference src/viso.cpp:62-71."""
        mp = cls()
        mp.enforce_epipolar = 1
        mp.sampson_thresh = 1.0
        mp.enforce_2nd_best = 0
        mp.ratio_2nd_best = 0.8
        mp.max_neighbors = 200
        mp.radius = 80.0
        Fa = np.asarray(F, dtype=np.float64).reshape(9)
        for i in range(9):
            mp.F[i] = float(Fa[i])
        return mp

    @classmethod
    def temporal(cls):
        """MatchParams(), reference src/viso.cpp:72-74."""
        mp = cls()
        mp.enforce_epipolar = 0
        mp.enforce_2nd_best = 1
        mp.ratio_2nd_best = 0.9
        mp.max_neighbors = 250
        mp.radius = 80.0
        return mp


class Param(C.Structure):
    """struct viso_param <- struct param, reference src/viso.h:58-72."""
    _fields_ = [
        ("base", C.c_double),
        ("ransac_iter", C.c_int32),
        ("save_debug", C.c_int32),
        ("inlier_threshold", C.c_double),
        ("thresh", C.c_double),
        ("f", C.c_double),
        ("cu", C.c_double),
        ("cv", C.c_double),
    ]

    @classmethod
    def default(cls, base=0.0, f=0.0, cu=0.0, cv=0.0):
        p = cls()
        p.ransac_iter = 50
        p.inlier_threshold = 2.0
        p.save_debug = 1
        p.thresh = 1e-4
        p.base, p.f, p.cu, p.cv = base, f, cu, cv
        return p

    @classmethod
    def kitti00(cls):
        """Calibration of KITTI seq 00 (values: reference test/test.cpp:11-20)."""
        return cls.default(base=386.1448 / 718.856, f=718.856, cu=607.1928, cv=185.2157)


def ptr(a, ctype):
    """Pointer to a contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.POINTER(ctype))


f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)
intp = C.POINTER(C.c_int)


def declare_common(lib, prefix):
    """Prototypes shared by libviso_hip.so (prefix 'viso_') and the oracle
    (prefix 'oracle_'): same POD signatures, include/viso_hip.h."""
    def fn(name, restype, argtypes):
        f = getattr(lib, prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        return f

    MP = C.POINTER(MatchParams)
    PP = C.POINTER(Param)
    fn("match_circle", C.c_int, [i32p, C.c_int, i32p, C.c_int, i32p, C.c_int, i32p, C.c_int,
                                  i32p, i32p, C.c_int, intp])
    fn("collect_matches", C.c_int, [f32p, C.c_int, f32p, C.c_int, i32p, C.c_int, f64p])
    fn("triangulate_rectified", C.c_int, [f64p, C.c_int, PP, f64p])
    fn("get_inliers", C.c_int, [f64p, f64p, C.c_int, f64p, PP, i32p, intp, f64p])
    fn("ransac_minimize_reproj", C.c_int, [f64p, f64p, C.c_int, f64p, i32p, intp, PP, i32p,
                                           C.c_uint64, C.c_uint64])
    fn("ransac_samples", None, [C.c_uint64, C.c_uint64, C.c_int, C.c_int, i32p])
    fn("tr2mat", None, [f64p, f64p])
    fn("pose_update", None, [f64p, f64p, f64p])
    fn("F_from_P", None, [f64p, f64p, f64p])
    fn("extract_descriptors", C.c_int, [u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_int, f32p])
    return MP, PP
