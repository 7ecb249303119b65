"""ctypes view of oracle/_ref/libsift3d_refprobe.so -- TEST INFRASTRUCTURE ONLY.

The library is the unmodified reference (built by `make -C oracle ref` from
/root/reference) plus the dump hooks of oracle/ref_probe.c.  It exists only in the
build container; it is used by oracle/make_golden.py to produce tests/golden/*.npz and by
the `refprobe`-marked tests that cross-check the CPU restatement directly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libsift3d_refprobe.so")
# the same sources compiled with -DCUBOID_EXTREMA (sift.c:24)
LIB_PATH_CUBOID = os.path.join(HERE, "_ref", "libsift3d_refprobe_cuboid.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def available():
    return os.path.exists(LIB_PATH)


_libs = {}


def lib(cuboid=False):
    if cuboid not in _libs:
        L = C.CDLL(LIB_PATH_CUBOID if cuboid else LIB_PATH)
        L.probe_gauss_filter.argtypes = [C.c_double, _f32p, C.c_int]
        L.probe_apply_sep_fir.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_double,
                                          _f32p, C.c_int, C.c_double]
        L.probe_fir_axis.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_double,
                                     _f32p, C.c_int, C.c_double, C.c_int]
        L.probe_downsample.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p]
        L.probe_eigen3.argtypes = [_f64p, _f64p, _f64p]
        L.probe_make.restype = C.c_void_p
        L.probe_free.argtypes = [C.c_void_p]
        L.probe_detector.restype = C.c_void_p
        L.probe_detector.argtypes = [C.c_void_p]
        L.probe_detect.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int,
                                   C.c_double, C.c_double, C.c_double]
        L.probe_detect_public.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int]
        L.probe_describe.argtypes = [C.c_void_p]
        L.probe_sort.argtypes = [C.c_void_p, C.c_int]
        for n in ("probe_num_octaves", "probe_num_cand", "probe_num_kp"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.probe_get_cand.argtypes = [C.c_void_p, _i32p, _f32p, _f64p]
        L.probe_get_kp.argtypes = [C.c_void_p, _i32p, _f64p, _f32p, _f32p]
        L.probe_level.restype = C.POINTER(C.c_float)
        L.probe_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _i32p, _f64p,
                                  C.POINTER(C.c_double)]
        L.probe_gss.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), _f32p]
        L.probe_mesh.argtypes = [C.c_void_p, _f32p, _i32p]
        L.probe_get_desc.argtypes = [C.c_void_p, _f32p, _f64p]
        L.probe_kp_mat.argtypes = [C.c_void_p, _f64p]
        L.probe_desc_mat.argtypes = [C.c_void_p, _f32p]
        L.probe_save.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.probe_time_detect.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _f64p]
        for n in ("sift3d_detector_set_peak_thresh", "sift3d_detector_set_corner_thresh",
                  "sift3d_detector_set_sigma_n", "sift3d_detector_set_sigma0"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double]
        L.sift3d_detector_set_num_kp_levels.argtypes = [C.c_void_p, C.c_uint]
        _libs[cuboid] = L
    return _libs[cuboid]


def gauss_filter(sigma):
    taps = np.zeros(512, np.float32)
    w = lib().probe_gauss_filter(float(sigma), taps, 512)
    return taps[:w].copy()


def apply_sep_fir(vol, taps, units=(1, 1, 1), unit=1.0):
    """vol: ndarray [nz, ny, nx] float32 (x fastest)."""
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    out = np.empty_like(vol)
    taps = np.ascontiguousarray(taps, np.float32)
    r = lib().probe_apply_sep_fir(vol, out, nx, ny, nz, *map(float, units), taps,
                                  len(taps), float(unit))
    assert r == 0
    return out


def fir_axis(vol, taps, axis, units=(1, 1, 1), unit=1.0):
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    out = np.empty_like(vol)
    taps = np.ascontiguousarray(taps, np.float32)
    r = lib().probe_fir_axis(vol, out, nx, ny, nz, *map(float, units), taps, len(taps),
                             float(unit), int(axis))
    assert r == 0
    return out


def downsample(vol):
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    out = np.empty((nz // 2, ny // 2, nx // 2), np.float32)
    assert lib().probe_downsample(vol, nx, ny, nz, out) == 0
    return out


def eigen3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    Q = np.zeros(9)
    L = np.zeros(3)
    assert lib().probe_eigen3(A, Q, L) == 0
    return Q.reshape(3, 3), L


class Probe:
    """One reference detector + stores, with stage dumps."""

    def __init__(self, peak_thresh=None, corner_thresh=None, num_kp_levels=None,
                 sigma_n=None, sigma0=None, cuboid_extrema=False):
        self.L = lib(bool(cuboid_extrema))
        self.h = self.L.probe_make()
        det = self.L.probe_detector(self.h)
        if sigma_n is not None:
            assert self.L.sift3d_detector_set_sigma_n(det, sigma_n) == 0
        if sigma0 is not None:
            assert self.L.sift3d_detector_set_sigma0(det, sigma0) == 0
        if peak_thresh is not None:
            assert self.L.sift3d_detector_set_peak_thresh(det, peak_thresh) == 0
        if corner_thresh is not None:
            assert self.L.sift3d_detector_set_corner_thresh(det, corner_thresh) == 0
        if num_kp_levels is not None:
            assert self.L.sift3d_detector_set_num_kp_levels(det, num_kp_levels) == 0

    def close(self):
        if self.h:
            self.L.probe_free(self.h)
            self.h = None

    def detect(self, vol, units=(1, 1, 1)):
        vol = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = vol.shape
        return self.L.probe_detect(self.h, vol, nx, ny, nz, *map(float, units))

    def detect_public(self, vol):
        vol = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = vol.shape
        return self.L.probe_detect_public(self.h, vol, nx, ny, nz)

    def time_detect(self, vol):
        vol = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = vol.shape
        t = np.zeros(5)
        assert self.L.probe_time_detect(self.h, vol, nx, ny, nz, t) == 0
        return t

    def describe(self):
        return self.L.probe_describe(self.h)

    def sort(self, limit):
        self.L.probe_sort(self.h, int(limit))

    @property
    def num_octaves(self):
        return self.L.probe_num_octaves(self.h)

    def candidates(self):
        n = self.L.probe_num_cand(self.h)
        osxyz = np.zeros((n, 5), np.int32)
        st = np.zeros(n, np.float32)
        sd = np.zeros(n, np.float64)
        if n:
            self.L.probe_get_cand(self.h, osxyz, st, sd)
        return dict(osxyz=osxyz, strength=st, sd=sd)

    def keypoints(self):
        n = self.L.probe_num_kp(self.h)
        os_ = np.zeros((n, 2), np.int32)
        xyzsd = np.zeros((n, 4), np.float64)
        st = np.zeros(n, np.float32)
        R = np.zeros((n, 9), np.float32)
        if n:
            self.L.probe_get_kp(self.h, os_, xyzsd, st, R)
        return dict(os=os_, xyzsd=xyzsd, strength=st, R=R.reshape(n, 3, 3))

    def level(self, which, o, s):
        dims = np.zeros(3, np.int32)
        units = np.zeros(3, np.float64)
        sc = C.c_double()
        p = self.L.probe_level(self.h, which, o, s, dims, units, C.byref(sc))
        nx, ny, nz = map(int, dims)
        a = np.ctypeslib.as_array(p, shape=(nz, ny, nx)).copy()
        return a, units.copy(), sc.value

    def gss(self):
        out = []
        taps = np.zeros(512, np.float32)
        sg = C.c_double()
        idx = -1
        while True:
            w = self.L.probe_gss(self.h, idx, C.byref(sg), taps)
            if w < 0:
                break
            out.append((sg.value, taps[:w].copy()))
            idx += 1
        return out

    def mesh(self):
        v = np.zeros((20, 3, 3), np.float32)
        idx = np.zeros((20, 3), np.int32)
        self.L.probe_mesh(self.h, v, idx)
        return v, idx

    def descriptors(self):
        n = self.L.probe_num_kp(self.h)
        hist = np.zeros((n, 768), np.float32)
        xyzsd = np.zeros((n, 4), np.float64)
        m = self.L.probe_get_desc(self.h, hist, xyzsd)
        return hist[:m], xyzsd[:m]

    def kp_mat(self):
        n = self.L.probe_num_kp(self.h)
        out = np.zeros((n, 3), np.float64)
        assert self.L.probe_kp_mat(self.h, out) == n
        return out

    def desc_mat(self):
        n = self.L.probe_num_kp(self.h)
        out = np.zeros((n, 771), np.float32)
        assert self.L.probe_desc_mat(self.h, out) == n
        return out

    def save(self, kp_path=None, desc_path=None):
        return self.L.probe_save(self.h, kp_path.encode() if kp_path else None,
                                 desc_path.encode() if desc_path else None)
