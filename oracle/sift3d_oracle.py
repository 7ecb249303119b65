"""ctypes wrapper of oracle/libsift3d_oracle.so -- TEST INFRASTRUCTURE ONLY.

The library is this repository's CPU restatement of the reference's detect+describe
path (oracle/sift3d_oracle.c).  Volumes are numpy float32 arrays of shape [nz, ny, nx]
(x fastest, as sift3d_image_data(), imutil.c:520-533).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsift3d_oracle.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class Candidate(C.Structure):
    _fields_ = [("o", C.c_int), ("s", C.c_int), ("x", C.c_int), ("y", C.c_int),
                ("z", C.c_int), ("strength", C.c_float), ("sd", C.c_double)]


class Keypoint(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("xd", C.c_double), ("yd", C.c_double),
                ("zd", C.c_double), ("sd", C.c_double), ("o", C.c_int), ("s", C.c_int),
                ("strength", C.c_float)]


class Descriptor(C.Structure):
    _fields_ = [("hist", C.c_float * 768), ("xd", C.c_double), ("yd", C.c_double),
                ("zd", C.c_double), ("sd", C.c_double)]


CAND_DTYPE = np.dtype([("o", "i4"), ("s", "i4"), ("x", "i4"), ("y", "i4"), ("z", "i4"),
                       ("strength", "f4"), ("sd", "f8")], align=True)
KP_DTYPE = np.dtype([("R", "f4", (3, 3)), ("xd", "f8"), ("yd", "f8"), ("zd", "f8"),
                     ("sd", "f8"), ("o", "i4"), ("s", "i4"), ("strength", "f4")],
                    align=True)
DESC_DTYPE = np.dtype([("hist", "f4", (768,)), ("xd", "f8"), ("yd", "f8"), ("zd", "f8"),
                       ("sd", "f8")], align=True)
assert CAND_DTYPE.itemsize == C.sizeof(Candidate)
assert KP_DTYPE.itemsize == C.sizeof(Keypoint)
assert DESC_DTYPE.itemsize == C.sizeof(Descriptor)

_lib = None


def build():
    """Compile the restatement (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_gauss_taps.argtypes = [C.c_double, _f32p, C.c_int]
        L.orc_fir_axis.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p,
                                   C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int]
        L.orc_blur.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_double,
                               C.c_double, C.c_double, _f32p, C.c_int, C.c_double, C.c_int]
        L.orc_downsample.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_eigen3.argtypes = [_f64p, _f64p, _f64p]
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        for n in ("orc_set_peak_thresh", "orc_set_corner_thresh", "orc_set_sigma_n",
                  "orc_set_sigma0"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_double]
        L.orc_set_num_kp_levels.argtypes = [C.c_void_p, C.c_uint]
        L.orc_set_fir_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_cuboid_extrema.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_volume.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_double]
        L.orc_detect.argtypes = L.orc_set_volume.argtypes
        for n in ("orc_build_pyramids", "orc_find_extrema", "orc_assign_orientations",
                  "orc_describe", "orc_num_octaves", "orc_num_candidates",
                  "orc_num_keypoints", "orc_num_descriptors"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_sort_by_strength.argtypes = [C.c_void_p, C.c_int]
        L.orc_candidates.restype = C.POINTER(Candidate)
        L.orc_candidates.argtypes = [C.c_void_p]
        L.orc_keypoints.restype = C.POINTER(Keypoint)
        L.orc_keypoints.argtypes = [C.c_void_p]
        L.orc_descriptors.restype = C.POINTER(Descriptor)
        L.orc_descriptors.argtypes = [C.c_void_p]
        L.orc_set_keypoints.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_level.restype = C.POINTER(C.c_float)
        L.orc_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _i32p, _f64p,
                                C.POINTER(C.c_double)]
        L.orc_dogmax.restype = C.c_float
        L.orc_dogmax.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_filter.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), _f32p]
        L.orc_mesh.argtypes = [C.c_void_p, _f32p, _i32p]
        L.orc_timings.restype = C.POINTER(C.c_double)
        L.orc_timings.argtypes = [C.c_void_p]
        L.sift3d_amd_synth_survey.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_uint64]
        L.sift3d_amd_synth_lattice.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_uint64]
        _lib = L
    return _lib


# ---- unit-level ---------------------------------------------------------------

def gauss_taps(sigma):
    taps = np.zeros(1024, np.float32)
    w = lib().orc_gauss_taps(float(sigma), taps, 1024)
    return taps[:w].copy()


def fir_axis(vol, taps, axis, uf=1.0, n_glob=None, off=0, out_lo=0, out_hi=None, mode=0):
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    n_loc = (nx, ny, nz)[axis]
    out = np.zeros_like(vol)
    taps = np.ascontiguousarray(taps, np.float32)
    r = lib().orc_fir_axis(vol, out, nx, ny, nz, axis, taps, len(taps), float(uf),
                           n_loc if n_glob is None else n_glob, off, out_lo,
                           n_loc if out_hi is None else out_hi, mode)
    return out, r


def blur(vol, taps, units=(1, 1, 1), unit=1.0, mode=0):
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    out = np.empty_like(vol)
    taps = np.ascontiguousarray(taps, np.float32)
    r = lib().orc_blur(vol, out, nx, ny, nz, *map(float, units), taps, len(taps),
                       float(unit), mode)
    assert r == 0
    return out


def downsample(vol):
    vol = np.ascontiguousarray(vol, np.float32)
    nz, ny, nx = vol.shape
    out = np.empty((nz // 2, ny // 2, nx // 2), np.float32)
    lib().orc_downsample(vol, nx, ny, nz, out)
    return out


def eigen3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    Q = np.zeros(9)
    L = np.zeros(3)
    lib().orc_eigen3(A, Q, L)
    return Q.reshape(3, 3), L


def synth_survey(n, nblob=None, seed=0):
    """SURVEY.md 8(d) volume; n is an int or (nx, ny, nz)."""
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    if nblob is None:
        nblob = int(round(200 * (nx * ny * nz) / 64.0 ** 3))
    v = np.zeros((nz, ny, nx), np.float32)
    lib().sift3d_amd_synth_survey(v, nx, ny, nz, nblob, seed)
    return v


def synth_lattice(n, seed=1):
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    v = np.zeros((nz, ny, nx), np.float32)
    lib().sift3d_amd_synth_lattice(v, nx, ny, nz, seed)
    return v


# ---- pipeline -------------------------------------------------------------------

class Oracle:
    def __init__(self, peak_thresh=None, corner_thresh=None, num_kp_levels=None,
                 sigma_n=None, sigma0=None, fir_mode=1, cuboid_extrema=False):
        self.L = lib()
        self.h = self.L.orc_create()
        self.L.orc_set_fir_mode(self.h, fir_mode)
        self.L.orc_set_cuboid_extrema(self.h, int(bool(cuboid_extrema)))
        if sigma_n is not None:
            assert self.L.orc_set_sigma_n(self.h, sigma_n) == 0
        if sigma0 is not None:
            assert self.L.orc_set_sigma0(self.h, sigma0) == 0
        if peak_thresh is not None:
            assert self.L.orc_set_peak_thresh(self.h, peak_thresh) == 0
        if corner_thresh is not None:
            assert self.L.orc_set_corner_thresh(self.h, corner_thresh) == 0
        if num_kp_levels is not None:
            assert self.L.orc_set_num_kp_levels(self.h, num_kp_levels) == 0

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _vol(self, vol):
        vol = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = vol.shape
        return vol, nx, ny, nz

    def set_volume(self, vol, units=(1, 1, 1)):
        vol, nx, ny, nz = self._vol(vol)
        return self.L.orc_set_volume(self.h, vol, nx, ny, nz, *map(float, units))

    def build_pyramids(self):
        return self.L.orc_build_pyramids(self.h)

    def find_extrema(self):
        return self.L.orc_find_extrema(self.h)

    def assign_orientations(self):
        return self.L.orc_assign_orientations(self.h)

    def detect(self, vol, units=(1, 1, 1)):
        vol, nx, ny, nz = self._vol(vol)
        return self.L.orc_detect(self.h, vol, nx, ny, nz, *map(float, units))

    def describe(self):
        return self.L.orc_describe(self.h)

    def sort_by_strength(self, limit):
        self.L.orc_sort_by_strength(self.h, int(limit))

    @property
    def num_octaves(self):
        return self.L.orc_num_octaves(self.h)

    def candidates(self):
        n = self.L.orc_num_candidates(self.h)
        if n == 0:
            return np.zeros(0, CAND_DTYPE)
        p = self.L.orc_candidates(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)),
                                     shape=(n * CAND_DTYPE.itemsize,)).view(CAND_DTYPE).copy()

    def keypoints(self):
        n = self.L.orc_num_keypoints(self.h)
        if n == 0:
            return np.zeros(0, KP_DTYPE)
        p = self.L.orc_keypoints(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)),
                                     shape=(n * KP_DTYPE.itemsize,)).view(KP_DTYPE).copy()

    def set_keypoints(self, kps):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        return self.L.orc_set_keypoints(self.h, kps.ctypes.data, len(kps))

    def descriptors(self):
        n = self.L.orc_num_descriptors(self.h)
        if n == 0:
            return np.zeros(0, DESC_DTYPE)
        p = self.L.orc_descriptors(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)),
                                     shape=(n * DESC_DTYPE.itemsize,)).view(DESC_DTYPE).copy()

    def level(self, which, o, s):
        dims = np.zeros(3, np.int32)
        units = np.zeros(3, np.float64)
        sc = C.c_double()
        p = self.L.orc_level(self.h, which, o, s, dims, units, C.byref(sc))
        nx, ny, nz = map(int, dims)
        return np.ctypeslib.as_array(p, shape=(nz, ny, nx)).copy(), units.copy(), sc.value

    def dogmax(self, o, s):
        return self.L.orc_dogmax(self.h, o, s)

    def filters(self):
        out = []
        taps = np.zeros(1024, np.float32)
        sg = C.c_double()
        idx = -1
        while True:
            w = self.L.orc_filter(self.h, idx, C.byref(sg), taps)
            if w < 0:
                break
            out.append((sg.value, taps[:w].copy()))
            idx += 1
        return out

    def mesh(self):
        v = np.zeros((20, 3, 3), np.float32)
        idx = np.zeros((20, 3), np.int32)
        self.L.orc_mesh(self.h, v, idx)
        return v, idx

    def timings(self):
        p = self.L.orc_timings(self.h)
        return dict(zip(("set_volume", "gauss", "dog", "extrema", "orient", "describe"),
                        [p[i] for i in range(6)]))

    def kp_mat(self):
        """sift3d_keypoint_store_to_mat_rm layout (sift.c:1644-1671): N x 3 double."""
        k = self.keypoints()
        f = np.ldexp(1.0, k["o"])
        return np.stack([f * k["xd"], f * k["yd"], f * k["zd"]], axis=1)

    def desc_mat(self):
        """sift3d_descriptor_store_to_mat_rm layout (sift.c:1683-1726): N x 771 float."""
        d = self.descriptors()
        out = np.zeros((len(d), 771), np.float32)
        out[:, 0] = d["xd"]
        out[:, 1] = d["yd"]
        out[:, 2] = d["zd"]
        out[:, 3:] = d["hist"]
        return out
