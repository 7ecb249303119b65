"""Python mirror of the reference's C API, bound with ctypes to libsift3d_amd.so.

Same names and argument meaning as sift.h / imutil.h of fatimp/SIFT3D v2.0
(reference: sift3d/sift.h:24-208, sift3d/imutil.h:39-110): functions return
SIFT3D_SUCCESS (0) / SIFT3D_FAILURE (-1) and print a message on stderr, objects are
opaque handles the caller frees.  The thin classes below only add lifetime management
and numpy views; they contain no algorithmic code.

Volumes are numpy float32 arrays of shape [nz, ny, nx] (x fastest), which is the memory
layout of sift3d_image_data() (reference: sift3d/imutil.c:520-533).
"""
import ctypes as C
import os

import numpy as np

from . import _native

SIFT3D_SUCCESS = 0
SIFT3D_FAILURE = -1
SIFT3D_DOUBLE, SIFT3D_FLOAT, SIFT3D_INT = 0, 1, 2
TIMED_BLURS = 8
NUM_TIMINGS = 10 + 2 * TIMED_BLURS + 4

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")

_bound = None


def lib():
    """The loaded library with argtypes/restypes declared for every exported symbol."""
    global _bound
    if _bound is not None:
        return _bound
    L = _native.load()
    vp = C.c_void_p
    sig = {
        # imutil.h
        "sift3d_make_image": (vp, [C.c_int] * 4),
        "sift3d_free_image": (None, [vp]),
        "sift3d_read_image": (vp, [C.c_char_p]),
        "sift3d_image_data": (C.POINTER(C.c_float), [vp]),
        "sift3d_make_mat_rm": (vp, []),
        "sift3d_free_mat_rm": (None, [vp]),
        "sift3d_mat_rm_data": (vp, [vp]),
        "sift3d_mat_rm_dimensions": (None, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "sift3d_mat_rm_type": (C.c_int, [vp]),
        # sift.h
        "sift3d_make_detector": (vp, []),
        "sift3d_free_detector": (None, [vp]),
        "sift3d_detector_set_peak_thresh": (C.c_int, [vp, C.c_double]),
        "sift3d_detector_set_corner_thresh": (C.c_int, [vp, C.c_double]),
        "sift3d_detector_set_num_kp_levels": (C.c_int, [vp, C.c_uint]),
        "sift3d_detector_set_sigma_n": (C.c_int, [vp, C.c_double]),
        "sift3d_detector_set_sigma0": (C.c_int, [vp, C.c_double]),
        "sift3d_detect_keypoints": (C.c_int, [vp, vp, vp]),
        "sift3d_extract_descriptors": (C.c_int, [vp, vp, vp]),
        "sift3d_make_keypoint_store": (vp, []),
        "sift3d_free_keypoint_store": (None, [vp]),
        "sift3d_keypoint_store_to_mat_rm": (C.c_int, [vp, vp]),
        "sift3d_keypoint_store_save": (C.c_int, [C.c_char_p, vp]),
        "sift3d_keypoint_store_sort_by_strength": (None, [vp, C.c_int]),
        "sift3d_make_descriptor_store": (vp, []),
        "sift3d_free_descriptor_store": (None, [vp]),
        "sift3d_descriptor_store_save": (C.c_int, [C.c_char_p, vp]),
        "sift3d_descriptor_store_to_mat_rm": (C.c_int, [vp, vp]),
        # sift3d_amd.h extensions
        "sift3d_amd_detect_keypoints_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int,
                                                         C.c_double, C.c_double, C.c_double, vp]),
        "sift3d_amd_image_set_units": (C.c_int, [vp, C.c_double, C.c_double, C.c_double]),
        "sift3d_amd_timings": (C.POINTER(C.c_double), [vp]),
        "sift3d_amd_num_candidates": (C.c_int, [vp]),
        "sift3d_amd_describe_clock": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "sift3d_amd_build_pyramid_device": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                      C.c_double]),
        "sift3d_amd_image_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
        "sift3d_amd_detector_set_cuboid_extrema": (C.c_int, [vp, C.c_int]),
        "sift3d_amd_detector_set_dogmax_pass": (C.c_int, [vp, C.c_int]),
        "sift3d_amd_detector_set_exact_descriptors": (C.c_int, [vp, C.c_int]),
        "sift3d_amd_detector_set_serial_orientation": (C.c_int, [vp, C.c_int]),
        "sift3d_amd_detector_dogmax": (C.c_int, [vp, vp, C.c_int]),
        "sift3d_amd_copy_level": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, _i32p]),
        "sift3d_amd_keypoint_store_size": (C.c_int, [vp]),
        "sift3d_amd_keypoint_store_get": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int),
                                                    C.POINTER(C.c_int), _f64p,
                                                    C.POINTER(C.c_float), _f32p]),
        "sift3d_amd_keypoint_store_set": (C.c_int, [vp, C.c_int, _i32p, _f64p, _f32p, _f32p]),
        "sift3d_amd_descriptor_store_size": (C.c_int, [vp]),
        "sift3d_amd_descriptor_store_set": (C.c_int, [vp, C.c_int, _f64p, _f32p, C.c_int, C.c_int, C.c_int]),
        "sift3d_amd_nn_match": (C.c_int, [vp, vp, C.c_double, _i32p]),
        "sift3d_amd_descriptor_store_keep_device": (C.c_int, [vp, C.c_int]),
        "sift3d_amd_descriptor_store_xyz_all": (C.c_int, [vp, _f64p]),
        "sift3d_amd_descriptor_store_xyz": (C.c_int, [vp, C.c_int, _f64p]),
        "sift3d_amd_ransac_affine": (C.c_int, [_f64p, _f64p, C.c_int, C.c_double, C.c_int, C.c_uint64,
                                              _f64p, np.ctypeslib.ndpointer(np.uint8),
                                              C.POINTER(C.c_int)]),
        "sift3d_amd_device_available": (C.c_int, []),
        "sift3d_amd_version": (C.c_char_p, []),
        "sift3d_amd_synth_survey": (None, [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]),
        "sift3d_amd_synth_lattice": (None, [_f32p, C.c_int, C.c_int, C.c_int, C.c_uint64]),
        "sift3d_amd_gauss_filter": (C.c_int, [C.c_double, _f32p, C.c_int]),
        "sift3d_amd_init": (C.c_int, []),
        "sift3d_amd_host_expf": (None, [_f32p, _f32p, C.c_size_t]),
        "sift3d_amd_host_eigen3": (None, [_f64p, _f64p, _f64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export the ABI
        fn.restype = res
        fn.argtypes = args
    _bound = L
    return L


def device_available():
    return bool(lib().sift3d_amd_device_available())


class MatRm:
    """sift3d_mat_rm (reference: sift3d/imutil.h:77-110)."""

    def __init__(self):
        self.h = lib().sift3d_make_mat_rm()
        if not self.h:
            raise MemoryError("sift3d_make_mat_rm")

    def free(self):
        # (at interpreter shutdown module globals may already be gone: nothing to do then)
        if getattr(self, "h", None) and lib is not None:
            lib().sift3d_free_mat_rm(self.h)
            self.h = None

    __del__ = free

    def dimensions(self):
        c, r = C.c_int(), C.c_int()
        lib().sift3d_mat_rm_dimensions(self.h, C.byref(c), C.byref(r))
        return c.value, r.value

    def type(self):
        return lib().sift3d_mat_rm_type(self.h)

    def numpy(self):
        cols, rows = self.dimensions()
        dt = {SIFT3D_DOUBLE: np.float64, SIFT3D_FLOAT: np.float32, SIFT3D_INT: np.int32}[self.type()]
        if rows * cols == 0:
            return np.zeros((rows, cols), dt)
        p = lib().sift3d_mat_rm_data(self.h)
        buf = (C.c_char * (rows * cols * np.dtype(dt).itemsize)).from_address(p)
        return np.frombuffer(buf, dt).reshape(rows, cols).copy()


class Image:
    """sift3d_image (reference: sift3d/imutil.h:39-65)."""

    def __init__(self, nx, ny, nz, nc=1):
        self.h = lib().sift3d_make_image(nx, ny, nz, nc)
        if not self.h:
            raise ValueError("sift3d_make_image(%d, %d, %d, %d) failed" % (nx, ny, nz, nc))
        self.shape = (nz, ny, nx) if nc == 1 else (nz, ny, nx, nc)

    @classmethod
    def read(cls, path):
        """sift3d_read_image: single-file NIFTI-1 (.nii, .nii.gz).  Raises IOError on failure."""
        h = lib().sift3d_read_image(os.fsencode(path))
        if not h:
            raise IOError("sift3d_read_image(%r) failed" % (path,))
        self = cls.__new__(cls)
        self.h = h
        dims = (C.c_int * 4)()
        lib().sift3d_amd_image_info(h, dims, None)
        nx, ny, nz, nc = dims
        self.shape = (nz, ny, nx) if nc == 1 else (nz, ny, nx, nc)
        return self

    @property
    def units(self):
        u = (C.c_double * 3)()
        lib().sift3d_amd_image_info(self.h, None, u)
        return tuple(u)

    @classmethod
    def from_array(cls, vol, units=None):
        vol = np.ascontiguousarray(vol, np.float32)
        nz, ny, nx = vol.shape
        im = cls(nx, ny, nz, 1)
        im.data()[...] = vol
        if units is not None:
            if lib().sift3d_amd_image_set_units(im.h, *map(float, units)) != 0:
                raise ValueError("invalid units")
        return im

    def data(self):
        p = lib().sift3d_image_data(self.h)
        return np.ctypeslib.as_array(p, shape=self.shape)

    def free(self):
        # (at interpreter shutdown module globals may already be gone: nothing to do then)
        if getattr(self, "h", None) and lib is not None:
            lib().sift3d_free_image(self.h)
            self.h = None

    __del__ = free


KP_DTYPE = np.dtype([("R", "f4", (3, 3)), ("xd", "f8"), ("yd", "f8"), ("zd", "f8"),
                     ("sd", "f8"), ("o", "i4"), ("s", "i4"), ("strength", "f4")])


class KeypointStore:
    """sift3d_keypoint_store (reference: sift3d/sift.h:121-165)."""

    def __init__(self):
        self.h = lib().sift3d_make_keypoint_store()

    def free(self):
        # (at interpreter shutdown module globals may already be gone: nothing to do then)
        if getattr(self, "h", None) and lib is not None:
            lib().sift3d_free_keypoint_store(self.h)
            self.h = None

    __del__ = free

    def __len__(self):
        return lib().sift3d_amd_keypoint_store_size(self.h)

    def to_mat_rm(self):
        m = MatRm()
        if lib().sift3d_keypoint_store_to_mat_rm(self.h, m.h) != 0:
            raise RuntimeError("sift3d_keypoint_store_to_mat_rm failed")
        return m.numpy()

    def sort_by_strength(self, limit=0):
        lib().sift3d_keypoint_store_sort_by_strength(self.h, int(limit))

    def save(self, path):
        return lib().sift3d_keypoint_store_save(path.encode(), self.h)

    def records(self):
        """All fields of every keypoint (the reference only exposes them through "%f" CSV)."""
        n = len(self)
        out = np.zeros(n, KP_DTYPE)
        o, s, st = C.c_int(), C.c_int(), C.c_float()
        xyz = np.zeros(4, np.float64)
        R = np.zeros(9, np.float32)
        L = lib()
        for i in range(n):
            assert L.sift3d_amd_keypoint_store_get(self.h, i, C.byref(o), C.byref(s), xyz,
                                                   C.byref(st), R) == 0
            out[i] = (R.reshape(3, 3), xyz[0], xyz[1], xyz[2], xyz[3], o.value, s.value, st.value)
        return out

    def set_records(self, recs):
        n = len(recs)
        os_ = np.ascontiguousarray(np.stack([recs["o"], recs["s"]], 1), np.int32).reshape(-1)
        xyz = np.ascontiguousarray(np.stack([recs["xd"], recs["yd"], recs["zd"], recs["sd"]], 1),
                                   np.float64).reshape(-1)
        st = np.ascontiguousarray(recs["strength"], np.float32)
        R = np.ascontiguousarray(recs["R"], np.float32).reshape(-1)
        if n == 0:
            os_ = np.zeros(2, np.int32); xyz = np.zeros(4); st = np.zeros(1, np.float32)
            R = np.zeros(9, np.float32)
        return lib().sift3d_amd_keypoint_store_set(self.h, n, os_, xyz, st, R)


class DescriptorStore:
    """sift3d_descriptor_store (reference: sift3d/sift.h:175-208)."""

    def __init__(self):
        self.h = lib().sift3d_make_descriptor_store()

    def free(self):
        # (at interpreter shutdown module globals may already be gone: nothing to do then)
        if getattr(self, "h", None) and lib is not None:
            lib().sift3d_free_descriptor_store(self.h)
            self.h = None

    __del__ = free

    def __len__(self):
        return lib().sift3d_amd_descriptor_store_size(self.h)

    def to_mat_rm(self):
        m = MatRm()
        if lib().sift3d_descriptor_store_to_mat_rm(self.h, m.h) != 0:
            raise RuntimeError("sift3d_descriptor_store_to_mat_rm failed")
        return m.numpy()

    def save(self, path):
        return lib().sift3d_descriptor_store_save(path.encode(), self.h)

    def keep_device(self, on=True):
        """Keep a copy of the histograms in HBM (written by extract_descriptors) for the matcher."""
        return lib().sift3d_amd_descriptor_store_keep_device(self.h, int(bool(on)))

    def xyz(self):
        """Keypoint coordinates in octave-0 voxels, one row per descriptor."""
        out = np.zeros((max(len(self), 1), 3), np.float64)
        if lib().sift3d_amd_descriptor_store_xyz_all(self.h, out.reshape(-1)) != 0:
            raise RuntimeError("sift3d_amd_descriptor_store_xyz_all failed")
        return out[:len(self)]

    def set(self, xyz_sd, hist, dims=(0, 0, 0)):
        """Fill the store from host arrays (tests of the writers without a device)."""
        xyz_sd = np.ascontiguousarray(xyz_sd, np.float64).reshape(-1, 4)
        hist = np.ascontiguousarray(hist, np.float32).reshape(-1, 768)
        assert len(xyz_sd) == len(hist)
        if len(hist) == 0:
            xyz_sd, hist = np.zeros((1, 4)), np.zeros((1, 768), np.float32)
            return lib().sift3d_amd_descriptor_store_set(self.h, 0, xyz_sd.reshape(-1), hist.reshape(-1),
                                                         *[int(v) for v in dims])
        return lib().sift3d_amd_descriptor_store_set(self.h, len(hist), xyz_sd.reshape(-1),
                                                     hist.reshape(-1), *[int(v) for v in dims])


class Detector:
    """sift3d_detector (reference: sift3d/sift.h:24-111)."""

    def __init__(self, peak_thresh=None, corner_thresh=None, num_kp_levels=None, sigma_n=None,
                 sigma0=None, cuboid_extrema=None, exact_descriptors=None):
        self.h = lib().sift3d_make_detector()
        if not self.h:
            raise MemoryError("sift3d_make_detector")
        for name, v in (("sigma_n", sigma_n), ("sigma0", sigma0), ("peak_thresh", peak_thresh),
                        ("corner_thresh", corner_thresh), ("num_kp_levels", num_kp_levels),
                        ("cuboid_extrema", cuboid_extrema), ("exact_descriptors", exact_descriptors)):
            if v is not None and getattr(self, "set_" + name)(v) != 0:
                raise ValueError("sift3d_detector_set_%s(%r) failed" % (name, v))

    def free(self):
        # (at interpreter shutdown module globals may already be gone: nothing to do then)
        if getattr(self, "h", None) and lib is not None:
            lib().sift3d_free_detector(self.h)
            self.h = None

    __del__ = free

    def set_peak_thresh(self, v):
        return lib().sift3d_detector_set_peak_thresh(self.h, float(v))

    def set_corner_thresh(self, v):
        return lib().sift3d_detector_set_corner_thresh(self.h, float(v))

    def set_num_kp_levels(self, v):
        return lib().sift3d_detector_set_num_kp_levels(self.h, int(v))

    def set_sigma_n(self, v):
        return lib().sift3d_detector_set_sigma_n(self.h, float(v))

    def set_sigma0(self, v):
        return lib().sift3d_detector_set_sigma0(self.h, float(v))

    def set_cuboid_extrema(self, on):
        """Run-time form of the reference's compile-time CUBOID_EXTREMA (sift.c:24)."""
        return lib().sift3d_amd_detector_set_cuboid_extrema(self.h, int(bool(on)))

    def set_serial_orientation(self, on):
        """A/B switch: the reference's serial window sums for every candidate (same results bit for bit)."""
        return lib().sift3d_amd_detector_set_serial_orientation(self.h, int(bool(on)))

    def set_exact_descriptors(self, mode):
        """0: automatic (wide windows in the reference's accumulation order), 1: always (descriptors bit-exact
        with the reference), -1: never."""
        return lib().sift3d_amd_detector_set_exact_descriptors(self.h, int(mode))

    def set_dogmax_pass(self, on):
        """A/B switch: octave 0's dogmax scan as a pass of its own (True) or gathered by the extrema sweep."""
        return lib().sift3d_amd_detector_set_dogmax_pass(self.h, int(bool(on)))

    def dogmax(self):
        """max|DoG| of every DoG level of the last detect call (float32, octave-major)."""
        out = np.zeros(1024, np.float32)
        n = lib().sift3d_amd_detector_dogmax(self.h, out.ctypes.data, out.size)
        if n < 0:
            raise RuntimeError("sift3d_amd_detector_dogmax")
        return out[:n].copy()

    def detect_keypoints(self, image, store):
        return lib().sift3d_detect_keypoints(self.h, image.h, store.h)

    def detect_keypoints_device(self, d_ptr, nx, ny, nz, store, units=(1.0, 1.0, 1.0)):
        """sift3d_amd_detect_keypoints_device: the volume is already resident in HBM."""
        return lib().sift3d_amd_detect_keypoints_device(self.h, d_ptr, nx, ny, nz,
                                                        *map(float, units), store.h)

    def extract_descriptors(self, kp_store, desc_store):
        return lib().sift3d_extract_descriptors(self.h, kp_store.h, desc_store.h)

    def timings(self):
        p = lib().sift3d_amd_timings(self.h)
        names = ("scale", "gauss", "dog", "extrema", "orient", "describe", "gauss_dev",
                 "detect_wall", "describe_wall", "yz_last")
        out = dict(zip(names, [p[i] for i in range(len(names))]))
        out["detect_dev"] = p[10 + 2 * TIMED_BLURS]       # first to last stage event of detect
        out["compact_host"] = p[10 + 2 * TIMED_BLURS + 1]  # the host's candidate -> keypoint compaction
        # first stage event -> end of the orientation of octave 0's candidates / of the other octaves' (0: one part)
        out["orient_oct0_end"] = p[10 + 2 * TIMED_BLURS + 2]
        out["orient_rest_end"] = p[10 + 2 * TIMED_BLURS + 3]
        return out

    def describe_clock(self):
        """(shader cycles, seconds) of the fast descriptor kernel of the last extract_descriptors, measured by
        the kernel itself (its first, persistent wave); None when nothing was recorded."""
        c, t = C.c_double(), C.c_double()
        if lib().sift3d_amd_describe_clock(self.h, C.byref(c), C.byref(t)) != 0 or t.value <= 0:
            return None
        return c.value, t.value

    def launch_timings(self):
        """Octave 0's pyramid launches of the last detect, HIP events around each on its stream:
        (x-pass seconds, fused y+z seconds) per blur s = 0 .. ngl-1; 0.0 where a blur did not take the
        fused kernel."""
        p = lib().sift3d_amd_timings(self.h)
        return [(p[10 + b], p[10 + TIMED_BLURS + b]) for b in range(TIMED_BLURS)]

    def build_pyramid_device(self, ptr, nx, ny, nz, units=(1.0, 1.0, 1.0)):
        return lib().sift3d_amd_build_pyramid_device(self.h, ptr, nx, ny, nz, *map(float, units))

    def num_candidates(self):
        return lib().sift3d_amd_num_candidates(self.h)

    def level(self, which, o, s):
        dims = np.zeros(3, np.int32)
        if lib().sift3d_amd_copy_level(self.h, which, o, s, None, dims) != 0:
            raise IndexError("no such level")
        out = np.empty((dims[2], dims[1], dims[0]), np.float32)
        if lib().sift3d_amd_copy_level(self.h, which, o, s, out.ctypes.data, dims) != 0:
            raise RuntimeError("sift3d_amd_copy_level failed")
        return out


def synth_survey(n, nblob=None, seed=0):
    """SURVEY.md 8(d) volume (sequential generator, host)."""
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


def gauss_filter(sigma):
    """Normalised Gaussian taps exactly as the detector computes them (imutil.c:1267-1319)."""
    taps = np.zeros(1024, np.float32)
    w = lib().sift3d_amd_gauss_filter(float(sigma), taps, 1024)
    if w < 1 or w > 1024:
        raise ValueError("gauss_filter(%r)" % sigma)
    return taps[:w].copy()


# ---- registration (BASELINE config 5; parity unpinned: removed from the reference fork) --------
class Matcher:
    """sift3d_amd_matcher: descriptor matching with reusable device scratch."""

    def __init__(self):
        L = lib()
        L.sift3d_amd_make_matcher.restype = C.c_void_p
        L.sift3d_amd_free_matcher.argtypes = [C.c_void_p]
        L.sift3d_amd_free_matcher.restype = None
        L.sift3d_amd_matcher_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                               np.ctypeslib.ndpointer(np.int32)]
        L.sift3d_amd_matcher_seconds.argtypes = [C.c_void_p]
        L.sift3d_amd_matcher_seconds.restype = C.c_double
        self.h = L.sift3d_amd_make_matcher()
        if not self.h:
            raise RuntimeError("sift3d_amd_make_matcher failed (no HIP device?)")

    def match(self, desc_a, desc_b, nn_thresh=0.8):
        """match[i] = index in desc_b of the descriptor matched to descriptor i of desc_a, or -1."""
        out = np.full(max(len(desc_a), 1), -1, np.int32)
        if lib().sift3d_amd_matcher_match(self.h, desc_a.h, desc_b.h, float(nn_thresh), out) != 0:
            raise RuntimeError("sift3d_amd_matcher_match failed")
        return out[:len(desc_a)]

    def seconds(self):
        """Device seconds of the two nearest-neighbour searches of the last match."""
        return float(lib().sift3d_amd_matcher_seconds(self.h))

    def free(self):
        if getattr(self, "h", None) and lib is not None:
            lib().sift3d_amd_free_matcher(self.h)
            self.h = None

    __del__ = free


def nn_match(desc_a, desc_b, nn_thresh=0.8):
    """match[i] = index in desc_b of the descriptor matched to descriptor i of desc_a, or -1."""
    out = np.full(max(len(desc_a), 1), -1, np.int32)
    if lib().sift3d_amd_nn_match(desc_a.h, desc_b.h, float(nn_thresh), out) != 0:
        raise RuntimeError("sift3d_amd_nn_match failed")
    return out[:len(desc_a)]


def ransac_affine(src, dst, err_thresh=5.0, num_iter=500, seed=1):
    """Affine map dst = A [src; 1]: returns (A 3x4, inlier mask)."""
    src = np.ascontiguousarray(src, np.float64).reshape(-1, 3)
    dst = np.ascontiguousarray(dst, np.float64).reshape(-1, 3)
    assert len(src) == len(dst)
    A = np.zeros(12, np.float64)
    inl = np.zeros(max(len(src), 1), np.uint8)
    cnt = C.c_int()
    rc = lib().sift3d_amd_ransac_affine(src.reshape(-1), dst.reshape(-1), len(src), float(err_thresh),
                                        int(num_iter), int(seed), A, inl, C.byref(cnt))
    if rc != 0:
        raise RuntimeError("sift3d_amd_ransac_affine: no model")
    return A.reshape(3, 4), inl[:len(src)].astype(bool)
