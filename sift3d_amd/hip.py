"""ctypes bindings of the device-level stage ABI (include/sift3d_amd.h, `sift3d_hip_*`).

Device buffers are torch tensors on the current HIP device (torch is only the allocator /
stream / process-group plumbing); every call takes raw device pointers and runs on torch's
current stream, so results are ordered with other torch work on that stream.
"""
import ctypes as C

import numpy as np

from . import _native

MAX_TAPS = 65
FACE_FLOATS = 19


class FirArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("nx", C.c_int), ("ny", C.c_int),
                ("nz", C.c_int), ("axis", C.c_int), ("width", C.c_int),
                ("taps", C.POINTER(C.c_float)), ("unit_factor", C.c_float), ("n_glob", C.c_int),
                ("off", C.c_int), ("z_lo", C.c_int), ("z_hi", C.c_int), ("variant", C.c_int)]


class ExtremaLevel(C.Structure):
    _fields_ = [("prev", C.c_void_p), ("cur", C.c_void_p), ("next", C.c_void_p),
                ("d_absmax", C.c_void_p), ("z_lo", C.c_int), ("z_hi", C.c_int), ("tag", C.c_int)]


class Level(C.Structure):
    _fields_ = [("data", C.c_void_p), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("z_off", C.c_int), ("nz_glob", C.c_int), ("ux", C.c_float), ("uy", C.c_float),
                ("uz", C.c_float), ("octave", C.c_int), ("sd", C.c_double)]


CAND_DTYPE = np.dtype([("idx", "u4"), ("tag", "i4"), ("val", "f4")])
KP_DTYPE = np.dtype([("R", "f4", (9,)), ("cx", "f4"), ("cy", "f4"), ("cz", "f4"),
                     ("level", "i4"), ("row1", "u4"), ("sd", "f8")], align=True)
LEVEL_DTYPE = np.dtype([("data", "u8"), ("nx", "i4"), ("ny", "i4"), ("nz", "i4"),
                        ("z_off", "i4"), ("nz_glob", "i4"), ("ux", "f4"), ("uy", "f4"),
                        ("uz", "f4"), ("octave", "i4"), ("sd", "f8")], align=True)
assert LEVEL_DTYPE.itemsize == C.sizeof(Level)
assert KP_DTYPE.itemsize == 64 and CAND_DTYPE.itemsize == 12

_bound = None


def lib():
    global _bound
    if _bound is not None:
        return _bound
    L = _native.load()
    vp = C.c_void_p
    sig = {
        "sift3d_hip_device_count": (C.c_int, []),
        "sift3d_hip_set_device": (C.c_int, [C.c_int]),
        "sift3d_hip_malloc": (vp, [C.c_size_t]),
        "sift3d_hip_free": (None, [vp]),
        "sift3d_hip_host_alloc": (vp, [C.c_size_t]),
        "sift3d_hip_host_free": (None, [vp]),
        "sift3d_hip_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "sift3d_hip_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "sift3d_hip_memcpy_d2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "sift3d_hip_memset": (C.c_int, [vp, C.c_int, C.c_size_t, vp]),
        "sift3d_hip_stream_create": (vp, []),
        "sift3d_hip_stream_destroy": (None, [vp]),
        "sift3d_hip_stream_sync": (C.c_int, [vp]),
        "sift3d_hip_event_create": (vp, []),
        "sift3d_hip_event_destroy": (None, [vp]),
        "sift3d_hip_event_record": (C.c_int, [vp, vp]),
        "sift3d_hip_event_elapsed_ms": (C.c_double, [vp, vp]),
        "sift3d_hip_absmax": (C.c_int, [vp, C.c_size_t, vp, vp]),
        "sift3d_hip_scale": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
        "sift3d_hip_fir": (C.c_int, [C.POINTER(FirArgs), vp]),
        "sift3d_hip_fir_yz_u1": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "sift3d_hip_nn2_work_floats": (C.c_size_t, [C.c_int, C.c_int]),
        "sift3d_hip_nn2": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
        "sift3d_hip_subtract_absmax": (C.c_int, [vp, vp, vp, C.c_size_t, vp, vp]),
        "sift3d_hip_dog_stack": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_size_t, vp, vp]),
        "sift3d_hip_host_device_ptr": (vp, [vp]),
        "sift3d_hip_downsample2": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
        "sift3d_hip_extrema_work_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
        "sift3d_hip_extrema": (C.c_int, [C.POINTER(ExtremaLevel), C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_double, vp, C.c_uint32, vp, vp, C.c_size_t, vp]),
        "sift3d_hip_extrema_mode": (C.c_int, [C.POINTER(ExtremaLevel), C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_int, vp, C.c_uint32, vp, vp, C.c_size_t, vp]),
        "sift3d_hip_orient": (C.c_int, [vp, vp, C.c_uint32, C.c_double, vp, vp, vp]),
        "sift3d_hip_orient_tab_bytes": (C.c_size_t, [C.c_int, C.c_uint32]),
        "sift3d_hip_orient_tab": (C.c_int, [vp, C.c_int, vp, C.c_uint32, C.c_double, vp, vp, vp, C.c_uint32, vp]),
        "sift3d_hip_orient_tab_part": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_uint32, C.c_uint32,
                                                 C.c_double, vp, vp, vp, C.c_uint32, C.c_int, vp]),
        "sift3d_hip_describe": (C.c_int, [vp, vp, C.c_uint32, vp, vp]),
        "sift3d_hip_set_mesh": (C.c_int, [C.POINTER(C.c_float)]),
        "sift3d_hip_synth_lattice": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, vp]),
        "sift3d_hip_test_expf": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "sift3d_hip_test_eigen3": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
        "sift3d_hip_last_error": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _bound = L
    return L


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, lib().sift3d_hip_last_error().decode()))


_stream = None


def current_stream(refresh=False):
    """torch's current HIP stream as a void*.  Looked up once and cached (the query costs
    ~0.1 ms of host time, far more than a kernel launch); call current_stream(refresh=True)
    after switching torch's current stream."""
    global _stream
    if _stream is None or refresh:
        import torch
        _stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return _stream


def fir(src, dst, axis, taps, unit_factor=1.0, n_glob=None, off=0, z_lo=0, z_hi=None, variant=0):
    """One 1-D pass (convolve_sep_gen, imutil.c:742-861) on torch CUDA tensors [nz, ny, nx]."""
    nz, ny, nx = src.shape
    assert src.is_contiguous() and dst.is_contiguous() and src.shape == dst.shape
    taps = np.ascontiguousarray(taps, np.float32)
    a = FirArgs(src.data_ptr(), dst.data_ptr(), nx, ny, nz, axis, len(taps),
                taps.ctypes.data_as(C.POINTER(C.c_float)), float(np.float32(unit_factor)),
                nz if n_glob is None else n_glob, off, z_lo, nz if z_hi is None else z_hi, variant)
    _check(lib().sift3d_hip_fir(C.byref(a), current_stream()), "sift3d_hip_fir")
    return dst


def fir_yz(src, dst, taps, n_glob=None, off=0, z_lo=0, z_hi=None):
    """Fused y+z passes with tap spacing 1.  Returns False when the configuration is not covered
    (the caller then issues the two passes separately)."""
    nz, ny, nx = src.shape
    assert src.is_contiguous() and dst.is_contiguous() and src.shape == dst.shape
    taps = np.ascontiguousarray(taps, np.float32)
    rc = lib().sift3d_hip_fir_yz_u1(src.data_ptr(), dst.data_ptr(), nx, ny, nz,
                                    taps.ctypes.data_as(C.POINTER(C.c_float)), len(taps),
                                    nz if n_glob is None else n_glob, off, z_lo,
                                    nz if z_hi is None else z_hi, current_stream())
    if rc == 1:
        return False
    _check(rc, "sift3d_hip_fir_yz_u1")
    return True


def nn2(a, b):
    """Nearest / second-nearest row of b for every row of a (CUDA float32 tensors [n, dim]):
    returns (index int32, squared distance, second squared distance) as tensors."""
    import torch
    assert a.is_contiguous() and b.is_contiguous() and a.shape[1] == b.shape[1]
    na, nb = a.shape[0], b.shape[0]
    j = torch.empty(max(na, 1), dtype=torch.int32, device=a.device)
    d1 = torch.empty(max(na, 1), dtype=torch.float32, device=a.device)
    d2 = torch.empty_like(d1)
    work = torch.empty(lib().sift3d_hip_nn2_work_floats(na, nb), dtype=torch.float32, device=a.device)
    _check(lib().sift3d_hip_nn2(a.data_ptr(), na, b.data_ptr(), nb, a.shape[1], j.data_ptr(),
                                d1.data_ptr(), d2.data_ptr(), work.data_ptr(), current_stream()),
           "sift3d_hip_nn2")
    return j[:na], d1[:na], d2[:na]


def absmax(src, out):
    _check(lib().sift3d_hip_absmax(src.data_ptr(), src.numel(), out.data_ptr(), current_stream()),
           "sift3d_hip_absmax")


def scale(src, dst, d_max):
    _check(lib().sift3d_hip_scale(src.data_ptr(), dst.data_ptr(), src.numel(), d_max.data_ptr(),
                                  current_stream()), "sift3d_hip_scale")


def subtract_absmax(a, b, dst, d_absmax=None):
    _check(lib().sift3d_hip_subtract_absmax(a.data_ptr(), b.data_ptr(), dst.data_ptr(), a.numel(),
                                            d_absmax.data_ptr() if d_absmax is not None else None,
                                            current_stream()), "sift3d_hip_subtract_absmax")


def dog_stack(gauss, dogs, d_absmax):
    """dogs[k] = gauss[k] - gauss[k+1] for one octave in a single pass; d_absmax: float32 tensor
    of len(gauss)-1 running maxima.  Returns False when the kernel does not cover the case."""
    n = len(gauss)
    assert len(dogs) == n - 1 and d_absmax.numel() >= n - 1
    g = (C.c_void_p * n)(*[t.data_ptr() for t in gauss])
    d = (C.c_void_p * (n - 1))(*[t.data_ptr() for t in dogs])
    rc = lib().sift3d_hip_dog_stack(g, d, n, gauss[0].numel(), d_absmax.data_ptr(), current_stream())
    if rc == 1:
        return False
    _check(rc, "sift3d_hip_dog_stack")
    return True


def downsample2(src, dst):
    nz, ny, nx = src.shape
    mz, my, mx = dst.shape
    _check(lib().sift3d_hip_downsample2(src.data_ptr(), nx, ny, dst.data_ptr(), mx, my, mz,
                                        current_stream()), "sift3d_hip_downsample2")


def synth_lattice(dst, z_off=0, seed=1):
    nz, ny, nx = dst.shape
    _check(lib().sift3d_hip_synth_lattice(dst.data_ptr(), nx, ny, nz, z_off, seed, current_stream()),
           "sift3d_hip_synth_lattice")
    return dst


def level_table(levels):
    """levels: list of dicts(data=tensor, off, nz_glob, units, octave, sd) -> device table
    (torch uint8 tensor) + the numpy mirror."""
    import torch
    tab = np.zeros(len(levels), LEVEL_DTYPE)
    for i, L in enumerate(levels):
        t = L["data"]
        tab[i] = (t.data_ptr(), t.shape[2], t.shape[1], t.shape[0], L["off"], L["nz_glob"],
                  np.float32(L["units"][0]), np.float32(L["units"][1]), np.float32(L["units"][2]),
                  L["octave"], L["sd"])
    dev = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
    return dev, tab


def extrema(levels, nx, ny, nz, peak_thresh, cap=1 << 18, cuboid=False):
    """levels: list of dict(prev, cur, next (tensors), absmax (1-elem tensor), z_lo, z_hi, tag).
    Returns CAND_DTYPE records in the reference's scan order."""
    import torch
    L = lib()
    arr = (ExtremaLevel * len(levels))()
    for i, lv in enumerate(levels):
        arr[i] = ExtremaLevel(lv["prev"].data_ptr(), lv["cur"].data_ptr(), lv["next"].data_ptr(),
                              lv["absmax"].data_ptr(), lv["z_lo"], lv["z_hi"], lv["tag"])
    wb = L.sift3d_hip_extrema_work_bytes(nx, ny, nz, len(levels))
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    while True:
        out = torch.empty(cap * CAND_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        count.zero_()
        _check(L.sift3d_hip_extrema_mode(arr, len(levels), nx, ny, nz, float(peak_thresh),
                                         int(bool(cuboid)), out.data_ptr(), cap, count.data_ptr(),
                                         work.data_ptr(), wb, current_stream()),
               "sift3d_hip_extrema_mode")
        n = int(count.item())
        if n <= cap:
            break
        cap = n + n // 4 + 1024
    return out[:n * CAND_DTYPE.itemsize].cpu().numpy().view(CAND_DTYPE).copy()



def orient(d_levels, cands, corner_thresh):
    import torch
    n = len(cands)
    if n == 0:
        return np.zeros((0, 9), np.float32), np.zeros(0, np.int32)
    dc = torch.from_numpy(np.ascontiguousarray(cands).view(np.uint8)).cuda()
    R = torch.empty((n, 9), dtype=torch.float32, device="cuda")
    keep = torch.empty(n, dtype=torch.int32, device="cuda")
    _check(lib().sift3d_hip_orient(d_levels.data_ptr(), dc.data_ptr(), n, float(corner_thresh),
                                   R.data_ptr(), keep.data_ptr(), current_stream()),
           "sift3d_hip_orient")
    return R.cpu().numpy(), keep.cpu().numpy()


def orient_tab(d_levels, nlevels, cands, corner_thresh, parts=None):
    """sift3d_hip_orient_tab over the whole list, or -- parts = [(lv_lo, lv_hi, first, n), (..)] -- the list
    in two parts that run at the same time on two streams (sift3d_hip_orient_tab_part)."""
    import torch
    n = len(cands)
    L = lib()
    dc = torch.from_numpy(np.ascontiguousarray(cands).view(np.uint8)).cuda()
    R = torch.zeros((n, 9), dtype=torch.float32, device="cuda")
    keep = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    tab = torch.zeros(L.sift3d_hip_orient_tab_bytes(nlevels, n), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    if parts is None:
        _check(L.sift3d_hip_orient_tab(d_levels.data_ptr(), nlevels, dc.data_ptr(), n, float(corner_thresh),
                                       R.data_ptr(), keep.data_ptr(), tab.data_ptr(), n, current_stream()),
               "sift3d_hip_orient_tab")
    else:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for slot, (lv_lo, lv_hi, first, m) in enumerate(parts):
            _check(L.sift3d_hip_orient_tab_part(d_levels.data_ptr(), nlevels, lv_lo, lv_hi, dc.data_ptr(), first, m,
                                                float(corner_thresh), R.data_ptr(), keep.data_ptr(),
                                                tab.data_ptr(), n, slot, streams[slot].cuda_stream),
                   "sift3d_hip_orient_tab_part")
    torch.cuda.synchronize()
    return R.cpu().numpy(), keep.cpu().numpy()


def describe(d_levels, kps):
    import torch
    n = len(kps)
    if n == 0:
        return np.zeros((0, 768), np.float32)
    dk = torch.from_numpy(np.ascontiguousarray(kps).view(np.uint8)).cuda()
    hist = torch.empty((n, 768), dtype=torch.float32, device="cuda")
    _check(lib().sift3d_hip_describe(d_levels.data_ptr(), dk.data_ptr(), n, hist.data_ptr(),
                                     current_stream()), "sift3d_hip_describe")
    return hist.cpu().numpy()
