"""CPU (not gpu): the C-ABI library loads, exports every symbol the headers declare, and its
host-side logic (objects, parameter checks, stores, converters, CSV writers, host math)
behaves like the reference.  No device compute is called here."""
import ctypes as C
import gzip
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from sift3d_amd import api as a
    a.lib()
    return a


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sift3d_(?:amd_|hip_)?[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(api):
    L = api.lib()
    names = _declared("sift3d/sift.h") + _declared("sift3d/imutil.h") + _declared("sift3d_amd.h")
    assert len(_declared("sift3d/sift.h")) + len(_declared("sift3d/imutil.h")) == 27
    for n in names:
        assert hasattr(L, n), "libsift3d_amd.so does not export %s" % n
    from sift3d_amd import hip
    hip.lib()


def test_reference_abi_names_match_reference_library():
    """Where oracle/_ref exists (build container): same 27 dynamic symbols as libsift3D.so."""
    ref = os.path.join(ROOT, "oracle", "_ref", "libsift3D_ref.so")
    if not os.path.exists(ref):
        pytest.skip("reference build not present")
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", ref]).decode()
    ref_syms = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    mine = _declared("sift3d/sift.h") + _declared("sift3d/imutil.h")
    assert sorted(mine) == ref_syms


def test_host_expf_is_libm_expf(api):
    libm = C.CDLL("libm.so.6")
    libm.expf.restype = C.c_float
    libm.expf.argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.random(200000).astype(np.float32) * 40,
                        rng.random(20000).astype(np.float32) * 10,
                        np.array([0, -0.0, -86.9, -87.1, -200, 88.5], np.float32)])
    got = np.empty_like(x)
    api.lib().sift3d_amd_host_expf(x, got, x.size)
    want = np.array([libm.expf(float(v)) for v in x], np.float32)
    np.testing.assert_array_equal(got, want)


def test_host_eigen3_matches_golden_lapack(api):
    from tests import util
    g = util.load("g2_fir")
    for A, Q, L in zip(g["eig_A"], g["eig_Q"], g["eig_L"]):
        q = np.zeros(9); l = np.zeros(3)
        api.lib().sift3d_amd_host_eigen3(np.ascontiguousarray(A.reshape(9)), q, l)
        q = q.reshape(3, 3)
        np.testing.assert_allclose(l, L, rtol=1e-12, atol=1e-14)
        for j in range(3):
            s = np.sign(np.dot(q[:, j], Q[:, j]))
            np.testing.assert_allclose(s * q[:, j], Q[:, j], atol=1e-12)


def test_objects_and_parameter_checks(api):
    im = api.Image(5, 4, 3)
    assert im.data().shape == (3, 4, 5) and float(np.abs(im.data()).max()) == 0.0
    with pytest.raises(ValueError):
        api.Image(0, 4, 3)
    assert api.lib().sift3d_read_image(b"/nonexistent/file.nii.gz") is None
    assert api.lib().sift3d_read_image(b"/nonexistent/file.txt") is None
    m = api.MatRm()
    assert m.dimensions() == (0, 0) and m.type() == api.SIFT3D_FLOAT
    d = api.Detector()
    assert d.set_peak_thresh(0.0) == -1 and d.set_peak_thresh(1.0) == 0
    assert d.set_peak_thresh(1.01) == -1
    assert d.set_corner_thresh(-0.1) == -1 and d.set_corner_thresh(1.0) == 0
    assert d.set_sigma_n(-1) == -1 and d.set_sigma0(-1) == -1
    assert d.set_sigma_n(5.0) == 0          # no image yet: not checked (imutil.c:1574 loop is empty)
    assert d.set_num_kp_levels(4) == 0
    with pytest.raises(ValueError):
        api.Detector(peak_thresh=2.0)


def _fill(api, n=7):
    rng = np.random.default_rng(4)
    recs = np.zeros(n, api.KP_DTYPE)
    recs["o"] = rng.integers(0, 3, n)
    recs["s"] = rng.integers(0, 3, n)
    recs["xd"], recs["yd"], recs["zd"] = rng.integers(1, 30, (3, n))
    recs["sd"] = 1.6 * 2.0 ** (recs["o"] + recs["s"] / 3.0)
    recs["strength"] = np.array([0.5, 0.25, 0.5, 0.75, 0.1, 0.25, 0.9], np.float32)[:n]
    recs["R"] = rng.standard_normal((n, 3, 3)).astype(np.float32)
    kp = api.KeypointStore()
    assert kp.set_records(recs) == 0
    return kp, recs


def test_keypoint_store_roundtrip_sort_and_matrix(api):
    kp, recs = _fill(api)
    got = kp.records()
    for f in recs.dtype.names:
        np.testing.assert_array_equal(got[f], recs[f])
    m = kp.to_mat_rm()
    assert m.dtype == np.float64 and m.shape == (7, 3)
    np.testing.assert_array_equal(m[:, 0], recs["xd"] * 2.0 ** recs["o"])
    # sort: descending strength, ties keep their order under glibc's merge sort with the
    # never-zero comparator (sift.c:1832-1837)
    kp.sort_by_strength(0)
    s = kp.records()
    assert list(s["strength"]) == sorted(recs["strength"], reverse=True)
    order = np.argsort(-recs["strength"], kind="stable")
    np.testing.assert_array_equal(s["xd"], recs["xd"][order])
    kp.sort_by_strength(3)
    assert len(kp) == 3
    kp.sort_by_strength(100)
    assert len(kp) == 3


def test_csv_writers(api, tmp_path):
    kp, recs = _fill(api)
    p = str(tmp_path / "kp.csv")
    assert kp.save(p) == 0
    rows = [l.split(",") for l in open(p).read().strip().split("\n")]
    assert len(rows) == 7 and all(len(r) == 15 for r in rows)   # strength,x,y,z,o,sd,R00..R22
    assert rows[0][0] == "%f" % recs["strength"][0]
    assert rows[2][1] == "%f" % recs["xd"][2] and rows[2][4] == "%f" % recs["o"][2]
    assert rows[3][6 + 5] == "%f" % float(recs["R"][3].reshape(9)[5])
    pz = str(tmp_path / "kp.csv.gz")
    assert kp.save(pz) == 0
    assert gzip.open(pz, "rt").read() == open(p).read()
    assert kp.save(str(tmp_path / "nodir" / "x.csv")) == -1
    d = api.DescriptorStore()
    assert d.save(str(tmp_path / "d.csv")) == -1                 # empty store (sift.c:1691)
    with pytest.raises(RuntimeError):
        d.to_mat_rm()


def test_csv_writers_against_reference_files(api, tmp_path):
    """sift3d_keypoint_store_save / sift3d_descriptor_store_save against files the UNMODIFIED
    reference wrote for the same records (tests/golden/g4_csv.npz, made by oracle/make_golden.py
    g4_csv: sift.c:1741-1830, write_Mat_rm imutil.c:405-479) -- byte for byte, plain and gzip."""
    import hashlib
    from sift3d_amd.api import KP_DTYPE
    from tests import util
    g = util.load("g4_csv")
    n = len(g["kp_strength"])
    recs = np.zeros(n, KP_DTYPE)
    recs["o"], recs["s"] = g["kp_os"][:, 0], g["kp_os"][:, 1]
    for i, f in enumerate(("xd", "yd", "zd", "sd")):
        recs[f] = g["kp_xyzsd"][:, i]
    recs["strength"] = g["kp_strength"]
    recs["R"] = g["kp_R"].reshape(n, 3, 3)
    kp = api.KeypointStore()
    assert kp.set_records(recs) == 0
    want = bytes(g["kp_csv"])
    p = str(tmp_path / "kp.csv")
    assert kp.save(p) == 0
    assert open(p, "rb").read() == want
    assert kp.save(p + ".gz") == 0
    assert gzip.open(p + ".gz", "rb").read() == want
    d = api.DescriptorStore()
    assert d.set(g["desc_xyzsd"], g["desc_hist"], tuple(int(v) for v in g["dims"])) == 0
    q = str(tmp_path / "desc.csv")
    assert d.save(q) == 0
    got = open(q, "rb").read()
    assert len(got) == int(g["desc_csv_bytes"])
    rows = got.split(b"\n")
    assert rows[0] == bytes(g["desc_csv_first_row"])
    assert (rows[-2] if rows[-1] == b"" else rows[-1]) == bytes(g["desc_csv_last_row"])
    assert got.endswith(b"\n") == bool(g["desc_csv_ends_with_newline"])
    assert hashlib.sha1(got).hexdigest() == str(g["desc_csv_sha1"])
    assert d.save(q + ".gz") == 0
    assert gzip.open(q + ".gz", "rb").read() == got
    # the same rows through the matrix converter (sift3d_descriptor_store_to_mat_rm)
    m = d.to_mat_rm()
    np.testing.assert_array_equal(m[:, 3:], g["desc_hist"])
    np.testing.assert_array_equal(m[:, :3], g["desc_xyzsd"][:, :3].astype(np.float32))


def test_hot_path_fails_loudly_without_device(api):
    if api.device_available():
        pytest.skip("a device is present")
    d = api.Detector()
    kp = api.KeypointStore()
    assert d.detect_keypoints(api.Image.from_array(np.ones((16, 16, 16), np.float32)), kp) == -1
    assert d.extract_descriptors(kp, api.DescriptorStore()) == -1


def test_synth_generators_match_oracle_build(api, oracle_mod):
    np.testing.assert_array_equal(api.synth_survey(24), oracle_mod.synth_survey(24))
    np.testing.assert_array_equal(api.synth_lattice((20, 12, 9), seed=5),
                                  oracle_mod.synth_lattice((20, 12, 9), seed=5))


# ----------------------------------------------------------------------------------------
# sift3d_read_image: single-file NIFTI-1 (the reference reads through nifticlib,
# nifti.c:52-167; no reference build with nifticlib exists here, so these tests pin the reader
# against files written from the NIFTI-1 specification)
# ----------------------------------------------------------------------------------------
_NII_TYPES = {np.dtype("u1"): 2, np.dtype("i1"): 256, np.dtype("i2"): 4, np.dtype("u2"): 512,
              np.dtype("i4"): 8, np.dtype("u4"): 768, np.dtype("i8"): 1024, np.dtype("u8"): 1280,
              np.dtype("f4"): 16, np.dtype("f8"): 64}


def _write_nii(path, arr, pixdim=(1.0, 1.0, 1.0), slope=0.0, inter=0.0, big_endian=False,
               magic=b"n+1\0", vox_offset=352):
    """arr: [nz, ny, nx] or [nt, nz, ny, nx] (file order: x fastest)."""
    import gzip
    import struct
    e = ">" if big_endian else "<"
    dims = list(arr.shape[::-1])
    dim = [len(dims)] + dims + [1] * (7 - len(dims))
    hdr = bytearray(348)
    struct.pack_into(e + "i", hdr, 0, 348)
    struct.pack_into(e + "8h", hdr, 40, *dim)
    struct.pack_into(e + "h", hdr, 70, _NII_TYPES[arr.dtype])
    struct.pack_into(e + "h", hdr, 72, arr.dtype.itemsize * 8)
    struct.pack_into(e + "8f", hdr, 76, 1.0, *pixdim, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into(e + "f", hdr, 108, float(vox_offset))
    struct.pack_into(e + "f", hdr, 112, slope)
    struct.pack_into(e + "f", hdr, 116, inter)
    hdr[344:348] = magic
    body = bytes(hdr) + b"\0" * (vox_offset - 348) + arr.astype(arr.dtype.newbyteorder(e)).tobytes()
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(body)


@pytest.mark.parametrize("dtype", ["u1", "i1", "i2", "u2", "i4", "u4", "i8", "u8", "f4", "f8"])
def test_read_image_nifti_types(api, tmp_path, dtype):
    rng = np.random.default_rng(5)
    dt = np.dtype(dtype)
    if dt.kind == "f":
        a = rng.standard_normal((5, 6, 7)).astype(dt)
    else:
        info = np.iinfo(dt)
        a = rng.integers(max(info.min, -10 ** 6), min(info.max, 10 ** 6), (5, 6, 7)).astype(dt)
    p = tmp_path / ("t_%s.nii" % dtype)
    _write_nii(p, a, pixdim=(0.5, 1.25, 3.0), slope=2.5, inter=-1.0)
    im = api.Image.read(str(p))
    assert im.shape == (5, 6, 7) and im.units == (0.5, 1.25, 3.0)
    want = (a.astype(np.float64) * 2.5 + (-1.0)).astype(np.float32)   # nifti.c:112-116
    np.testing.assert_array_equal(im.data(), want)


def test_read_image_nifti_variants(api, tmp_path):
    rng = np.random.default_rng(6)
    a = rng.standard_normal((4, 5, 9)).astype(np.float32)
    # gzip, big-endian, slope 0 = identity, data not directly after the header
    p = tmp_path / "v.nii.gz"
    _write_nii(p, a, big_endian=True, vox_offset=400)
    im = api.Image.read(str(p))
    np.testing.assert_array_equal(im.data(), a)
    assert im.units == (1.0, 1.0, 1.0)
    # a 4th dimension becomes channels, stored innermost (nifti.c:44-46,97)
    b = rng.integers(0, 100, (3, 4, 5, 6)).astype(np.int16)
    p = tmp_path / "c.nii"
    _write_nii(p, b)
    im = api.Image.read(str(p))
    assert im.shape == (4, 5, 6, 3)
    np.testing.assert_array_equal(im.data(), np.moveaxis(b, 0, -1).astype(np.float32))
    # trailing singleton dimensions are ignored; a real 5th dimension is refused (nifti.c:76-80)
    p = tmp_path / "s.nii"
    _write_nii(p, a[None, None])
    assert api.Image.read(str(p)).shape == (4, 5, 9)
    p = tmp_path / "5d.nii"
    _write_nii(p, np.zeros((2, 1, 3, 3, 3), np.float32))
    with pytest.raises(IOError):
        api.Image.read(str(p))
    # header/image pairs, truncated files, unknown sample types, wrong extension
    p = tmp_path / "pair.nii"
    _write_nii(p, a, magic=b"ni1\0")
    with pytest.raises(IOError):
        api.Image.read(str(p))
    p = tmp_path / "trunc.nii"
    _write_nii(p, a)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-10])
    with pytest.raises(IOError):
        api.Image.read(str(p))
    p = tmp_path / "cplx.nii"
    _write_nii(p, a)
    raw = bytearray(open(p, "rb").read())
    raw[70:72] = (32).to_bytes(2, "little")                 # NIFTI_TYPE_COMPLEX64
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        api.Image.read(str(p))
    with pytest.raises(IOError):
        api.Image.read(str(tmp_path / "missing.nii"))
    with pytest.raises(IOError):
        api.Image.read(str(tmp_path / "x.img"))


def test_bench_contract_helpers():
    """bench.py's algorithmic byte model is SURVEY.md 8(d)'s (24 B per voxel and blur; 6 blurs on octave 0, 5 on
    every later one: 21.63 GB at 512^3, 2.70 GB at 256^3, 42.2 MB at 64^3), and the counts every run checks itself
    against are the reference's (tests/golden/g5_512.npz) where the reference ran the workload."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert abs(b.pyramid_algorithmic_bytes(512, 512, 512) / 1e9 - 21.63) < 0.01
    assert abs(b.pyramid_algorithmic_bytes(256, 256, 256) / 1e9 - 2.70) < 0.01
    assert abs(b.pyramid_algorithmic_bytes(64, 64, 64) / 1e6 - 42.2) < 0.1
    assert b.HBM_PEAK_GBS == 8000.0
    assert b.expected_counts(False, 512, 1) == b.expected_counts(True, 512, 1)
    assert b.expected_counts(True, 1024, 8) == (1249357, 332413) and b.expected_counts(False, 300, 1) is None
    g = np.load(os.path.join(ROOT, "tests", "golden", "g5_512.npz"))
    assert b.expected_counts(False, 512, 1) == (int(g["ncand"]), int(g["nkp"]))
