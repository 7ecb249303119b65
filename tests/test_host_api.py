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
