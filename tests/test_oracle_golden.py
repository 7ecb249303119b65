"""CPU (not gpu): pin the oracle (oracle/sift3d_oracle.c) to the reference's golden vectors.

The fixtures under tests/golden/ are OUTPUTS of the unmodified reference (fatimp/SIFT3D
v2.0 compiled by oracle/Makefile, dumped by oracle/make_golden.py).  The bar is the one
BASELINE.json states: bit-exact float32 pyramid / candidate / keypoint lists; R and
descriptors within 1e-5 relative (in practice the restatement is bit-exact there too).
"""
import json

import numpy as np
import pytest

from tests import util

E2E = ["g3_64", "g3_70x50x41", "g3_aniso", "g3_params", "g3_lattice48", "g3_cuboid64",
       "g3_cuboid_params",   # the last two: CUBOID_EXTREMA build of the reference (sift.c:24)
       "g3_sigma3", "g3_sigma5", "g3_switch285", "g3_switch_aniso",
       "g3_sigma8"]                # octave filters of 31 ... 75 taps   # wide windows (sift.c:1453-1456): a bin receives ~7x / ~30x the terms
BIG = [n for n in ("g5_128",) if util.have(n)]


def test_g1_gauss_taps(oracle_mod):
    g = util.load("g1_filters")
    for i, s in enumerate(g["sigmas"]):
        np.testing.assert_array_equal(oracle_mod.gauss_taps(s), g["taps_%d" % i])
    o = oracle_mod.Oracle()
    assert o.set_volume(oracle_mod.synth_survey(16, nblob=4)) == 0
    bank = o.filters()
    assert len(bank) == len(g["bank_sigma"]) == 6
    for i, (s, t) in enumerate(bank):
        assert s == g["bank_sigma"][i]
        np.testing.assert_array_equal(t, g["bank_%d" % i])
    # widths quoted in SURVEY.md section 8(a5)
    assert [len(t) for _, t in bank] == [5, 7, 9, 11, 13, 17]


@pytest.mark.parametrize("mode", [0, 1])
def test_g2_fir_axis(oracle_mod, mode):
    g = util.load("g2_fir")
    vol = g["vol"]
    n = 0
    for key in g.files:
        if not key.startswith("axis"):
            continue
        ax = int(key[4])
        _, w, u = key.split("_")
        taps = g["taps_asym"] if w == "asym" else g["taps_" + w]
        uf = np.float32(1.0 / float(u[1:]))
        out, r = oracle_mod.fir_axis(vol, taps, ax, uf=uf, mode=mode)
        assert r == 0
        np.testing.assert_array_equal(out, g[key], err_msg=key)
        n += 1
    assert n == 33


@pytest.mark.parametrize("mode", [0, 1])
def test_g2_blur3(oracle_mod, mode):
    g = util.load("g2_fir")
    vol = g["vol2"]
    cases = {"iso1": ((1, 1, 1), 1.0), "iso2": ((2, 2, 2), 1.0), "iso4": ((4, 4, 4), 1.0),
             "aniso": (tuple(g["units_aniso"]), 1.0), "aniso3": (tuple(g["units_aniso3"]), 1.0),
             "default": ((2, 2, 2), -1.0)}
    for name, (units, unit) in cases.items():
        for w in (5, 17):
            out = oracle_mod.blur(vol, g["taps_w%d" % w], units=units, unit=unit, mode=mode)
            np.testing.assert_array_equal(out, g["blur_%s_w%d" % (name, w)], err_msg=name)
    np.testing.assert_array_equal(oracle_mod.downsample(vol), g["down"])
    np.testing.assert_array_equal(oracle_mod.downsample(g["vol"]), g["down_odd"])


def test_g2_fir_slab_equals_whole(oracle_mod):
    """Slab form (global mirror rules + halo) reproduces the unsharded result bit-for-bit."""
    g = util.load("g2_fir")
    vol = g["vol2"]  # nz = 17
    for w, u in ((5, 1.0), (17, 1.0), (17, 2.0), (9, 4.0)):
        taps = g["taps_w%d" % w]
        uf = np.float32(1.0 / u)
        whole, r = oracle_mod.fir_axis(vol, taps, 2, uf=uf)
        assert r == 0
        reach = int(np.ceil((w // 2) * uf)) + 1
        for z0, z1 in ((0, 6), (6, 11), (11, 17)):
            lo, hi = max(0, z0 - reach), min(17, z1 + reach)
            part, r = oracle_mod.fir_axis(vol[lo:hi], taps, 2, uf=uf, n_glob=17, off=lo,
                                          out_lo=z0 - lo, out_hi=z1 - lo, mode=1)
            assert r == 0
            np.testing.assert_array_equal(part[z0 - lo:z1 - lo], whole[z0:z1])
        # a halo that is too thin must be reported, not silently clamped
        if reach > 1 and w > 5:
            _, r = oracle_mod.fir_axis(vol[6:11], taps, 2, uf=uf, n_glob=17, off=6, out_lo=0,
                                       out_hi=5, mode=1)
            assert r != 0


def test_g2_eigen(oracle_mod):
    g = util.load("g2_fir")
    for A, Q, L in zip(g["eig_A"], g["eig_Q"], g["eig_L"]):
        q, l = oracle_mod.eigen3(A)
        np.testing.assert_allclose(l, L, rtol=1e-12, atol=1e-14)
        for j in range(3):  # sign of an eigenvector is arbitrary (LAPACK's too)
            s = np.sign(np.dot(q[:, j], Q[:, j]))
            np.testing.assert_allclose(s * q[:, j], Q[:, j], rtol=0, atol=1e-12)
        np.testing.assert_allclose(q @ np.diag(l) @ q.T, A, atol=1e-12)


def _run(oracle_mod, g, mode=1):
    vol = util.golden_input(g, oracle_mod)
    o = oracle_mod.Oracle(fir_mode=mode, **util.golden_params(g))
    assert o.detect(vol, tuple(g["units"])) == 0
    return o, vol


def check_detect_against_golden(g, num_octaves, level_fn, cand, kp, K):
    """Shared by the oracle tests and the GPU tests (tests/test_gpu_parity.py)."""
    assert num_octaves == int(g["num_octaves"])
    dig = json.loads(str(g["digests"]))
    for key, want in dig.items():
        if key == "IM":
            a = level_fn(2, 0, 0)
        else:
            which = 0 if key[0] == "G" else 1
            o, s = key[2:].split("_")
            a = level_fn(which, int(o[1:]), int(s[1:]))
        assert util.digest(a) == want, "level %s differs from the reference" % key
        if "level_" + key in g.files:
            np.testing.assert_array_equal(a, g["level_" + key])
    np.testing.assert_array_equal(np.stack([cand[k] for k in "osxyz"], 1), g["cand_osxyz"])
    np.testing.assert_array_equal(cand["strength"], g["cand_strength"])
    np.testing.assert_array_equal(cand["sd"], g["cand_sd"])
    np.testing.assert_array_equal(np.stack([kp["o"], kp["s"]], 1), g["kp_os"])
    np.testing.assert_array_equal(np.stack([kp[k] for k in ("xd", "yd", "zd", "sd")], 1),
                                  g["kp_xyzsd"])
    # stale strength quirk Q2 is part of the contract
    np.testing.assert_array_equal(kp["strength"], g["kp_strength"])
    assert util.rel_err(kp["R"], g["kp_R"]) <= 1e-5


@pytest.mark.parametrize("name", E2E + BIG)
def test_end_to_end(oracle_mod, name):
    g = util.load(name)
    o, _ = _run(oracle_mod, g)
    K = util.golden_params(g).get("num_kp_levels", 3)
    check_detect_against_golden(g, o.num_octaves, lambda w, oc, s: o.level(w, oc, s)[0],
                                o.candidates(), o.keypoints(), K)
    kp = o.keypoints()
    # the restatement is in fact bit-exact for R
    np.testing.assert_array_equal(kp["R"], g["kp_R"])
    np.testing.assert_array_equal(o.kp_mat(), g["kp_mat"])
    assert o.describe() == 0
    d = o.descriptors()
    np.testing.assert_array_equal(np.stack([d[k] for k in ("xd", "yd", "zd", "sd")], 1),
                                  g["desc_xyzsd"])
    idx = g["desc_idx"]
    assert util.rel_err(d["hist"][idx], g["desc_hist"]) <= 1e-5
    np.testing.assert_array_equal(d["hist"][idx], g["desc_hist"])
    np.testing.assert_allclose(d["hist"].astype(np.float64).sum(1), g["desc_rowsum"], rtol=1e-6)
    assert util.assert_desc_projection(d["hist"], g["desc_proj"], rtol=1e-12) < 1e-12   # every row
    for lim in (0, 10):
        o2, _ = _run(oracle_mod, g)
        o2.sort_by_strength(lim)
        k2 = o2.keypoints()
        np.testing.assert_array_equal(np.stack([k2["o"], k2["s"]], 1), g["sort%d_os" % lim])
        np.testing.assert_array_equal(np.stack([k2[k] for k in ("xd", "yd", "zd", "sd")], 1),
                                      g["sort%d_xyzsd" % lim])
        np.testing.assert_array_equal(k2["strength"], g["sort%d_strength" % lim])


def test_literal_fir_mode_matches_restructured(oracle_mod):
    g = util.load("g3_70x50x41")
    a, _ = _run(oracle_mod, g, mode=0)
    b, _ = _run(oracle_mod, g, mode=1)
    for oc in range(a.num_octaves):
        for s in range(-1, 5):
            np.testing.assert_array_equal(a.level(0, oc, s)[0], b.level(0, oc, s)[0])


def test_mesh_quirk_q1(oracle_mod):
    """All 20 faces have v[0]<->v[1] swapped while idx[] keeps table order."""
    o = oracle_mod.Oracle()
    v, idx = o.mesh()
    g = 1.6180339887
    vert = np.array([[0, 1, g], [0, -1, g], [0, 1, -g], [0, -1, -g], [1, g, 0], [-1, g, 0],
                     [1, -g, 0], [-1, -g, 0], [g, 0, 1], [-g, 0, 1], [g, 0, -1], [-g, 0, -1]])
    vert /= np.linalg.norm(vert, axis=1, keepdims=True)
    for f in range(20):
        np.testing.assert_allclose(v[f, 0], vert[idx[f, 1]], atol=1e-6)
        np.testing.assert_allclose(v[f, 1], vert[idx[f, 0]], atol=1e-6)
        np.testing.assert_allclose(v[f, 2], vert[idx[f, 2]], atol=1e-6)


def test_error_behaviour(oracle_mod):
    """Failure cases of SURVEY.md section 8(b) 'Errors'."""
    o = oracle_mod.Oracle()
    assert o.L.orc_set_peak_thresh(o.h, 0.0) != 0
    assert o.L.orc_set_peak_thresh(o.h, 1.5) != 0
    assert o.L.orc_set_corner_thresh(o.h, -0.1) != 0
    assert o.L.orc_set_sigma_n(o.h, -1.0) != 0
    assert o.L.orc_set_sigma0(o.h, -1.0) != 0
    assert o.detect(np.zeros((7, 16, 16), np.float32)) != 0  # a dimension < 8
    o2 = oracle_mod.Oracle()
    assert o2.detect(np.zeros((16, 16, 16), np.float32)) == 0  # all-zero volume: no keypoints
    assert len(o2.keypoints()) == 0
    assert o2.describe() != 0  # describe with zero keypoints fails (sift.c:1178-1182)
    o3 = oracle_mod.Oracle()
    assert o3.set_volume(oracle_mod.synth_survey(16, nblob=4)) == 0
    assert o3.L.orc_set_sigma_n(o3.h, 5.0) != 0  # sigma_n > sigma0 * 2^(-1/K)
