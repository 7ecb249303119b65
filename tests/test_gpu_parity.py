"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
  (1) the committed golden vectors of the unmodified reference (tests/golden/), and
  (2) the oracle (oracle/sift3d_oracle.c, itself pinned to those vectors) on seeded inputs.

Bar (BASELINE.json north_star): bit-exact float32 pyramid, candidate counts, keypoint
indices / scales / strengths; orientation and descriptor floats within 1e-5 relative (the
tolerance is written next to each check).  Orientation sums are accumulated in the
reference's order, so R is in practice bit-exact (it feeds threshold decisions).  Descriptor
histograms use every per-voxel value of the reference bit for bit but a different -- fixed,
reproducible -- summation order (two half-wave partial histograms), so they agree to a few
float ulps per element, checked ELEMENTWISE against 1e-5 (util.rel_err has no absolute term).
"""
import json
import os

import numpy as np
import pytest

from tests import util
from tests.test_oracle_golden import check_detect_against_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RTOL = 1e-5  # north_star: "descriptor and orientation floats within 1e-5 relative"


@pytest.fixture(scope="module")
def gpu():
    import torch
    from sift3d_amd import api, hip
    if not torch.cuda.is_available() or not api.device_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    hip.lib()
    return api, hip, torch


# ----------------------------------------------------------------------------------------
# device math
# ----------------------------------------------------------------------------------------
def test_device_expf_matches_host_libm(gpu):
    api, hip, torch = gpu
    rng = np.random.default_rng(1)
    x = np.concatenate([-rng.random(1 << 20).astype(np.float32) * 30.0,
                        -np.exp(rng.random(1 << 18) * 10 - 8).astype(np.float32),
                        np.array([0.0, -0.0, -1e-30, -87.5, -100.0, -1e30], np.float32)])
    want = np.exp(x.astype(np.float32))  # numpy float32 exp == libm expf on this platform?
    # authoritative host value: the library's own host evaluation, which tests/test_host_api.py
    # pins against libm expf bit-for-bit
    host = np.empty_like(x)
    api.lib().sift3d_amd_host_expf(x, host, x.size)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty_like(d_in)
    assert hip.lib().sift3d_hip_test_expf(d_in.data_ptr(), d_out.data_ptr(), x.size,
                                          hip.current_stream()) == 0
    got = d_out.cpu().numpy()
    np.testing.assert_array_equal(got, host)
    assert np.abs(got - want).max() <= 1e-6


def test_device_eigen3_matches_host(gpu):
    api, hip, torch = gpu
    rng = np.random.default_rng(2)
    n = 4096
    m = rng.standard_normal((n, 3, 3))
    A = m @ m.transpose(0, 2, 1)
    Qh = np.zeros((n, 9)); Lh = np.zeros((n, 3))
    for i in range(n):
        api.lib().sift3d_amd_host_eigen3(np.ascontiguousarray(A[i].reshape(9)), Qh[i], Lh[i])
    dA = torch.from_numpy(A.reshape(n, 9).copy()).cuda()
    dQ = torch.empty((n, 9), dtype=torch.float64, device="cuda")
    dL = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    assert hip.lib().sift3d_hip_test_eigen3(dA.data_ptr(), dQ.data_ptr(), dL.data_ptr(), n,
                                            hip.current_stream()) == 0
    np.testing.assert_array_equal(dL.cpu().numpy(), Lh)
    np.testing.assert_array_equal(dQ.cpu().numpy(), Qh)
    w = np.linalg.eigvalsh(A)
    np.testing.assert_allclose(Lh, w, rtol=1e-10, atol=1e-12)


# ----------------------------------------------------------------------------------------
# 1-D FIR stage: every kernel variant vs the reference's golden outputs and the oracle
# ----------------------------------------------------------------------------------------
def _fir_gpu(hip, torch, vol, taps, axis, uf, variant=0, **kw):
    """Runs one pass with the source embedded between NaN guard bands: any read outside the
    volume (even one with weight 0) poisons the output."""
    vol = np.ascontiguousarray(vol, np.float32)
    guard = 4096
    big = torch.full((vol.size + 2 * guard,), float("nan"), device="cuda")
    src = big[guard:guard + vol.size].view(vol.shape)
    src.copy_(torch.from_numpy(vol))
    dst = torch.full(vol.shape, float("nan"), device="cuda")
    hip.fir(src, dst, axis, taps, unit_factor=uf, variant=variant, **kw)
    torch.cuda.synchronize()
    return dst.cpu().numpy()


@pytest.mark.parametrize("variant", [0, 1])
def test_fir_golden(gpu, variant):
    """g2_fir.npz: outputs of the reference's convolve_sep (three widths, units 1/2/4,
    asymmetric taps, a 9-long axis under the 17-tap filter)."""
    api, hip, torch = gpu
    g = util.load("g2_fir")
    vol = g["vol"]
    n = 0
    for key in g.files:
        if not key.startswith("axis"):
            continue
        ax = int(key[4])
        _, w, u = key.split("_")
        taps = g["taps_asym"] if w == "asym" else g["taps_" + w]
        got = _fir_gpu(hip, torch, vol, taps, ax, np.float32(1.0 / float(u[1:])), variant)
        np.testing.assert_array_equal(got, g[key], err_msg=key)
        n += 1
    assert n == 33


@pytest.mark.parametrize("shape", [(40, 36, 128), (33, 30, 516), (16, 24, 64), (19, 21, 23),
                                   (70, 64, 64),
                                   # rows of whole 512-float segments (k_fir_x_u1f: two rows in flight): a last
                                   # wave with fewer than its eight rows, two segments per row
                                   (5, 7, 512), (3, 9, 1024)])
def test_fir_fast_paths_vs_oracle(gpu, oracle_mod, shape):
    """All specialised kernels (x register-window, y/z register-ring, dyadic tables) on
    shapes that exercise vector / scalar paths, partial segments and short axes."""
    api, hip, torch = gpu
    rng = np.random.default_rng(sum(shape))
    vol = rng.standard_normal(shape).astype(np.float32)
    for sigma in (0.5387011637869722, 0.9732939207323564, 1.5450077936447955,
                  2.4525469969308156):
        taps = oracle_mod.gauss_taps(sigma)
        for uf in (1.0, 0.5, 0.25, 0.125):
            for ax in range(3):
                want, r = oracle_mod.fir_axis(vol, taps, ax, uf=np.float32(uf), mode=0)
                assert r == 0
                got = _fir_gpu(hip, torch, vol, taps, ax, np.float32(uf))
                np.testing.assert_array_equal(got, want, err_msg="sigma %g uf %g axis %d"
                                              % (sigma, uf, ax))


def test_fir_wider_than_the_tap_tables_vs_oracle(gpu, oracle_mod):
    """Filters of more than SIFT3D_HIP_MAX_TAPS = 65 taps (sigma0 above ~7: the reference accepts any sigma0,
    sift.c:553-565, half width ceil(3 sigma), imutil.c:1275-1277) take the literal kernel in chunks of 65 taps,
    the running sums carried in dst: bit-identical to the oracle for every axis, unit spacing, dyadic and
    non-dyadic tap spacings (the interior branch's coordinate round trip, imutil.c:811-817), axes SHORTER than
    the filter, and as Z-slabs."""
    api, hip, torch = gpu
    rng = np.random.default_rng(77)
    vol = rng.standard_normal((41, 90, 100)).astype(np.float32)
    ran = 0
    for sigma in (11.0, 12.26, 25.0, 44.0):          # 67, 75, 151 (three chunks), 265 taps
        taps = oracle_mod.gauss_taps(sigma)
        assert len(taps) > 65 and len(taps) == 2 * int(np.ceil(3 * sigma)) + 1
        for uf in (1.0, 0.5, 0.125, 1.0 / 1.5, 1.0 / 0.7):
            for ax in range(3):
                if (len(taps) // 2) * uf > vol.shape[2 - ax] - 2:
                    continue        # a mirrored sample would leave the row: undefined in the reference too
                want, r = oracle_mod.fir_axis(vol, taps, ax, uf=np.float32(uf), mode=0)
                assert r == 0
                got = _fir_gpu(hip, torch, vol, taps, ax, np.float32(uf))
                np.testing.assert_array_equal(got, want, err_msg="sigma %g uf %g axis %d" % (sigma, uf, ax))
                ran += 1
    assert ran >= 36
    # a Z-slab of the volume (planes 10 .. 30 of 41; outputs 16 .. 24: the taps reach 37 / 8 + 1 planes) under
    # the 75-tap filter at spacing 1/8
    taps = oracle_mod.gauss_taps(12.26)
    want, r = oracle_mod.fir_axis(vol, taps, 2, uf=np.float32(0.125), mode=0)
    src = torch.from_numpy(vol[10:30].copy()).cuda()
    dst = torch.full(src.shape, float("nan"), device="cuda")
    hip.fir(src, dst, 2, taps, unit_factor=0.125, n_glob=41, off=10, z_lo=6, z_hi=14)
    np.testing.assert_array_equal(dst.cpu().numpy()[6:14], want[16:24])


@pytest.mark.parametrize("shape", [(1, 1, 512), (1, 3, 1024), (2, 2, 512), (1, 17, 1536)])
def test_fir_x_few_rows(gpu, oracle_mod, shape):
    """The x pass with whole 512-float segments (k_fir_x_u1f) on volumes with fewer rows than one wave takes."""
    api, hip, torch = gpu
    rng = np.random.default_rng(sum(shape))
    vol = rng.standard_normal(shape).astype(np.float32)
    for sigma in (0.5387011637869722, 1.2262734984654078, 2.4525469969308156):
        taps = oracle_mod.gauss_taps(sigma)
        want, r = oracle_mod.fir_axis(vol, taps, 0, uf=np.float32(1.0), mode=0)
        assert r == 0
        got = _fir_gpu(hip, torch, vol, taps, 0, np.float32(1.0))
        np.testing.assert_array_equal(got, want, err_msg="sigma %g" % sigma)


def test_fir_non_dyadic_literal(gpu, oracle_mod):
    """Anisotropic units give non-dyadic unit factors: the literal kernel with the
    reference's coordinate round trip (imutil.c:811-817)."""
    api, hip, torch = gpu
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((17, 20, 24)).astype(np.float32)
    taps = oracle_mod.gauss_taps(1.2262734984654078)
    for units in ((1.0, 1.5, 0.7), (3.0, 0.9, 6.0)):
        for ax in range(3):
            uf = np.float32(1.0 / units[ax])
            want, r = oracle_mod.fir_axis(vol, taps, ax, uf=uf, mode=0)
            got = _fir_gpu(hip, torch, vol, taps, ax, uf)
            np.testing.assert_array_equal(got, want)


def test_fir_slab_z(gpu, oracle_mod):
    """Z-slab form: local buffer + halo, global mirror rules -> identical to unsharded."""
    api, hip, torch = gpu
    rng = np.random.default_rng(9)
    nz = 48
    vol = rng.standard_normal((nz, 20, 32)).astype(np.float32)
    for sigma, uf in ((2.4525469969308156, 1.0), (1.5450077936447955, 0.5), (0.9732939207323564, 1.0)):
        taps = oracle_mod.gauss_taps(sigma)
        whole, _ = oracle_mod.fir_axis(vol, taps, 2, uf=np.float32(uf), mode=0)
        reach = int(np.ceil((len(taps) // 2) * uf)) + 1
        for z0, z1 in ((0, 16), (16, 32), (32, 48)):
            lo, hi = max(0, z0 - reach), min(nz, z1 + reach)
            got = _fir_gpu(hip, torch, vol[lo:hi], taps, 2, np.float32(uf), n_glob=nz, off=lo,
                           z_lo=z0 - lo, z_hi=z1 - lo)
            np.testing.assert_array_equal(got[z0 - lo:z1 - lo], whole[z0:z1])


@pytest.mark.parametrize("shape", [(40, 48, 64), (70, 33, 128), (24, 100, 20), (36, 80, 256), (22, 150, 128),
                                   (30, 128, 64),
                                   # whole 64 x 64 / 64 x 32 tiles (k_fir_yz_dma: rows by LDS-DMA): several tile
                                   # rows incl. the one with the virtual rows, more planes than one request list
                                   (20, 192, 128), (300, 128, 64),
                                   # ... and the fewest planes the fused kernel accepts for 17 taps (2 hw + 2)
                                   (18, 128, 64), (19, 256, 192)])
def test_fir_fused_yz_vs_oracle(gpu, oracle_mod, shape):
    """Fused y+z kernel == FIR_z(FIR_y(.)) of the oracle, whole volume and as Z-slabs (partial
    tiles in x and y, both global z faces, interior slab faces)."""
    api, hip, torch = gpu
    rng = np.random.default_rng(shape[0])
    vol = rng.standard_normal(shape).astype(np.float32)
    nz = shape[0]
    for sigma in (0.5387011637869722, 1.2262734984654078, 2.4525469969308156):
        taps = oracle_mod.gauss_taps(sigma)
        ty, r = oracle_mod.fir_axis(vol, taps, 1, uf=np.float32(1.0), mode=0)
        want, r2 = oracle_mod.fir_axis(ty, taps, 2, uf=np.float32(1.0), mode=0)
        assert r == 0 and r2 == 0
        guard = 4096
        big = torch.full((vol.size + 2 * guard,), float("nan"), device="cuda")
        src = big[guard:guard + vol.size].view(vol.shape)
        src.copy_(torch.from_numpy(vol))
        dst = torch.full(vol.shape, float("nan"), device="cuda")
        assert hip.fir_yz(src, dst, taps)
        np.testing.assert_array_equal(dst.cpu().numpy(), want)
        reach = len(taps) // 2 + 1
        for z0, z1 in ((0, nz // 3), (nz // 3, 2 * nz // 3), (2 * nz // 3, nz)):
            lo, hi = max(0, z0 - reach), min(nz, z1 + reach)
            big = torch.full(((hi - lo) * shape[1] * shape[2] + 2 * guard,), float("nan"), device="cuda")
            s2 = big[guard:guard + (hi - lo) * shape[1] * shape[2]].view((hi - lo,) + shape[1:])
            s2.copy_(torch.from_numpy(vol[lo:hi]))
            d2 = torch.full(s2.shape, float("nan"), device="cuda")
            assert hip.fir_yz(s2, d2, taps, n_glob=nz, off=lo, z_lo=z0 - lo, z_hi=z1 - lo)
            np.testing.assert_array_equal(d2.cpu().numpy()[z0 - lo:z1 - lo], want[z0:z1])
    # not covered: nx % 4 != 0 -> the caller must fall back
    odd = torch.zeros((20, 20, 22), device="cuda")
    assert hip.fir_yz(odd, torch.empty_like(odd), oracle_mod.gauss_taps(1.0)) is False


def test_scale_dog_downsample(gpu, oracle_mod):
    api, hip, torch = gpu
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((18, 21, 37)) * 3).astype(np.float32)
    b = rng.standard_normal((18, 21, 37)).astype(np.float32)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    mx = torch.zeros(1, device="cuda")
    hip.absmax(da, mx)
    assert mx.item() == np.abs(a).max()
    out = torch.empty_like(da)
    hip.scale(da, out, mx)
    np.testing.assert_array_equal(out.cpu().numpy(), a / np.abs(a).max())
    dm = torch.zeros(1, device="cuda")
    hip.subtract_absmax(da, db, out, dm)
    np.testing.assert_array_equal(out.cpu().numpy(), a - b)
    assert dm.item() == np.abs(a - b).max()
    # the one-pass octave kernel == level pairs, incl. a size that is not a multiple of 4
    for shape in ((18, 21, 37), (16, 16, 16)):
        g = [torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).cuda() for _ in range(6)]
        dd = [torch.empty_like(g[0]) for _ in range(5)]
        mm = torch.zeros(5, device="cuda")
        assert hip.dog_stack(g, dd, mm)
        for k in range(5):
            want = g[k].cpu().numpy() - g[k + 1].cpu().numpy()
            np.testing.assert_array_equal(dd[k].cpu().numpy(), want)
            assert mm[k].item() == np.abs(want).max()
    assert not hip.dog_stack(g + g, dd + dd + [dd[0]], torch.zeros(11, device="cuda"))  # 12 levels
    ds = torch.empty((9, 10, 18), device="cuda")
    hip.downsample2(da, ds)
    np.testing.assert_array_equal(ds.cpu().numpy(), oracle_mod.downsample(a))
    z = torch.zeros_like(da)
    mz = torch.zeros(1, device="cuda")
    hip.absmax(z, mz)
    hip.scale(z, out, mz)  # all-zero input: no division (imutil.c:706-707)
    assert float(out.abs().max()) == 0.0


# ----------------------------------------------------------------------------------------
# end to end through the drop-in C API
# ----------------------------------------------------------------------------------------
def _run_api(api, vol, units=(1, 1, 1), params=None, device_input=False):
    det = api.Detector(**(params or {}))
    kp = api.KeypointStore()
    if device_input:
        import torch
        d = torch.from_numpy(np.ascontiguousarray(vol)).cuda()
        nz, ny, nx = vol.shape
        rc = det.detect_keypoints_device(d.data_ptr(), nx, ny, nz, kp, units)
        torch.cuda.synchronize()
    else:
        rc = det.detect_keypoints(api.Image.from_array(vol, units), kp)
    return det, kp, rc


@pytest.mark.parametrize("name", ["g3_64", "g3_70x50x41", "g3_aniso", "g3_params",
                                  "g3_lattice48", "g5_128", "g3_cuboid64", "g3_cuboid_params",
                                  "g3_sigma3", "g3_sigma5", "g3_switch285", "g3_switch_aniso", "g3_sigma8"])
def test_detect_describe_golden(gpu, oracle_mod, name):
    """Against the reference's own outputs: every pyramid level (sha1 digests + small levels
    in full), candidate count, the keypoint list incl. the stale-strength quirk, R,
    descriptors, sort order."""
    api, hip, torch = gpu
    g = util.load(name)
    vol = util.golden_input(g, oracle_mod)
    params = util.golden_params(g)
    det, kp, rc = _run_api(api, vol, tuple(g["units"]), params)
    assert rc == 0
    k = kp.records()
    # candidates are not part of the public API; their count is, and the oracle's list is
    # checked against the same fixture in tests/test_oracle_golden.py
    assert det.num_candidates() == len(g["cand_sd"])
    K = params.get("num_kp_levels", 3)
    num_oct = int(g["num_octaves"])
    dig = json.loads(str(g["digests"]))
    for key, want in dig.items():
        if key == "IM":
            a = det.level(2, 0, 0)
        else:
            o, s = key[2:].split("_")
            a = det.level(0 if key[0] == "G" else 1, int(o[1:]), int(s[1:]))
        assert util.digest(a) == want, "level %s differs from the reference" % key
    with pytest.raises(IndexError):
        det.level(0, num_oct, 0)
    np.testing.assert_array_equal(np.stack([k["o"], k["s"]], 1), g["kp_os"])
    np.testing.assert_array_equal(np.stack([k[f] for f in ("xd", "yd", "zd", "sd")], 1),
                                  g["kp_xyzsd"])
    np.testing.assert_array_equal(k["strength"], g["kp_strength"])  # quirk Q2 included
    assert util.rel_err(k["R"], g["kp_R"]) <= RTOL
    np.testing.assert_array_equal(kp.to_mat_rm(), g["kp_mat"])
    desc = api.DescriptorStore()
    assert det.extract_descriptors(kp, desc) == 0
    m = desc.to_mat_rm()
    assert m.shape == (len(k), 771) and m.dtype == np.float32
    np.testing.assert_array_equal(m[:, :3].astype(np.float64), g["desc_xyzsd"][:, :3].astype(np.float32))
    idx = g["desc_idx"]
    assert util.rel_err(m[idx, 3:], g["desc_hist"]) <= RTOL
    np.testing.assert_allclose(m[:, 3:].astype(np.float64).sum(1), g["desc_rowsum"], rtol=1e-5)
    # R: in practice bit-exact (reference-order sums)
    assert np.mean(k["R"] == g["kp_R"]) > 0.999
    # the reference-order descriptor kernel (sift3d_amd_detector_set_exact_descriptors(1)): every histogram
    # the reference's BIT FOR BIT -- the fixtures' sampled rows in full, all rows by their projections
    assert det.set_exact_descriptors(1) == 0
    desc2 = api.DescriptorStore()
    assert det.extract_descriptors(kp, desc2) == 0
    m2 = desc2.to_mat_rm()
    np.testing.assert_array_equal(m2[idx, 3:] + np.float32(0), g["desc_hist"] + np.float32(0))
    assert util.assert_desc_projection(m2[:, 3:], g["desc_proj"], rtol=1e-12) < 1e-12
    if name == "g3_sigma8":
        # sigma0 = 8: Gaussian filters of 31 ... 75 taps (the last one wider than SIFT3D_HIP_MAX_TAPS: the chunked
        # literal kernel) -- every level digest above is the reference's; the windows hold the whole volume
        assert max(len(t) for t in (api.gauss_filter(sg) for sg in
                                    (12.26, 9.73))) > 65
        np.testing.assert_array_equal(m, m2)
    if name in ("g3_sigma3", "g3_sigma5"):
        # windows of 2e5 .. 4e6 voxels: the automatic mode has taken the reference-order kernel for them
        np.testing.assert_array_equal(m, m2)
        # ... because the fast commit (another summation order; here also the weights computed per voxel, not
        # tabulated) drifts with the square root of the terms a bin receives: the difference to the
        # reference's own sequential float sums reaches 1e-5 at sigma0 = 5 (measured 1.03e-5; bar here 3e-5)
        assert det.set_exact_descriptors(-1) == 0
        desc3 = api.DescriptorStore()
        assert det.extract_descriptors(kp, desc3) == 0
        e = util.rel_err(desc3.to_mat_rm()[idx, 3:], g["desc_hist"])
        print("%s: fast commit, max elementwise relative difference to the reference %.3g" % (name, e))
        assert 0.0 < e <= 3e-5
    if name in ("g3_switch285", "g3_switch_aniso"):
        # Fixtures AT the fast / reference-order switch (sift3d_host.c exact_desc_first_level: windows of more
        # than 1.9e5 voxels take the reference-order kernel).  g3_switch285 (sigma0 2.85): level s = 0 has
        # windows of 1.85e5 voxels -- the largest the fast commit is ever used for --, s = 1, 2 switch;
        # g3_switch_aniso (default sigma0, units 1 x 0.8 x 0.8): only s = 2 (2.05e5 voxels) switches.
        first_exact = 1 if name == "g3_switch285" else 2
        assert det.set_exact_descriptors(-1) == 0
        desc3 = api.DescriptorStore()
        assert det.extract_descriptors(kp, desc3) == 0
        m3 = desc3.to_mat_rm()
        slow = k["s"] >= first_exact
        assert slow.any() and (~slow).any()
        # the automatic mode switched exactly where the code says: rows above the switch are the reference-order
        # kernel's (= the reference's, bit for bit), rows below it the fast commit's
        np.testing.assert_array_equal(m[slow], m2[slow])
        np.testing.assert_array_equal(m[~slow], m3[~slow])
        assert not np.array_equal(m3[slow], m2[slow])
        sel = ~slow[idx]
        e = util.rel_err(m3[idx, 3:][sel], g["desc_hist"][sel])
        print("%s: fast commit just below the switch, max elementwise relative difference to the reference %.3g"
              % (name, e))
        assert 0.0 < e <= RTOL
    for lim in (0, 10):
        det2, kp2, rc = _run_api(api, vol, tuple(g["units"]), params)
        kp2.sort_by_strength(lim)
        k2 = kp2.records()
        np.testing.assert_array_equal(np.stack([k2["o"], k2["s"]], 1), g["sort%d_os" % lim])
        np.testing.assert_array_equal(np.stack([k2[f] for f in ("xd", "yd", "zd", "sd")], 1),
                                      g["sort%d_xyzsd" % lim])
        np.testing.assert_array_equal(k2["strength"], g["sort%d_strength" % lim])


@pytest.mark.parametrize("n,gen", [(96, "survey"), (160, "lattice"), ((100, 72, 90), "survey"),
                                   ((130, 126, 122), "lattice"),    # no dimension a multiple of 4
                                   ((320, 72, 64), "lattice")])     # rows of 1.25 256-voxel extrema tiles
def test_detect_describe_vs_oracle(gpu, oracle_mod, n, gen):
    api, hip, torch = gpu
    vol = oracle_mod.synth_survey(n) if gen == "survey" else oracle_mod.synth_lattice(n, seed=3)
    det, kp, rc = _run_api(api, vol, device_input=True)
    assert rc == 0
    o = oracle_mod.Oracle()
    assert o.detect(vol) == 0
    for oc in range(o.num_octaves):
        for s in range(-1, 5):
            np.testing.assert_array_equal(det.level(0, oc, s), o.level(0, oc, s)[0],
                                          err_msg="G %d %d" % (oc, s))
        for s in range(-1, 4):
            np.testing.assert_array_equal(det.level(1, oc, s), o.level(1, oc, s)[0],
                                          err_msg="D %d %d" % (oc, s))
    assert det.num_candidates() == len(o.candidates())
    k, ok = kp.records(), o.keypoints()
    assert len(k) == len(ok) and len(k) > 20
    for f in ("o", "s", "xd", "yd", "zd", "sd", "strength"):
        np.testing.assert_array_equal(k[f], ok[f], err_msg=f)
    assert util.rel_err(k["R"], ok["R"]) <= RTOL
    desc = api.DescriptorStore()
    assert det.extract_descriptors(kp, desc) == 0
    assert o.describe() == 0
    m, om = desc.to_mat_rm(), o.desc_mat()
    assert util.rel_err(m, om) <= RTOL
    # describe after sort+truncate (the CLI's order, cli/kpSift3D.c:122)
    kp.sort_by_strength(25)
    o.sort_by_strength(25)
    assert det.extract_descriptors(kp, desc) == 0 and o.describe() == 0
    assert util.rel_err(desc.to_mat_rm(), o.desc_mat()) <= RTOL


@pytest.mark.parametrize("case", ["coarse_only", "fine_only", "nothing"])
def test_detect_when_a_part_of_the_list_is_empty_vs_oracle(gpu, oracle_mod, case):
    """The detector emits and orients octave 0's candidates while the smaller octaves are still swept, then the
    others' behind them (sift.c:835-868: octave order).  Either part may be empty: a volume whose only structure
    is coarse (no extremum in octave 0), one whose candidates all lie in octave 0, a flat one."""
    api, hip, torch = gpu
    n = 96
    z, y, x = np.meshgrid(*(np.arange(n, dtype=np.float32),) * 3, indexing="ij")
    rng = np.random.default_rng(17)
    vol = np.zeros((n, n, n), np.float32)
    if case == "coarse_only":
        for c in rng.uniform(20, n - 20, size=(6, 3)):
            vol += np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (2 * 7.0 ** 2)).astype(np.float32)
        kw = dict(peak_thresh=0.05)
    elif case == "fine_only":
        # (two octaves, 16^3 and 8^3; every candidate of this noise lies in the first)
        vol = np.random.default_rng(1).random((16, 16, 16), dtype=np.float32)
        kw = dict(peak_thresh=0.3)
    else:
        vol += 1.0
        kw = {}
    det, kp = api.Detector(**kw), api.KeypointStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    o = oracle_mod.Oracle(**kw)
    assert o.detect(vol) == 0
    cand = o.candidates()
    assert det.num_candidates() == len(cand)
    k, ok = kp.records(), o.keypoints()
    assert len(k) == len(ok)
    for f in ("o", "s", "xd", "yd", "zd", "sd", "strength"):
        np.testing.assert_array_equal(k[f], ok[f], err_msg=f)
    if len(k):
        assert util.rel_err(k["R"], ok["R"]) <= RTOL
    octs = set(int(v) for v in cand["o"])
    if case == "coarse_only":
        assert len(ok) > 0 and 0 not in octs
    elif case == "fine_only":
        assert len(ok) > 0 and octs == {0} and o.num_octaves == 2
    else:
        assert len(cand) == 0 and len(ok) == 0


def test_g5_256_golden(gpu, oracle_mod):
    """BASELINE configs[1] (256^3, detect+describe, 1x MI355X) against the reference."""
    if not util.have("g5_256"):
        pytest.skip("g5_256 fixture not generated")
    api, hip, torch = gpu
    g = util.load("g5_256")
    vol = util.golden_input(g, oracle_mod)
    det, kp, rc = _run_api(api, vol, device_input=True)
    assert rc == 0
    assert det.num_candidates() == len(g["cand_sd"]) == 13082
    k = kp.records()
    assert len(k) == 3481
    np.testing.assert_array_equal(np.stack([k["o"], k["s"]], 1), g["kp_os"])
    np.testing.assert_array_equal(np.stack([k[f] for f in ("xd", "yd", "zd", "sd")], 1),
                                  g["kp_xyzsd"])
    np.testing.assert_array_equal(k["strength"], g["kp_strength"])
    assert util.rel_err(k["R"], g["kp_R"]) <= RTOL
    dig = json.loads(str(g["digests"]))
    for key in ("G_o0_s4", "G_o1_s2", "D_o0_s1", "D_o5_s3", "G_o3_s-1"):
        o, s = key[2:].split("_")
        a = det.level(0 if key[0] == "G" else 1, int(o[1:]), int(s[1:]))
        assert util.digest(a) == dig[key], key
    desc = api.DescriptorStore()
    assert det.extract_descriptors(kp, desc) == 0
    m = desc.to_mat_rm()
    idx = g["desc_idx"]
    assert util.rel_err(m[idx, 3:], g["desc_hist"]) <= RTOL
    np.testing.assert_allclose(m[:, 3:].astype(np.float64).sum(1), g["desc_rowsum"], rtol=1e-5)
    # ALL 3 481 rows, bin placement included (sift.c:1355-1373): 8 random projections per row
    util.assert_desc_projection(m[:, 3:], g["desc_proj"])


def test_reuse_and_errors(gpu, oracle_mod):
    """Detector / store reuse across images of different size, and the failure cases of
    SURVEY.md section 8(b)."""
    api, hip, torch = gpu
    det = api.Detector()
    kp = api.KeypointStore()
    desc = api.DescriptorStore()
    assert det.extract_descriptors(kp, desc) == -1            # no keypoints, no pyramid
    for n in (40, 24, 40):
        vol = oracle_mod.synth_survey(n)
        assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
        o = oracle_mod.Oracle()
        o.detect(vol)
        np.testing.assert_array_equal(kp.to_mat_rm(), o.kp_mat())
    assert det.detect_keypoints(api.Image.from_array(np.zeros((7, 16, 16), np.float32)), kp) == -1
    im2 = api.Image(16, 16, 16, 2)
    assert det.detect_keypoints(im2, kp) == -1                # nc != 1
    assert det.detect_keypoints(api.Image.from_array(np.zeros((16, 16, 16), np.float32)), kp) == 0
    assert len(kp) == 0
    assert det.extract_descriptors(kp, desc) == -1            # zero keypoints (sift.c:1178)
    assert det.set_sigma_n(5.0) == -1                         # sigma_n too large for sigma0
    assert det.set_num_kp_levels(2) == -1                     # 1.6*2^(-1/2) < sigma_n = 1.15
    assert det.set_sigma_n(1.0) == 0
    assert det.set_num_kp_levels(2) == 0                      # reallocates with an image set
    vol = oracle_mod.synth_survey(32)
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    o = oracle_mod.Oracle(sigma_n=1.0, num_kp_levels=2)
    o.detect(vol)
    np.testing.assert_array_equal(kp.to_mat_rm(), o.kp_mat())


def test_descriptors_across_reuse(gpu, oracle_mod):
    """One detector and one pair of stores on alternating volumes: the describe kernel reads the keypoint
    records from the page-locked host list and writes the histograms into the store's page-locked array --
    every call must see THIS call's records (bitwise equal to a fresh detector's result)."""
    api, hip, torch = gpu
    vols = [oracle_mod.synth_lattice(96, seed=s) for s in (3, 4)]
    fresh = []
    for v in vols:
        det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
        assert det.detect_keypoints(api.Image.from_array(v), kp) == 0
        assert det.extract_descriptors(kp, desc) == 0
        fresh.append((kp.to_mat_rm().copy(), desc.to_mat_rm().copy()))
    assert len(fresh[0][1]) > 0 and len(fresh[1][1]) > 0 and fresh[0][1].shape != fresh[1][1].shape or \
        not np.array_equal(fresh[0][1], fresh[1][1])
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    for which in (0, 1, 0, 0, 1):
        assert det.detect_keypoints(api.Image.from_array(vols[which]), kp) == 0
        assert det.extract_descriptors(kp, desc) == 0
        np.testing.assert_array_equal(kp.to_mat_rm(), fresh[which][0])
        np.testing.assert_array_equal(desc.to_mat_rm(), fresh[which][1])


def test_g5_512_golden(gpu, oracle_mod):
    """BASELINE configs[2] -- the bench workload (512^3 lattice volume, seed 11) -- against the
    reference's own results (sha1 digests + strided samples, tests/golden/g5_512.npz)."""
    if not util.have("g5_512"):
        pytest.skip("g5_512 fixture not generated")
    api, hip, torch = gpu
    g = util.load("g5_512")
    n = 512
    vol = torch.empty((n, n, n), device="cuda")
    hip.synth_lattice(vol, 0, 11)
    dig = json.loads(str(g["digests"]))
    assert util.digest(vol.cpu().numpy()) == str(g["input_digest"]), "device generator drifted"
    det, kp = api.Detector(), api.KeypointStore()
    assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0
    torch.cuda.synchronize()
    assert det.num_candidates() == int(g["ncand"])
    k = kp.records()
    assert len(k) == int(g["nkp"])
    assert util.digest(np.stack([k["o"], k["s"]], 1)) == dig["kp_os"]
    assert util.digest(np.stack([k[f] for f in ("xd", "yd", "zd", "sd")], 1)) == dig["kp_xyzsd"]
    assert util.digest(k["strength"]) == dig["kp_strength"]
    idx = g["kp_idx"]
    assert util.rel_err(k["R"][idx], g["kp_R_s"]) <= RTOL
    exact_R = util.digest(k["R"]) == dig["kp_R"]
    for key in ("G_o0_s-1", "G_o0_s4", "D_o0_s2", "G_o1_s3", "D_o2_s0", "G_o6_s4"):
        o, s = key[2:].split("_")
        a = det.level(0 if key[0] == "G" else 1, int(o[1:]), int(s[1:]))
        assert util.digest(a) == dig[key], key
    desc = api.DescriptorStore()
    assert det.extract_descriptors(kp, desc) == 0
    m = desc.to_mat_rm()
    assert util.rel_err(m[idx, 3:], g["desc_hist_s"]) <= RTOL
    np.testing.assert_allclose(m[:, 3:].astype(np.float64).sum(1), g["desc_rowsum"], rtol=1e-5)
    # ALL 42 501 rows, bin placement included (sift.c:1355-1373, 1514-1526)
    worst = util.assert_desc_projection(m[:, 3:], g["desc_proj"])
    print("g5_512: worst projection error over all rows: %.3g of the row norm" % worst)
    # R is accumulated in the reference's order: the digest of all 42 501 matrices must match
    assert exact_R, "R differs from the reference at 512^3"
    print("g5_512: max elementwise relative descriptor error on the sampled rows: %.3g"
          % util.rel_err(m[idx, 3:], g["desc_hist_s"]))
    # the reference-order kernel: the digest of ALL 42 501 x 768 histogram values is the reference's
    assert det.set_exact_descriptors(1) == 0
    assert det.extract_descriptors(kp, desc) == 0
    assert util.digest(desc.to_mat_rm()[:, 3:]) == dig["desc_hist"]
    print("g5_512: reference-order descriptors: %.1f ms (fast commit: see the bench line)"
          % (1e3 * det.timings()["describe"]))


def test_read_image_then_detect(gpu, oracle_mod, tmp_path):
    """kpSift3D's flow (cli/kpSift3D.c:96-146): sift3d_read_image -> detect -> describe, on an
    anisotropic int16 NIFTI file with scl_slope/scl_inter; equals the same volume passed as an
    array with the same spacing, and the oracle."""
    from tests.test_host_api import _write_nii
    api, hip, torch = gpu
    vol = oracle_mod.synth_survey((40, 33, 47))
    raw = np.round(vol * 2000).astype(np.int16)
    units = (1.0, 1.5, 0.7)
    p = tmp_path / "vol.nii.gz"
    _write_nii(p, raw, pixdim=units, slope=0.5, inter=3.0)
    im = api.Image.read(str(p))
    want = (raw.astype(np.float64) * 0.5 + 3.0).astype(np.float32)
    np.testing.assert_array_equal(im.data(), want)
    assert im.units == tuple(np.float32(u).item() for u in units)
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(im, kp) == 0 and det.extract_descriptors(kp, desc) == 0
    o = oracle_mod.Oracle()
    assert o.detect(want, im.units) == 0 and o.describe() == 0
    k, ok = kp.records(), o.keypoints()
    assert len(k) == len(ok) > 5 and det.num_candidates() == len(o.candidates())
    for f in ("o", "s", "xd", "yd", "zd", "sd", "strength"):
        np.testing.assert_array_equal(k[f], ok[f], err_msg=f)
    assert util.rel_err(k["R"], ok["R"]) <= RTOL
    assert util.rel_err(desc.to_mat_rm(), o.desc_mat()) <= RTOL


@pytest.mark.parametrize("case", ["lattice160", "survey96", "aniso", "corner0", "params", "wide"])
def test_orientation_parallel_sums_equal_serial_sums(gpu, oracle_mod, case):
    """sift3d_amd_detector_set_serial_orientation: the default path (parallel double sums, decisions by margin, serial
    re-run of the undecided candidates) must give the keypoint list AND the R bits of the path
    that adds every window in the reference's scan order (sift.c:978-990)."""
    api, hip, torch = gpu
    units, kw = (1.0, 1.0, 1.0), {}
    if case == "lattice160":
        vol = oracle_mod.synth_lattice(160, seed=21)
    elif case == "survey96":
        vol = oracle_mod.synth_survey(96)
    elif case == "aniso":
        vol, units = oracle_mod.synth_survey((72, 60, 81)), (1.0, 1.5, 0.7)
    elif case == "corner0":
        vol, kw = oracle_mod.synth_lattice(80, seed=5), dict(corner_thresh=0.0, peak_thresh=0.02)
    elif case == "wide":
        # sigma0 = 5: orientation spheres of 36-45 voxels radius -- beyond the window tables' capacity, so
        # every candidate of those levels goes through the serial re-run
        vol, kw = oracle_mod.synth_survey(72), dict(sigma0=5.0, peak_thresh=0.02, corner_thresh=0.2)
    else:
        vol, kw = oracle_mod.synth_survey(64), dict(num_kp_levels=2, sigma0=2.0, sigma_n=1.0,
                                                    peak_thresh=0.05, corner_thresh=0.3)
    got = {}
    for mode in (1, 0):
        det, kp = api.Detector(**kw), api.KeypointStore()
        assert det.set_serial_orientation(mode) == 0
        assert det.detect_keypoints(api.Image.from_array(vol, units=units), kp) == 0
        got[mode] = (det.num_candidates(), kp.records())
    assert got[0][0] == got[1][0] and len(got[0][1]) == len(got[1][1]) > 0
    for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
        np.testing.assert_array_equal(got[0][1][f], got[1][1][f], err_msg=f)


@pytest.mark.gpu
def test_orientation_in_two_parts_equals_one_call(gpu, oracle_mod):
    """sift3d_hip_orient_tab_part: the detector orients octave 0's candidates while the smaller octaves' extrema
    are still being found -- two parts of one list, disjoint level ranges, two streams, one scratch.  R and the
    keep flags must be those of one call over the whole list, and of the serial sums (sift.c:926-1102)."""
    api, hip, torch = gpu
    rng = np.random.default_rng(11)
    vols = [oracle_mod.synth_survey(72), oracle_mod.synth_lattice(48, seed=3)]
    levels, sd = [], [2.0, 2.5, 3.2]
    for o, v in enumerate(vols):
        for s in range(3):
            t = torch.from_numpy(np.ascontiguousarray(v) * np.float32(1.0 + 0.1 * s)).cuda()
            levels.append(dict(data=t, off=0, nz_glob=t.shape[0], units=(2.0 ** o,) * 3, octave=o,
                               sd=sd[s] * 2.0 ** o))
    d_levels, tab = hip.level_table(levels)
    cands = np.zeros(0, hip.CAND_DTYPE)
    first = []
    for tag, L in enumerate(levels):
        nz, ny, nx = L["data"].shape
        m = 150 if tag < 3 else 40
        # (sorted voxel indices: the list order of the extrema stage; some windows clipped by the faces)
        idx = np.sort(rng.choice(np.arange(nx * ny * nz, dtype=np.uint32), m, replace=False))
        z, y, x = idx // (nx * ny), (idx // nx) % ny, idx % nx
        ok = (x >= 1) & (x <= nx - 2) & (y >= 1) & (y <= ny - 2) & (z >= 1) & (z <= nz - 2)
        c = np.zeros(int(ok.sum()), hip.CAND_DTYPE)
        c["idx"], c["tag"], c["val"] = idx[ok], tag, 1.0
        first.append(len(cands))
        cands = np.concatenate([cands, c])
    n, na = len(cands), first[3]
    R1, k1 = hip.orient_tab(d_levels, len(levels), cands, 0.4)
    R2, k2 = hip.orient_tab(d_levels, len(levels), cands, 0.4, parts=[(0, 3, 0, na), (3, 6, na, n - na)])
    R0, k0 = hip.orient(d_levels, cands, 0.4)
    assert set(np.unique(k1)) <= {0, 1} and 0 < int(k1.sum()) < n
    np.testing.assert_array_equal(k1, k2)
    np.testing.assert_array_equal(k1, k0)
    np.testing.assert_array_equal(R1[k1 == 1], R2[k1 == 1])
    np.testing.assert_array_equal(R1[k1 == 1], R0[k1 == 1])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["lattice160", "noise96", "spike128", "aniso", "retry128", "odd256x93x95"])
def test_dogmax_gathered_by_the_sweep_equals_its_own_pass(gpu, oracle_mod, case):
    """Octave 0's dogmax scan (sift.c:821-826) has no pass of its own by default: the extrema sweep is
    thresholded with LOWER bounds from a sub-lattice, gathers the exact maxima and the reference's threshold
    is applied to the marked voxels afterwards (sift3d_hip_extrema_gauss6_est_phase).  Maxima, candidates
    and keypoints must be those of the two-pass path -- also when the sub-lattice misses the maximum by far
    (spike128: one voxel off the sub-lattice carries it, so the sweep marks nearly every extremum)."""
    api, hip, torch = gpu
    units, kw = (1.0, 1.0, 1.0), {}
    if case == "lattice160":
        vol = oracle_mod.synth_lattice(160, seed=21)
    elif case == "noise96":
        vol = np.random.default_rng(3).standard_normal((96, 96, 96)).astype(np.float32)
    elif case == "spike128":
        vol = np.random.default_rng(4).random((128, 128, 128), dtype=np.float32)
        vol[68, 34, 77] = 400.0            # (z, y, x): two planes and a row off the sub-lattice z = 1 (mod 5), y = 0 (mod 3)
    elif case == "odd256x93x95":
        # rows and planes that are no multiples of the sub-lattice's strides (3 and 5), an odd row count
        vol = np.random.default_rng(6).standard_normal((95, 93, 256)).astype(np.float32)
    elif case == "retry128":
        # more candidates than the first candidate buffer holds: the sweep runs twice (the maxima it gathers
        # must come out the same)
        vol = np.random.default_rng(5).random((128, 128, 128), dtype=np.float32)
        kw = dict(peak_thresh=0.001, corner_thresh=0.9)
    else:
        vol, units = oracle_mod.synth_survey((72, 60, 80)), (1.0, 1.5, 0.7)
    got = {}
    for own_pass in (True, False):
        det, kp = api.Detector(**kw), api.KeypointStore()
        assert det.set_dogmax_pass(own_pass) == 0
        assert det.detect_keypoints(api.Image.from_array(vol, units=units), kp) == 0
        got[own_pass] = (det.num_candidates(), kp.records(), det.dogmax())
    assert got[True][0] == got[False][0] and len(got[True][1]) == len(got[False][1])
    np.testing.assert_array_equal(got[True][2], got[False][2])
    assert (got[True][2] > 0).all()
    for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
        np.testing.assert_array_equal(got[True][1][f], got[False][1][f], err_msg=f)


def test_repeated_runs_are_bitwise_identical(gpu):
    """The window kernels rely on the issue order of a wave's LDS operations and on block-level
    reductions through atomicMax only: every run must give the same bits."""
    import hashlib
    api, hip, torch = gpu
    n = 160
    vol = torch.empty((n, n, n), device="cuda")
    hip.synth_lattice(vol, 0, 21)
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    seen = set()
    for _ in range(6):
        assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0
        assert det.extract_descriptors(kp, desc) == 0
        k = kp.records()
        seen.add((hashlib.sha1(desc.to_mat_rm().tobytes()).hexdigest(),
                  hashlib.sha1(np.ascontiguousarray(k["R"]).tobytes()).hexdigest(),
                  det.num_candidates(), len(k)))
    assert len(seen) == 1 and next(iter(seen))[3] > 100


def test_host_loops_with_a_smaller_openmp_team(gpu):
    """The host loops between the stages (candidate -> keypoint compaction, the launch-order sort of
    describe) cut their lists by the size of the OpenMP team they REALLY got: under OMP_THREAD_LIMIT=3
    (a team smaller than the eight threads asked for, as inside a caller's own parallel region) the
    results must be those of the full team.  OMP_THREAD_LIMIT is read when libgomp starts: subprocess."""
    import hashlib
    import subprocess
    import sys
    api, hip, torch = gpu
    code = (
        "import hashlib, numpy as np, torch\n"
        "from sift3d_amd import api, hip\n"
        "n = 256\n"
        "vol = torch.empty((n, n, n), device='cuda'); hip.synth_lattice(vol, 0, 21)\n"
        "det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()\n"
        "assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0\n"
        "assert det.extract_descriptors(kp, desc) == 0\n"
        "k = kp.records()\n"
        "h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()\n"
        "print('RESULT', det.num_candidates(), len(k), h(k['xd']), h(k['strength']), h(k['R']), "
        "h(desc.to_mat_rm()))\n")
    got = {}
    for limit in ("3", None):
        env = dict(os.environ)
        env.pop("OMP_THREAD_LIMIT", None)
        if limit:
            env["OMP_THREAD_LIMIT"] = limit
        out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        assert line, out.stdout[-2000:]
        got[limit] = line[0].split()[1:]
    assert int(got["3"][0]) > 8192 and int(got["3"][1]) > 4096      # both host loops ran multi-threaded
    assert got["3"] == got[None]


def test_scaled_image_is_not_formed_from_a_borrowed_pointer(gpu, oracle_mod):
    """sift3d_amd_copy_level(which = 2) forms the scaled input image (im_scale, imutil.c:698-713) on
    demand from the volume the detector uploaded ITSELF; after sift3d_amd_detect_keypoints_device the
    volume was the caller's, no pointer to it is kept, and the call fails instead of reading memory the
    caller may have freed."""
    api, hip, torch = gpu
    vol = oracle_mod.synth_survey(48)
    det, kp = api.Detector(), api.KeypointStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    want = det.level(2, 0, 0)
    np.testing.assert_array_equal(want, (vol / np.abs(vol).max()).astype(np.float32))
    t = torch.from_numpy(vol).cuda()
    assert det.detect_keypoints_device(t.data_ptr(), 48, 48, 48, kp) == 0
    del t
    with pytest.raises((RuntimeError, IndexError)):
        det.level(2, 0, 0)
    np.testing.assert_array_equal(det.level(0, 0, -1).shape, (48, 48, 48))   # the pyramid itself stays readable
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    np.testing.assert_array_equal(det.level(2, 0, 0), want)


def test_pyramid_only_entry_and_launch_timings(gpu, oracle_mod):
    """sift3d_amd_build_pyramid_device (bench.py's pyramid-only leg) builds exactly the pyramid of a detect call;
    the per-launch timings of octave 0 (sift3d_amd_timings [10..]: HIP events around every x and fused y+z launch)
    are filled for the blurs that took the fused kernel, and the descriptor kernel's own clock probe reads a
    plausible shader clock."""
    api, hip, torch = gpu
    n = 128                                     # whole 64 x 64 tiles: the fused y+z kernel runs
    vol = torch.empty((n, n, n), device="cuda")
    hip.synth_lattice(vol, 0, 5)
    torch.cuda.synchronize()
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0
    t = det.timings()
    lt = det.launch_timings()
    for b in range(6):                          # six blurs of octave 0, x pass and fused y+z pass each
        assert 0.0 < lt[b][0] < 0.01 and 0.0 < lt[b][1] < 0.01, (b, lt[b])
    assert lt[6] == (0.0, 0.0) and lt[7] == (0.0, 0.0)
    assert t["yz_last"] == lt[5][1]
    assert 0.0 < t["detect_dev"] <= t["detect_wall"] + 1e-4
    assert t["gauss"] > 0 and t["extrema"] > 0 and t["orient"] > 0
    # the default schedule orients octave 0's candidates on their own: both parts end inside the device span
    assert 0.0 < t["orient_oct0_end"] <= t["detect_dev"] + 1e-6
    assert 0.0 < t["orient_rest_end"] <= t["detect_dev"] + 1e-6
    want = [det.level(0, o, s) for o in range(3) for s in range(-1, 5)]
    nk = len(kp)
    assert nk > 50
    # the pyramid alone, on a fresh detector and on the one that has just detected
    for d2 in (api.Detector(), det):
        assert d2.build_pyramid_device(vol.data_ptr(), n, n, n) == 0
        got = [d2.level(0, o, s) for o in range(3) for s in range(-1, 5)]
        for a, b in zip(want, got):
            np.testing.assert_array_equal(a, b)
        assert d2.timings()["gauss"] > 0 and d2.num_candidates() == 0
    # ... and the detector still detects and describes afterwards
    assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0 and len(kp) == nk
    assert det.extract_descriptors(kp, desc) == 0
    clk = det.describe_clock()
    assert clk is not None
    cycles, seconds = clk
    assert 0.5e9 < cycles / seconds < 3.5e9, (cycles, seconds)
    assert 0.0 < seconds <= det.timings()["describe"] + 1e-4     # (the probe wave lives no longer than the stage)
    assert api.Detector().describe_clock() is None          # nothing launched yet: says so, reads nothing
