"""Seeded random configurations (-m gpu): the drop-in API against the oracle.

Volume shapes (rows that are whole quads and rows that are not -- the two extrema paths --, 30..92 voxels a side),
three kinds of content, thresholds, sigma0, the number of keypoint levels and anisotropic units are drawn at random;
every case must give the oracle's candidates and keypoints exactly (count, octave, level, position, scale,
strength), R and the descriptors within 1e-5 relative (north_star's tolerance), and the reference's refusals
(e.g. sigma_n too large for the units) must be refused the same way.  The fixed cases of test_gpu_parity.py pin
known configurations; this one covers what nobody thought of (600 further cases were run once when it was written:
no mismatch)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def gpu():
    import torch
    from sift3d_amd import api, hip
    if not torch.cuda.is_available() or not api.device_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    hip.lib()
    return api, hip, torch


def _case(rng, om):
    quad = rng.random() < 0.7
    dims = [int(rng.integers(8, 24)) * 4 if quad else int(rng.integers(30, 90)) for _ in range(3)]   # (nx, ny, nz)
    if not quad:
        dims[0] = int(rng.integers(30, 90))
    kw = {}
    if rng.random() < 0.5:
        kw["peak_thresh"] = float(rng.choice([0.02, 0.05, 0.1, 0.2]))
    if rng.random() < 0.4:
        kw["corner_thresh"] = float(rng.choice([0.0, 0.2, 0.4, 0.6]))
    if rng.random() < 0.3:
        kw["sigma0"] = float(rng.choice([1.3, 1.6, 2.0, 2.4]))
    if rng.random() < 0.2:
        kw["num_kp_levels"] = int(rng.choice([2, 3, 4]))
    units = (1.0, 1.0, 1.0)
    if rng.random() < 0.25:
        units = tuple(float(rng.choice([0.7, 1.0, 1.3, 1.5])) for _ in range(3))
    gen = int(rng.integers(0, 3))
    if gen == 0:
        vol = om.synth_survey(tuple(dims), seed=int(rng.integers(1, 1000)))
    elif gen == 1:
        vol = om.synth_lattice(tuple(dims), seed=int(rng.integers(1, 1000)))
    else:
        vol = rng.random((dims[2], dims[1], dims[0]), dtype=np.float32)
    return dims, kw, units, gen, vol


@pytest.mark.parametrize("seed", [123, 7])
def test_random_configurations_vs_oracle(gpu, oracle_mod, seed):
    api, hip, torch = gpu
    rng = np.random.default_rng(seed)
    detected = 0
    for case in range(40):
        dims, kw, units, gen, vol = _case(rng, oracle_mod)
        what = "case %d: dims %s %s units %s content %d" % (case, dims, kw, units, gen)
        det, kp, desc = api.Detector(**kw), api.KeypointStore(), api.DescriptorStore()
        rc = det.detect_keypoints(api.Image.from_array(vol, units=units), kp)
        o = oracle_mod.Oracle(**kw)
        assert rc == o.detect(vol, units), what
        if rc != 0:
            continue                              # (refused by both, e.g. sigma_n too large for these units)
        assert det.num_candidates() == len(o.candidates()), what
        k, ok = kp.records(), o.keypoints()
        assert len(k) == len(ok), what
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength"):
            np.testing.assert_array_equal(k[f], ok[f], err_msg=what + " " + f)
        if len(k):
            detected += 1
            assert util.rel_err(k["R"], ok["R"]) <= RTOL, what
            assert det.extract_descriptors(kp, desc) == 0 and o.describe() == 0, what
            assert util.rel_err(desc.to_mat_rm(), o.desc_mat()) <= RTOL, what
    assert detected >= 30
