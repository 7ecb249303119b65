"""GPU (-m gpu): the Z-slab driver on the HIP backend.  The GPU box has ONE device, so two
ranks share it and talk through gloo (the driver stages tensors through the host for gloo);
the kernels, slab geometry and exchange logic are exactly those of the RCCL run.  The
result must equal the single-GPU drop-in API bit-for-bit."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dims, outdir, cuboid=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from sift3d_amd import api
    from tests import sharded_py as sharded

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        nx, ny, nz = dims
        vol = api.synth_lattice(dims, seed=5)
        job = sharded.ShardedSift3D(nx, ny, nz, dist.group.WORLD, cuboid_extrema=cuboid)
        z0, z1 = job.in_own
        job.set_local_volume(torch.from_numpy(vol[z0:z1]).cuda())
        kp = job.detect()
        idx, hist = job.describe()
        mat = job.gather_descriptors()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), kp=kp, idx=idx, mat=mat,
                 ncand=job.ncand, o_shard=job.g.o_shard)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,cuboid", [(2, (64, 72, 256), False), (3, (48, 48, 160), False),
                                               (2, (48, 48, 160), True)])
def test_sharded_hip_equals_single_gpu(world, dims, cuboid):
    import torch
    import torch.multiprocessing as mp
    from sift3d_amd import api
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), dims, d, cuboid), nprocs=world, join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
    vol = api.synth_lattice(dims, seed=5)
    det, kp, desc = api.Detector(cuboid_extrema=cuboid), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    k = kp.records()
    m = desc.to_mat_rm()
    assert len(k) > 20
    covered = np.zeros(len(k), int)
    for g in res:
        assert int(g["ncand"]) == det.num_candidates()
        assert int(g["o_shard"]) >= 1
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            np.testing.assert_array_equal(g["kp"][f], k[f], err_msg=f)
        np.testing.assert_array_equal(g["mat"], m)
        covered[g["idx"]] += 1
    np.testing.assert_array_equal(covered, 1)


def _worker_c(rank, world, port, dims, outdir, det_kw=None):
    """The C slab driver (sift3d_amd_sharded_*) with host-staged gloo exchanges."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from sift3d_amd import api, sharded_c

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        nx, ny, nz = dims
        vol = api.synth_lattice(dims, seed=5)
        det = api.Detector(**det_kw) if det_kw else None
        job = sharded_c.CShardedSift3D(nx, ny, nz, sharded_c.DistTransport(), detector=det)
        z0, z1 = job.in_own
        job.set_local_volume(vol[z0:z1])
        job.detect()
        idx, desc = job.describe()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), kp=job.keypoints(), idx=idx,
                 mat=desc.to_mat_rm() if len(idx) else np.zeros((0, 771), np.float32),
                 ncand=job.ncand, o_shard=job.o_shard)
        job.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,det_kw", [
    (2, (64, 72, 256), None), (3, (48, 48, 320), None), (2, (32, 40, 48), None),
    # what the reference API accepts and the default sweep does not cover: rows that are not whole
    # quads (130 -> 65 -> 32 -> 16: imutil.c:1545-1547), the cuboid neighbourhood (sift.c:24,
    # 761-796), another level count (sift.c:527-533) -- stored DoG levels + sift3d_hip_extrema_mode
    (2, (130, 126, 244), None), (2, (64, 72, 256), dict(cuboid_extrema=True)),
    (2, (48, 52, 240), dict(num_kp_levels=2, sigma_n=1.0)),
    (3, (50, 46, 330), dict(num_kp_levels=4, cuboid_extrema=True, peak_thresh=0.05))])
def test_c_slab_driver_equals_single_gpu(world, dims, det_kw):
    """sift3d_amd_sharded_detect / _describe (host orchestration in C, exchanges through the
    transport vtable) == the single-GPU drop-in API, bit for bit: sharded octaves, the
    sharded -> replicated transition, a volume too small to shard at all, and the
    configurations that take stored DoG levels."""
    import torch
    import torch.multiprocessing as mp
    from sift3d_amd import api
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_c, args=(world, _free_port(), dims, d, det_kw), nprocs=world, join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
    vol = api.synth_lattice(dims, seed=5)
    det, kp, desc = api.Detector(**(det_kw or {})), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    k = kp.records()
    m = desc.to_mat_rm()
    assert len(k) > 5
    covered = np.zeros(len(k), int)
    for g in res:
        assert int(g["ncand"]) == det.num_candidates()
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            np.testing.assert_array_equal(g["kp"][f], k[f], err_msg=f)
        np.testing.assert_array_equal(g["mat"], m[g["idx"]])
        covered[g["idx"]] += 1
    np.testing.assert_array_equal(covered, 1)
    assert int(res[0]["o_shard"]) >= (1 if dims[2] >= 200 else 0)


def _worker_1024(rank, world, port, n, outdir):
    """BASELINE configs[3] geometry: one 1024^3 volume as `world` Z-slabs, driven by the C slab
    driver (the product path for N > 1).  The slab is generated on the device (order-independent
    generator), results leave the process as digests."""
    sys.path.insert(0, ROOT)
    import hashlib
    import torch
    import torch.distributed as dist
    from sift3d_amd import sharded_c

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        job = sharded_c.CShardedSift3D(n, n, n, sharded_c.DistTransport())
        job.synth(seed=11)
        job.detect()
        idx, desc = job.describe()
        kp = job.keypoints()
        hist = desc.to_mat_rm()[:, 3:]
        h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), idx=idx, ncand=job.ncand,
                 nkp=len(kp), o_shard=job.o_shard, bounds=np.array(job.in_own),
                 kp_digest=np.array([h(kp[f]) for f in ("o", "s", "xd", "yd", "zd", "sd",
                                                         "strength", "R")]),
                 desc_digest=h(hist))
        job.close()
    finally:
        dist.destroy_process_group()


def test_config4_1024_sharded_equals_single_gpu():
    """BASELINE configs[3] at FULL size: a 1024^3 float32 volume as two Z-slabs (two ranks on the
    one device of the GPU box, the C slab driver with its exchanges staged through gloo) must
    equal the single-GPU drop-in C API bit for bit --
    candidate count, every keypoint field, every descriptor -- and reproduce the counts of the
    single-GPU 1024^3 run recorded in DESIGN.md (1 249 357 candidates -> 332 413 keypoints)."""
    import hashlib
    import torch
    import torch.multiprocessing as mp
    from sift3d_amd import api, hip
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    n, world = 1024, 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_1024, args=(world, _free_port(), n, d), nprocs=world, join=True)
        res = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(world)]
    vol = torch.empty((n, n, n), device="cuda")
    hip.synth_lattice(vol, 0, 11)
    torch.cuda.synchronize()
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    k = kp.records()
    assert det.num_candidates() == 1249357 and len(k) == 332413
    m = desc.to_mat_rm()[:, 3:]
    h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    want = [h(k[f]) for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R")]
    covered = np.zeros(len(k), int)
    for g in res:
        assert int(g["ncand"]) == det.num_candidates() and int(g["nkp"]) == len(k)
        assert int(g["o_shard"]) >= 3                   # 512, 256, 128 planes per rank are slabs
        assert list(g["bounds"]) in ([0, 512], [512, 1024])
        assert list(g["kp_digest"]) == want
        idx = g["idx"]
        assert str(g["desc_digest"]) == h(m[idx])
        covered[idx] += 1
    np.testing.assert_array_equal(covered, 1)
    del det, vol
    torch.cuda.empty_cache()


def _run_thread_ranks(world, dims, transport, det_kw=None, seed=5, synth_on_device=False):
    """`world` C slab drivers as threads of THIS process on the one device; returns one dict per rank.
    transport: "stream" = the library's stream-ordered thread transport (event-ordered device copies, the
    completion semantics of ncclSend/ncclRecv, no host-side stream sync), "host" = ThreadTransport
    (drains the stream around every exchange)."""
    import threading
    from sift3d_amd import api, sharded_c
    nx, ny, nz = dims
    group = sharded_c.StreamThreadGroup(world) if transport == "stream" else sharded_c.ThreadGroup(world)
    make = sharded_c.StreamThreadTransport if transport == "stream" else sharded_c.ThreadTransport
    vol = None if synth_on_device else api.synth_lattice(dims, seed=seed)
    out, err = [None] * world, []

    def run(rank):
        try:
            det = api.Detector(**det_kw) if det_kw else None
            tr = make(group, rank)
            job = sharded_c.CShardedSift3D(nx, ny, nz, tr, detector=det)
            if synth_on_device:
                job.synth(seed=seed)
            else:
                z0, z1 = job.in_own
                job.set_local_volume(vol[z0:z1])
            job.detect()
            idx, desc = job.describe()
            out[rank] = dict(kp=job.keypoints(), idx=idx.copy(), own=job.in_own, o_shard=job.o_shard,
                             num_octaves=job.num_octaves, ncand=job.ncand,
                             mat=desc.to_mat_rm() if len(idx) else np.zeros((0, 771), np.float32))
            job.close()
            tr.close()
        except Exception as e:  # noqa: BLE001
            err.append((rank, repr(e)))
            try:
                group.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    if transport == "stream":
        group.close()
    assert not err, err
    assert all(o is not None for o in out)
    return out


@pytest.mark.parametrize("world,dims,det_kw", [
    (2, (64, 72, 256), None), (3, (48, 48, 320), None), (4, (64, 64, 512), None),
    (2, (130, 126, 244), None), (2, (64, 72, 256), dict(cuboid_extrema=True)),
    (3, (50, 46, 330), dict(num_kp_levels=4, cuboid_extrema=True, peak_thresh=0.05))])
def test_c_slab_driver_over_stream_ordered_transport(world, dims, det_kw):
    """The configurations of test_c_slab_driver_equals_single_gpu with the exchanges ordered by HIP
    events ALONE (sift3d_thread_transport.c: what ncclSend/ncclRecv guarantee and no more): every halo
    exchange on the communication stream must be fenced against the chain streams that produce and
    consume its planes by the driver's own events -- a missing edge is a bit difference here."""
    import torch
    from sift3d_amd import api
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    res = _run_thread_ranks(world, dims, "stream", det_kw)
    vol = api.synth_lattice(dims, seed=5)
    det, kp, desc = api.Detector(**(det_kw or {})), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    k, m = kp.records(), desc.to_mat_rm()
    assert len(k) > 5
    covered = np.zeros(len(k), int)
    for g in res:
        assert g["ncand"] == det.num_candidates() and g["o_shard"] >= 1
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            np.testing.assert_array_equal(g["kp"][f], k[f], err_msg=f)
        np.testing.assert_array_equal(g["mat"], m[g["idx"]])
        covered[g["idx"]] += 1
    np.testing.assert_array_equal(covered, 1)


def _thread_ranks(world, dims, body, transport="stream", seed=5):
    """Run body(rank, job) on `world` C slab drivers, each a thread of this process; returns body's results."""
    import threading
    from sift3d_amd import api, sharded_c
    nx, ny, nz = dims
    group = sharded_c.StreamThreadGroup(world) if transport == "stream" else sharded_c.ThreadGroup(world)
    make = sharded_c.StreamThreadTransport if transport == "stream" else sharded_c.ThreadTransport
    vol = api.synth_lattice(dims, seed=seed)
    out, err = [None] * world, []

    def run(rank):
        try:
            tr = make(group, rank)
            job = sharded_c.CShardedSift3D(nx, ny, nz, tr)
            z0, z1 = job.in_own
            job.set_local_volume(vol[z0:z1])
            out[rank] = body(rank, job)
            job.close()
            tr.close()
        except Exception as e:  # noqa: BLE001
            err.append((rank, repr(e)))
            try:
                group.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    if transport == "stream":
        group.close()
    assert not err, err
    return out, vol


@pytest.mark.parametrize("root", [-1, 1])
def test_descriptor_gather_equals_single_gpu(root):
    """sift3d_amd_sharded_gather_descriptors (descriptors computed by the owning rank and
    gathered, N x 771 f32): the matrix in the global keypoint order, on every rank or on one root, equals
    the single-GPU drop-in API's bit for bit."""
    import torch
    from sift3d_amd import api
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    world, dims = 3, (64, 72, 288)

    def body(rank, job):
        job.detect()
        job.describe()
        g = job.gather_descriptors(root)
        return None if g is None else g.to_mat_rm()

    out, vol = _thread_ranks(world, dims, body)
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    m = desc.to_mat_rm()
    assert len(m) > 50
    for r in range(world):
        if root < 0 or r == root:
            np.testing.assert_array_equal(out[r], m, err_msg="rank %d" % r)
        else:
            assert out[r] is None


@pytest.mark.parametrize("where", [1, 2, 3, 4])
def test_rank_local_failure_returns_failure_on_every_rank(where):
    """A failure that ONE rank meets between two exchanges (injected: before the pyramid, after the extrema,
    between the two all-gathers of detect, in the descriptor gather) must come back as SIFT3D_FAILURE from
    the same call on EVERY rank -- the failing rank keeps issuing the step's exchanges, its status word
    travels behind the gathered blocks -- and leave the transport in step: the next, clean call succeeds
    everywhere with the right result (the reference: every stage returns -1, immacros.h:27-32)."""
    import torch
    from sift3d_amd import api
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    world, dims, bad = 3, (48, 48, 320), 1

    def body(rank, job):
        failed = None
        if rank == bad:
            job.inject_failure(where)
        try:
            job.detect()
            job.describe()
            job.gather_descriptors(-1)
            failed = False
        except RuntimeError:
            failed = True
        job.inject_failure(0)
        job.detect()
        job.describe()
        g = job.gather_descriptors(-1)
        return failed, job.keypoints(), g.to_mat_rm()

    out, vol = _thread_ranks(world, dims, body)
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    k, m = kp.records(), desc.to_mat_rm()
    for r in range(world):
        failed, kk, mm = out[r]
        assert failed is True, "rank %d did not see rank %d's failure" % (r, bad)
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            np.testing.assert_array_equal(kk[f], k[f], err_msg="rank %d field %s" % (r, f))
        np.testing.assert_array_equal(mm, m)


def _check_against_reference_fixture(g, kp, mat_rows, idx, ncand):
    """A rank's results against the reference's own run of the volume (tests/golden/g5_256x256x1024.npz,
    made by oracle/make_golden.py from the unmodified reference): keypoint fields and R by digest,
    descriptor rows of this rank by their 8 projections (every row) and the sampled rows elementwise."""
    import json
    from tests import util
    dig = json.loads(str(g["digests"]))
    assert ncand == int(g["ncand"]) and len(kp) == int(g["nkp"])
    assert util.digest(np.stack([kp["o"], kp["s"]], 1)) == dig["kp_os"]
    xyzsd = np.stack([kp[f] for f in ("xd", "yd", "zd", "sd")], 1)
    assert util.digest(xyzsd) == dig["kp_xyzsd"]
    st = int(g["stride"])
    np.testing.assert_array_equal(np.stack([kp["o"], kp["s"]], 1)[::st], g["kp_os_s"])
    np.testing.assert_array_equal(xyzsd[::st], g["kp_xyzsd_s"])
    np.testing.assert_array_equal(kp["strength"][::st], g["kp_strength_s"])
    assert util.digest(kp["strength"]) == dig["kp_strength"]
    assert util.digest(np.ascontiguousarray(kp["R"])) == dig["kp_R"]
    if len(idx):
        h = mat_rows[:, 3:]
        util.assert_desc_projection(h, g["desc_proj"][idx], rtol=1e-5)
        samp = np.nonzero(idx % st == 0)[0]
        if len(samp):
            assert util.rel_err(h[samp], g["desc_hist_s"][idx[samp] // st]) <= 1e-5


@pytest.mark.parametrize("transport", ["stream", "host"])
def test_config4_geometry_eight_ranks(transport):
    """BASELINE configs[3]'s real geometry: EIGHT Z-slabs.  256 x 256 x 1024 gives every rank the
    128 planes (octave 0) and 64 planes (octave 1, against a ~40-plane window halo) it has at
    1024^3, o_shard = 2 and the sharded -> replicated transition at octave 2.  The eight ranks are
    eight threads of this process (the pool allows at most 6 processes on the card), each with its
    own C slab driver.  Every rank's result must equal (a) the REFERENCE's own run of this volume
    (tests/golden/g5_256x256x1024.npz) and (b) the single-GPU drop-in API bit for bit -- over the
    stream-ordered transport (exchanges ordered by events only, as over RCCL) and over the
    host-synchronous one."""
    import torch
    from sift3d_amd import api, hip
    from tests import util
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    world, dims = 8, (256, 256, 1024)
    nx, ny, nz = dims
    out = _run_thread_ranks(world, dims, transport, seed=11, synth_on_device=True)
    vol = torch.empty((nz, ny, nx), device="cuda")
    hip.synth_lattice(vol, 0, 11)
    torch.cuda.synchronize()
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints_device(vol.data_ptr(), nx, ny, nz, kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    k, m = kp.records(), desc.to_mat_rm()
    assert len(k) > 5000
    ref = util.load("g5_256x256x1024") if util.have("g5_256x256x1024") else None
    assert ref is not None, "tests/golden/g5_256x256x1024.npz is missing"
    covered = np.zeros(len(k), int)
    for r, g in enumerate(out):
        assert g["own"] == (128 * r, 128 * (r + 1)) and g["o_shard"] == 2 and g["num_octaves"] == 6
        assert g["ncand"] == det.num_candidates()
        _check_against_reference_fixture(ref, g["kp"], g["mat"], g["idx"], g["ncand"])
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            np.testing.assert_array_equal(g["kp"][f], k[f], err_msg="rank %d field %s" % (r, f))
        np.testing.assert_array_equal(g["mat"], m[g["idx"]])
        covered[g["idx"]] += 1
    np.testing.assert_array_equal(covered, 1)
    del det, vol
    torch.cuda.empty_cache()


def _worker_fixture(rank, world, port, dims, seed, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from sift3d_amd import sharded_c

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        job = sharded_c.CShardedSift3D(dims[0], dims[1], dims[2], sharded_c.DistTransport())
        job.synth(seed=seed)
        job.detect()
        idx, desc = job.describe()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), kp=job.keypoints(), idx=idx,
                 mat=desc.to_mat_rm(), ncand=job.ncand, o_shard=job.o_shard)
        job.close()
    finally:
        dist.destroy_process_group()


def test_two_slabs_against_reference_fixture():
    """Two ranks (processes, exchanges staged through gloo) on the 256 x 256 x 1024 volume, compared with
    the REFERENCE's own run of it (tests/golden/g5_256x256x1024.npz), not with this library's single-GPU
    path: 78 466 candidates -> 21 180 keypoints, keypoint fields and R by digest, every descriptor row
    of a rank by its projections."""
    import torch
    import torch.multiprocessing as mp
    from tests import util
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    ref = util.load("g5_256x256x1024")
    world, dims = 2, (256, 256, 1024)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_fixture, args=(world, _free_port(), dims, 11, d), nprocs=world, join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
    covered = np.zeros(int(ref["nkp"]), int)
    for g in res:
        assert int(g["o_shard"]) >= 3
        _check_against_reference_fixture(ref, g["kp"], g["mat"], g["idx"], int(g["ncand"]))
        covered[g["idx"]] += 1
    np.testing.assert_array_equal(covered, 1)


def _worker_rccl1(rank, world, port, dims, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from sift3d_amd import api, sharded_c

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        nx, ny, nz = dims
        vol = api.synth_lattice(dims, seed=5)
        tr = sharded_c.RcclTransport()
        job = sharded_c.CShardedSift3D(nx, ny, nz, tr)
        job.set_local_volume(vol)
        job.detect()
        idx, desc = job.describe()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), kp=job.keypoints(), idx=idx,
                 mat=desc.to_mat_rm(), ncand=job.ncand)
        job.close()
        tr.close()
    finally:
        dist.destroy_process_group()


def test_rccl_transport_one_rank():
    """The library's own RCCL transport (librccl loaded with dlopen, communicator from a unique
    id, ncclAllReduce / ncclAllGather on the driver's stream) on a communicator of ONE rank --
    all the GPU box can host; the exchanges between ranks themselves are covered by the gloo
    runs above, the RCCL calls by this one.  Result == the single-GPU API."""
    import torch
    import torch.multiprocessing as mp
    from sift3d_amd import api
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    dims = (64, 72, 96)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_rccl1, args=(1, _free_port(), dims, d), nprocs=1, join=True)
        g = np.load(os.path.join(d, "rank0.npz"))
    vol = api.synth_lattice(dims, seed=5)
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.from_array(vol), kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    k = kp.records()
    assert len(k) > 5 and int(g["ncand"]) == det.num_candidates()
    for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
        np.testing.assert_array_equal(g["kp"][f], k[f], err_msg=f)
    np.testing.assert_array_equal(g["idx"], np.arange(len(k)))
    np.testing.assert_array_equal(g["mat"], desc.to_mat_rm())
