"""CPU (not gpu): the Z-slab multi-GPU orchestration (tests/sharded_py.py, the Python restatement of the C slab driver) with gloo,
world_size 2 and 3, on the oracle compute backend (tests/cpu_backend.py).

What is under test is the sharding logic itself: slab bounds and their 2^k alignment, the
per-blur z-halo exchange, the max all-reduces, the window halos, the replicated coarse
octaves, the all-gather-v and its global (o, s, z, y, x) order incl. the stale-strength
quirk.  The result must equal the single-process oracle bit-for-bit.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dims, outdir, cuboid=False, sigma0=1.6, units=(1.0, 1.0, 1.0)):
    sys.path.insert(0, ROOT)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch
    import torch.distributed as dist
    from oracle import sift3d_oracle as so
    from tests import sharded_py as sharded
    from tests.cpu_backend import OracleBackend

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    try:
        nx, ny, nz = dims
        vol = so.synth_survey(dims, nblob=int(200 * nx * ny * nz / 64 ** 3))
        job = sharded.ShardedSift3D(nx, ny, nz, dist.group.WORLD, backend=OracleBackend(),
                                    cuboid_extrema=cuboid, sigma0=sigma0, units=units)
        z0, z1 = job.in_own
        job.set_local_volume(vol[z0:z1])
        kp = job.detect()
        idx, hist = job.describe()
        mat = job.gather_descriptors()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), kp=kp, idx=idx, mat=mat,
                 ncand=job.ncand, o_shard=job.g.o_shard, bounds=np.array(job.g.b0),
                 num_octaves=job.g.num_octaves, halo=job.halo)
    finally:
        dist.destroy_process_group()


def _reference(dims, cuboid=False):
    from oracle import sift3d_oracle as so
    nx, ny, nz = dims
    vol = so.synth_survey(dims, nblob=int(200 * nx * ny * nz / 64 ** 3))
    o = so.Oracle(cuboid_extrema=cuboid)
    assert o.detect(vol) == 0 and o.describe() == 0
    return o


@pytest.mark.parametrize("world,dims,cuboid", [(2, (40, 44, 200), False), (3, (36, 40, 152), False),
                                               (2, (24, 24, 40), False), (2, (40, 44, 200), True)])
def test_sharded_equals_single(world, dims, cuboid):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), dims, d, cuboid), nprocs=world, join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
    o = _reference(dims, cuboid)
    ok = o.keypoints()
    assert len(ok) > 5
    want_mat = o.desc_mat()
    covered = np.zeros(len(ok), int)
    for r, g in enumerate(res):
        kp = g["kp"]
        assert int(g["ncand"]) == len(o.candidates())
        assert len(kp) == len(ok)
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength"):
            np.testing.assert_array_equal(kp[f], ok[f], err_msg="rank %d field %s" % (r, f))
        np.testing.assert_array_equal(kp["R"], ok["R"])
        np.testing.assert_array_equal(g["mat"], want_mat)
        covered[g["idx"]] += 1
    # every keypoint is described by exactly one rank
    np.testing.assert_array_equal(covered, 1)
    if dims[2] >= 150:
        assert int(res[0]["o_shard"]) >= 1           # really sharded, not just replicated
        b = res[0]["bounds"]
        assert all(v % (1 << int(res[0]["o_shard"])) == 0 for v in b[:-1])
    else:
        assert int(res[0]["o_shard"]) == 0           # tiny volume: replicated, work split only


def test_config4_geometry_eight_ranks_cpu():
    """BASELINE configs[3]'s slab geometry with EIGHT ranks: 32 x 32 x 1024 gives every rank the 128
    planes (octave 0) and 64 planes (octave 1) it has at 1024^3, o_shard = 2 and the sharded ->
    replicated transition at octave 2 -- on the CPU backend with gloo."""
    import torch.multiprocessing as mp
    world, dims = 8, (32, 32, 1024)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), dims, d, False), nprocs=world, join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
    o = _reference(dims)
    ok, want_mat = o.keypoints(), o.desc_mat()
    assert len(ok) > 50
    covered = np.zeros(len(ok), int)
    for r, g in enumerate(res):
        assert int(g["o_shard"]) == 2 and int(g["num_octaves"]) == 3
        assert list(g["bounds"]) == [128 * i for i in range(9)]
        assert int(g["ncand"]) == len(o.candidates())
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            np.testing.assert_array_equal(g["kp"][f], ok[f], err_msg="rank %d field %s" % (r, f))
        np.testing.assert_array_equal(g["mat"], want_mat)
        covered[g["idx"]] += 1
    np.testing.assert_array_equal(covered, 1)


def test_sharded_wide_windows_equal_single():
    """sigma0 = 2.0 and a 0.7 voxel spacing along z: the descriptor window reaches
    ceil(14.1422 * 2.0 * 2^(2/3) / 0.7) + 2 = 67 planes, far beyond the default 40-plane halo.
    The driver must size its halos (and its minimum slab) from sigma0 / units, and the result
    must still equal the single-process run bit for bit."""
    import torch.multiprocessing as mp
    from oracle import sift3d_oracle as so
    dims, world, sigma0, units = (28, 30, 320), 2, 2.0, (1.0, 1.2, 0.7)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), dims, d, False, sigma0, units), nprocs=world,
                 join=True)
        res = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(world)]
    nx, ny, nz = dims
    vol = so.synth_survey(dims, nblob=int(200 * nx * ny * nz / 64 ** 3))
    o = so.Oracle(sigma0=sigma0)
    assert o.detect(vol, units) == 0 and o.describe() == 0
    ok, want_mat = o.keypoints(), o.desc_mat()
    assert len(ok) > 5
    for g in res:
        assert int(g["halo"]) == 67 and int(g["o_shard"]) >= 1
        assert int(g["ncand"]) == len(o.candidates())
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            np.testing.assert_array_equal(g["kp"][f], ok[f], err_msg=f)
        np.testing.assert_array_equal(g["mat"], want_mat)


def test_sharded_refuses_what_does_not_fit():
    from tests.sharded_py import ShardedSift3D
    from tests.cpu_backend import OracleBackend
    with pytest.raises(ValueError):
        ShardedSift3D(32, 32, 64, None, backend=OracleBackend(), units=(1.0, 1.0, 0.0))
    with pytest.raises(ValueError):                       # a 1400-plane window
        ShardedSift3D(32, 32, 64, None, backend=OracleBackend(), sigma0=40.0)


def test_geometry_alignment():
    from tests.sharded_py import Geometry, MIN_SLAB
    g = Geometry(512, 512, 4096, 8)
    assert g.num_octaves == 7 and g.o_shard == 4
    assert g.b0 == [512 * r for r in range(9)]
    for o in range(g.o_shard):
        for r in range(8):
            z0, z1 = g.own(o, r)
            assert z0 % 2 == 0 and z1 - z0 >= MIN_SLAB
    g = Geometry(1024, 1024, 1024, 8)                 # BASELINE configs[3]
    assert g.num_octaves == 8 and g.o_shard == 2 and g.own(0, 3) == (384, 512)
    g = Geometry(100, 90, 333, 4)
    assert sum(g.own(0, r)[1] - g.own(0, r)[0] for r in range(4)) == 333
    g1 = Geometry(64, 64, 64, 1)
    assert g1.o_shard == 0 and g1.own(0, 0) == (0, 64)
