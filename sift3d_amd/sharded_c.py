"""Bindings of the C Z-slab driver (sift3d_amd/csrc/sift3d_sharded.c, include/sift3d_amd.h).

The orchestration of the multi-GPU run lives in C (the reference's host language); this module
only (a) builds the transport the C driver talks through and (b) exposes the run to bench.py and
the tests:

  * transport "rccl": the library's own RCCL communicator (send/recv, all-reduce, all-gather over
    xGMI).  The 128-byte unique id is made on rank 0 and distributed with torch.distributed --
    the only thing torch is used for here;
  * transport "dist": callbacks that stage the exchanged buffers through the host and move them
    with torch.distributed (gloo).  For rehearsals and tests on a box where the ranks share one
    device; exactly the same C code drives both.
"""
import ctypes as C

import numpy as np

from . import api, hip

KP_DTYPE = np.dtype([("R", "f4", (3, 3)), ("xd", "f8"), ("yd", "f8"), ("zd", "f8"),
                     ("sd", "f8"), ("o", "i4"), ("s", "i4"), ("strength", "f4")])

_HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                    C.c_void_p)
_ARMAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
_AGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("ctx", C.c_void_p), ("halo", _HALO),
                ("allreduce_max", _ARMAX), ("allgather", _AGATHER)]


_bound = False


def _lib():
    global _bound
    L = api.lib()
    if not _bound:
        vp = C.c_void_p
        L.sift3d_amd_rccl_unique_id.argtypes = [vp]
        L.sift3d_amd_rccl_transport.argtypes = [C.POINTER(Transport), C.c_int, C.c_int, vp]
        L.sift3d_amd_rccl_transport_free.argtypes = [C.POINTER(Transport)]
        L.sift3d_amd_rccl_transport_free.restype = None
        L.sift3d_amd_thread_group_create.restype = vp
        L.sift3d_amd_thread_group_create.argtypes = [C.c_int]
        L.sift3d_amd_thread_group_free.argtypes = [vp]
        L.sift3d_amd_thread_group_free.restype = None
        L.sift3d_amd_thread_group_abort.argtypes = [vp]
        L.sift3d_amd_thread_group_abort.restype = None
        L.sift3d_amd_thread_transport.argtypes = [C.POINTER(Transport), vp, C.c_int]
        L.sift3d_amd_thread_transport_free.argtypes = [C.POINTER(Transport)]
        L.sift3d_amd_thread_transport_free.restype = None
        L.sift3d_amd_sharded_create.restype = vp
        L.sift3d_amd_sharded_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Transport), vp,
                                                C.c_double, C.c_double, C.c_double]
        L.sift3d_amd_sharded_free.argtypes = [vp]
        L.sift3d_amd_sharded_free.restype = None
        L.sift3d_amd_sharded_own_planes.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sift3d_amd_sharded_input.restype = vp
        L.sift3d_amd_sharded_input.argtypes = [vp]
        L.sift3d_amd_sharded_synth.argtypes = [vp, C.c_uint64]
        L.sift3d_amd_sharded_detect.argtypes = [vp, vp]
        L.sift3d_amd_sharded_describe.argtypes = [vp, vp, vp, np.ctypeslib.ndpointer(np.int32),
                                                  C.POINTER(C.c_int)]
        L.sift3d_amd_sharded_gather_descriptors.argtypes = [vp, vp, vp, np.ctypeslib.ndpointer(np.int32), C.c_int,
                                                            vp, C.c_int]
        L.sift3d_amd_sharded_inject_failure.argtypes = [vp, C.c_int]
        L.sift3d_amd_sharded_num_candidates.argtypes = [vp]
        L.sift3d_amd_sharded_timings.restype = C.POINTER(C.c_double)
        L.sift3d_amd_sharded_timings.argtypes = [vp]
        L.sift3d_amd_sharded_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]
        _bound = True
    return L


class DistTransport:
    """Host-staged exchanges over a torch.distributed group (gloo): every call first drains the
    stream the C driver enqueued its producers on, so the semantics equal the stream-ordered RCCL
    transport."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.H = hip.lib()
        self._cb = (_HALO(self._halo), _ARMAX(self._armax), _AGATHER(self._agather))  # keep alive
        self.t = Transport(self.rank, self.world, None, *self._cb)

    def _d2h(self, dptr, nbytes, stream):
        a = np.empty(nbytes, np.uint8)
        hip._check(self.H.sift3d_hip_memcpy_d2h(a.ctypes.data, dptr, nbytes, stream), "d2h")
        hip._check(self.H.sift3d_hip_stream_sync(stream), "sync")
        return a

    def _h2d(self, dptr, a, stream):
        hip._check(self.H.sift3d_hip_memcpy_h2d(dptr, a.ctypes.data, a.nbytes, stream), "h2d")
        hip._check(self.H.sift3d_hip_stream_sync(stream), "sync")

    def _halo(self, ctx, send_lo, recv_lo, send_hi, recv_hi, nbytes, stream):
        try:
            torch, dist = self.torch, self.dist
            ops, back = [], []
            for sp, rp, peer in ((send_lo, recv_lo, self.rank - 1), (send_hi, recv_hi, self.rank + 1)):
                if rp:
                    buf = torch.empty(nbytes, dtype=torch.uint8)
                    back.append((rp, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
                if sp:
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(self._d2h(sp, nbytes, stream)),
                                          peer, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for rp, buf in back:
                self._h2d(rp, buf.numpy(), stream)
            return 0
        except Exception as e:  # a Python exception must not unwind through C
            print("DistTransport.halo: %r" % (e,))
            return -1

    def _armax(self, ctx, dbuf, n, stream):
        try:
            a = self._d2h(dbuf, 4 * n, stream).view(np.float32)
            t = self.torch.from_numpy(a.copy())
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
            self._h2d(dbuf, t.numpy(), stream)
            return 0
        except Exception as e:
            print("DistTransport.allreduce_max: %r" % (e,))
            return -1

    def _agather(self, ctx, dsend, drecv, nbytes, stream):
        try:
            mine = self.torch.from_numpy(self._d2h(dsend, nbytes, stream))
            outs = [self.torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(outs, mine, group=self.group)
            self._h2d(drecv, np.concatenate([o.numpy() for o in outs]), stream)
            return 0
        except Exception as e:
            print("DistTransport.allgather: %r" % (e,))
            return -1

    def close(self):
        pass


class ThreadGroup:
    """What the N thread-ranks of ONE process share (ThreadTransport)."""

    def __init__(self, world):
        import queue
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.q = {(a, b): queue.Queue() for a in range(world) for b in (a - 1, a + 1) if 0 <= b < world}
        self.ack = {k: queue.Queue() for k in self.q}


class ThreadTransport:
    """N ranks as N threads of one process on one device: device-to-device copies between the
    slab drivers, rendezvous through queues / a barrier.  Host-synchronous like DistTransport
    (every call first drains the caller's stream).  This is how a one-GPU box runs BASELINE
    configs[3]'s real geometry -- 8 ranks -- without 8 processes on the card (the pool allows 6)."""
    TIMEOUT = 300

    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world
        self.H = hip.lib()
        self._cb = (_HALO(self._halo), _ARMAX(self._armax), _AGATHER(self._agather))  # keep alive
        self.t = Transport(self.rank, self.world, None, *self._cb)

    def _sync(self, stream):
        hip._check(self.H.sift3d_hip_stream_sync(stream), "sync")

    def _halo(self, ctx, send_lo, recv_lo, send_hi, recv_hi, nbytes, stream):
        try:
            g, r = self.g, self.rank
            self._sync(stream)                                   # my planes are final
            for sp, peer in ((send_lo, r - 1), (send_hi, r + 1)):
                if sp:
                    g.q[(r, peer)].put(sp)
            for rp, peer in ((recv_lo, r - 1), (recv_hi, r + 1)):
                if rp:
                    src = g.q[(peer, r)].get(timeout=self.TIMEOUT)
                    hip._check(self.H.sift3d_hip_memcpy_d2d(rp, src, nbytes, stream), "d2d")
            self._sync(stream)
            for rp, peer in ((recv_lo, r - 1), (recv_hi, r + 1)):
                if rp:
                    g.ack[(peer, r)].put(1)                      # the sender's planes have been read
            for sp, peer in ((send_lo, r - 1), (send_hi, r + 1)):
                if sp:
                    g.ack[(r, peer)].get(timeout=self.TIMEOUT)
            return 0
        except Exception as e:  # a Python exception must not unwind through C
            print("ThreadTransport.halo (rank %d): %r" % (self.rank, e))
            return -1

    def _collect(self, mine):
        g = self.g
        g.slots[self.rank] = mine
        g.barrier.wait(self.TIMEOUT)
        allv = [np.array(v, copy=True) for v in g.slots]
        g.barrier.wait(self.TIMEOUT)                             # the slots may be reused
        return allv

    def _armax(self, ctx, dbuf, n, stream):
        try:
            a = np.empty(n, np.float32)
            hip._check(self.H.sift3d_hip_memcpy_d2h(a.ctypes.data, dbuf, 4 * n, stream), "d2h")
            self._sync(stream)
            m = np.maximum.reduce(self._collect(a))
            hip._check(self.H.sift3d_hip_memcpy_h2d(dbuf, m.ctypes.data, 4 * n, stream), "h2d")
            self._sync(stream)
            return 0
        except Exception as e:
            print("ThreadTransport.allreduce_max (rank %d): %r" % (self.rank, e))
            return -1

    def _agather(self, ctx, dsend, drecv, nbytes, stream):
        try:
            a = np.empty(nbytes, np.uint8)
            hip._check(self.H.sift3d_hip_memcpy_d2h(a.ctypes.data, dsend, nbytes, stream), "d2h")
            self._sync(stream)
            out = np.concatenate(self._collect(a))
            hip._check(self.H.sift3d_hip_memcpy_h2d(drecv, out.ctypes.data, out.nbytes, stream), "h2d")
            self._sync(stream)
            return 0
        except Exception as e:
            print("ThreadTransport.allgather (rank %d): %r" % (self.rank, e))
            return -1

    def close(self):
        pass


class StreamThreadGroup:
    """The C side's group of N thread-ranks (sift3d_amd_thread_group): mailboxes and events of the
    STREAM-ORDERED thread transport.  `barrier.abort()` mirrors ThreadGroup's interface for the callers
    that give up on an exception."""

    class _Abort:
        def __init__(self, g):
            self.g = g

        def abort(self):
            if self.g.h:
                _lib().sift3d_amd_thread_group_abort(self.g.h)

    def __init__(self, world):
        self.world = world
        self.h = _lib().sift3d_amd_thread_group_create(world)
        if not self.h:
            raise RuntimeError("sift3d_amd_thread_group_create failed")
        self.barrier = StreamThreadGroup._Abort(self)

    def close(self):
        if getattr(self, "h", None):
            _lib().sift3d_amd_thread_group_free(self.h)
            self.h = None


class StreamThreadTransport:
    """N ranks as N threads of one process, exchanging through the library's stream-ordered thread
    transport (sift3d_thread_transport.c): device-to-device copies ordered by HIP events only -- the
    completion semantics of ncclSend/ncclRecv, NO host-side stream synchronisation.  A missing event edge
    between the slab driver's streams shows here as a wrong result on one GPU."""

    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world
        self.t = Transport()
        if _lib().sift3d_amd_thread_transport(C.byref(self.t), group.h, rank) != 0:
            raise RuntimeError("sift3d_amd_thread_transport failed")

    def close(self):
        if self.t.ctx:
            _lib().sift3d_amd_thread_transport_free(C.byref(self.t))


class RcclTransport:
    """The library's own RCCL communicator; torch.distributed only carries the unique id."""

    def __init__(self, group=None, device="cuda"):
        import torch
        import torch.distributed as dist
        L = _lib()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        idbuf = np.zeros(128, np.uint8)
        if rank == 0 and L.sift3d_amd_rccl_unique_id(idbuf.ctypes.data) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        dev = device if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.from_numpy(idbuf).to(dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        idbuf = t.cpu().numpy()
        self.t = Transport()
        if L.sift3d_amd_rccl_transport(C.byref(self.t), world, rank, idbuf.ctypes.data) != 0:
            raise RuntimeError("sift3d_amd_rccl_transport failed")
        self.rank, self.world = rank, world

    def close(self):
        if self.t.ctx:
            _lib().sift3d_amd_rccl_transport_free(C.byref(self.t))


class _Single:
    def __init__(self):
        self.rank, self.world = 0, 1
        self.t = Transport(0, 1, None, _HALO(), _ARMAX(), _AGATHER())

    def close(self):
        pass


class CShardedSift3D:
    """One nx*ny*nz volume cut into Z-slabs, driven by the C slab driver."""

    def __init__(self, nx, ny, nz, transport=None, detector=None, units=(1.0, 1.0, 1.0)):
        L = _lib()
        self.tr = transport if transport is not None else _Single()
        self.det = detector               # api.Detector carrying thresholds / scales, or None
        self.h = L.sift3d_amd_sharded_create(nx, ny, nz, C.byref(self.tr.t),
                                             detector.h if detector is not None else None,
                                             float(units[0]), float(units[1]), float(units[2]))
        if not self.h:
            raise ValueError("sift3d_amd_sharded_create refused the configuration (see stderr)")
        z0, z1 = C.c_int(), C.c_int()
        L.sift3d_amd_sharded_own_planes(self.h, C.byref(z0), C.byref(z1))
        self.in_own = (z0.value, z1.value)
        self.dims = (nx, ny, nz)
        self.rank, self.world = self.tr.rank, self.tr.world
        self.kp_store, self.desc_store = api.KeypointStore(), api.DescriptorStore()
        self.own_idx = np.zeros(0, np.int32)
        no, osh, halo = C.c_int(), C.c_int(), C.c_int()
        L.sift3d_amd_sharded_info(self.h, C.byref(no), C.byref(osh), C.byref(halo))
        self.num_octaves, self.o_shard, self.halo = no.value, osh.value, halo.value

    def close(self):
        if getattr(self, "h", None):
            _lib().sift3d_amd_sharded_free(self.h)
            self.h = None

    __del__ = close

    def set_local_volume(self, own_planes):
        """own_planes: [z1 - z0, ny, nx] float32 (numpy array or CUDA tensor)."""
        L = _lib()
        z0, z1 = self.in_own
        nx, ny, _ = self.dims
        n = (z1 - z0) * ny * nx
        dst = L.sift3d_amd_sharded_input(self.h)
        H = hip.lib()
        if isinstance(own_planes, np.ndarray):
            a = np.ascontiguousarray(own_planes, np.float32)
            assert a.size == n
            hip._check(H.sift3d_hip_memcpy_h2d(dst, a.ctypes.data, 4 * n, None), "h2d")
        else:
            t = own_planes.contiguous()
            assert t.numel() == n
            hip._check(H.sift3d_hip_memcpy_d2d(dst, t.data_ptr(), 4 * n, None), "d2d")
        hip._check(H.sift3d_hip_stream_sync(None), "sync")

    def synth(self, seed=11):
        if _lib().sift3d_amd_sharded_synth(self.h, seed) != 0:
            raise RuntimeError("sift3d_amd_sharded_synth failed")

    def detect(self):
        if _lib().sift3d_amd_sharded_detect(self.h, self.kp_store.h) != 0:
            raise RuntimeError("sift3d_amd_sharded_detect failed: %s"
                               % hip.lib().sift3d_hip_last_error().decode())
        self.ncand = _lib().sift3d_amd_sharded_num_candidates(self.h)
        return self.kp_store

    def describe(self):
        n = len(self.kp_store)
        idx = np.zeros(max(n, 1), np.int32)
        cnt = C.c_int()
        if _lib().sift3d_amd_sharded_describe(self.h, self.kp_store.h, self.desc_store.h, idx,
                                              C.byref(cnt)) != 0:
            raise RuntimeError("sift3d_amd_sharded_describe failed: %s"
                               % hip.lib().sift3d_hip_last_error().decode())
        self.own_idx = idx[:cnt.value].copy()
        return self.own_idx, self.desc_store

    def gather_descriptors(self, root=-1):
        """The descriptors of ALL keypoints in the global order (N x 771 through DescriptorStore.to_mat_rm):
        on every rank (root < 0) or on rank `root` only (the others return None).  Collective."""
        allstore = api.DescriptorStore()
        idx = self.own_idx if len(self.own_idx) else np.zeros(1, np.int32)
        if _lib().sift3d_amd_sharded_gather_descriptors(self.h, self.kp_store.h, self.desc_store.h, idx,
                                                        len(self.own_idx), allstore.h, int(root)) != 0:
            raise RuntimeError("sift3d_amd_sharded_gather_descriptors failed")
        return allstore if root < 0 or root == self.rank else None

    def inject_failure(self, where):
        """Test hook: the next detect (1, 2, 3) / descriptor gather (4) of this rank fails locally; 0 clears."""
        if _lib().sift3d_amd_sharded_inject_failure(self.h, int(where)) != 0:
            raise ValueError("sift3d_amd_sharded_inject_failure(%r)" % (where,))

    def keypoints(self):
        """Global keypoint list as the Python driver's record array."""
        r = self.kp_store.records()
        kp = np.zeros(len(r), KP_DTYPE)
        for f in ("o", "s", "xd", "yd", "zd", "sd", "strength", "R"):
            kp[f] = r[f]
        return kp

    # ---- bench hooks ------------------------------------------------------------------------
    def step(self):
        self.detect()
        self.describe()

    def stats(self):
        return dict(candidates=int(self.ncand), keypoints=int(len(self.kp_store)),
                    keypoints_rank0=int(len(self.own_idx)), sharded_octaves=int(self.o_shard),
                    octaves=int(self.num_octaves), driver="C (sift3d_amd_sharded_*)")

    def pyramid_seconds(self):
        return float(_lib().sift3d_amd_sharded_timings(self.h)[0])

    def breakdown(self):
        """Seconds of the last step by stage (this rank)."""
        t = _lib().sift3d_amd_sharded_timings(self.h)
        return dict(scale=t[6], pyramid=t[0], dog_extrema=t[3], halo_wait_orient=t[4], gathers=t[5],
                    detect_wall=t[1], describe_wall=t[2])
