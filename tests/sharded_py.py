"""TEST INFRASTRUCTURE: the Z-slab orchestration once more in Python, on a pluggable compute backend.

The product's slab driver is C (sift3d_amd/csrc/sift3d_sharded.c).  This module restates the same
orchestration on top of a small backend interface so that the sharding LOGIC -- slab geometry, halo
exchange, reductions, gather order -- can run with gloo on the CPU oracle (tests/cpu_backend.py) on
machines without a GPU, and on the HIP stage ABI as a cross-check of the C driver.

Z-slab multi-GPU driver: one process per GPU, torch.distributed.

The volume is split along z into one contiguous slab per rank.  Everything that is local
in z runs unchanged on the slab (x / y FIR passes, DoG, down-sampling on 2^k-aligned slab
boundaries); the three exchange steps of the path are

  * z-halo exchange with the two slab neighbours before every z FIR pass
    (R = ceil(hw * unit_factor) + 1 planes of the y-pass output, point-to-point send/recv --
    xGMI is point-to-point, only nearest-neighbour links are used);
  * all-reduce(max) of single floats: max|input| (im_scale, imutil.c:699-713) and
    max|DoG| per level (sift.c:821-826) -- order-independent, hence exact;
  * window halos (H planes of the three keypoint levels per octave) so that orientation
    and descriptor windows (sift.c:926, sift.c:1442) are complete, then an all-gather-v of
    the candidate / keypoint records.  Global order (o, s, z, y, x) is kept by concatenating
    ranks in slab order inside each (o, s), so the result -- including the stale-strength
    quirk of the reference's compaction (sift.c:372-384) -- is identical to one GPU.

Octaves whose slabs would be thinner than the window halo are gathered to every rank and
computed replicated (they hold < 1 % of the voxels); their keypoints are still partitioned by
z so that no rank does another rank's window work.

Mirror / boundary rules apply at the GLOBAL faces only (the stage kernels take the global
length and the slab offset), so interior slab faces reproduce the unsharded arithmetic
bit-for-bit.

The compute backend is the HIP stage ABI (sift3d_amd.hip).  Tests inject a CPU backend built
on the oracle to exercise this orchestration with gloo; the product never does.
"""
import math

import numpy as np

KP_DTYPE = np.dtype([("R", "f4", (3, 3)), ("xd", "f8"), ("yd", "f8"), ("zd", "f8"),
                     ("sd", "f8"), ("o", "i4"), ("s", "i4"), ("strength", "f4")])
# oriented keypoint as exchanged between ranks (global coordinates)
GKP_DTYPE = np.dtype([("o", "i4"), ("s", "i4"), ("x", "i4"), ("y", "i4"), ("z", "i4"),
                      ("R", "f4", (9,))])

# Defaults (sigma0 1.6, K 3, unit spacing): the descriptor window of the last keypoint level
# reaches ceil(2 * 7.0711 * 1.6 * 2^(2/3)) + 2 = 38 planes; a sharded octave keeps at least the
# halo + 8 planes per rank.  ShardedSift3D derives both from its own sigma0 / units / K.
WINDOW_HALO = 40
MIN_SLAB = 48


class Geometry:
    """Octave / slab bookkeeping (pure Python)."""

    def __init__(self, nx, ny, nz, world, num_kp_levels=3, min_slab=MIN_SLAB):
        self.world = world
        self.min_slab = int(min_slab)
        mn = min(nx, ny, nz)
        last = int(math.log2(mn)) - 3                      # sift.c:442-444
        if last < 0:
            raise ValueError("input image is too small: must have at least 8 voxels in each "
                             "dimension")
        self.num_octaves = last + 1
        self.K = num_kp_levels
        self.ngl, self.ndl = num_kp_levels + 3, num_kp_levels + 2
        self.dims = []
        d = [nx, ny, nz]
        for _ in range(self.num_octaves):
            self.dims.append(tuple(d))
            d = [v // 2 for v in d]                        # imutil.c:1545-1547
        # number of sharded octaves
        self.o_shard = 0
        if world > 1:
            while (self.o_shard < self.num_octaves and
                   (self.dims[self.o_shard][2] // world) >= self.min_slab):
                self.o_shard += 1
        # slab bounds at octave 0: multiples of 2^o_shard, so that every slab boundary is EVEN in
        # every sharded octave and im_downsample_2x (dst z <- src 2z) never needs a neighbour's
        # plane
        align = 1 << self.o_shard
        b0 = [0]
        for r in range(1, world):
            b0.append(int(round(nz * r / world / align)) * align)
        b0.append(nz)
        self.b0 = b0
        self.bounds = []
        for o in range(self.num_octaves):
            nzo = self.dims[o][2]
            if o < self.o_shard:
                b = [min(v >> o, nzo) for v in b0[:-1]] + [nzo]
            else:                                          # replicated: work split only
                b = [nzo * r // world for r in range(world)] + [nzo]
            self.bounds.append(b)

    def sharded(self, o):
        return o < self.o_shard

    def own(self, o, r):
        return self.bounds[o][r], self.bounds[o][r + 1]


class HipBackend:
    """The product backend: torch CUDA tensors + the sift3d_hip_* stage kernels."""
    device = "cuda"

    def __init__(self):
        import torch
        from sift3d_amd import api, hip
        self.torch, self.hip, self.api = torch, hip, api
        hip.lib()
        if not api.device_available():
            raise RuntimeError("sift3d_amd: no HIP device is available; there is no CPU path")
        # icosahedron tables of the descriptor kernel (init_geometry, sift.c:148-259)
        if api.lib().sift3d_amd_init() != 0:
            raise RuntimeError("sift3d_amd_init failed")

    def gauss_filter(self, sigma):
        return self.api.gauss_filter(sigma)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device="cuda")

    def scalar(self, n=1):
        return self.torch.zeros(n, dtype=self.torch.float32, device="cuda")

    def absmax(self, t, out):
        self.hip.absmax(t, out)

    def scale(self, src, dst, mx):
        self.hip.scale(src, dst, mx)

    def fir(self, src, dst, axis, taps, uf, n_glob, off, z_lo, z_hi):
        self.hip.fir(src, dst, axis, taps, unit_factor=uf, n_glob=n_glob, off=off, z_lo=z_lo,
                     z_hi=z_hi)

    def fir_yz(self, src, dst, taps, n_glob, off, z_lo, z_hi):
        return self.hip.fir_yz(src, dst, taps, n_glob=n_glob, off=off, z_lo=z_lo, z_hi=z_hi)

    def subtract_absmax(self, a, b, dst, out):
        self.hip.subtract_absmax(a, b, dst, out)

    def dog_stack(self, gauss, dogs, out):
        """All DoG levels of an octave in one pass; False = not covered (use level pairs)."""
        return self.hip.dog_stack(gauss, dogs, out)

    def downsample2(self, src, dst):
        self.hip.downsample2(src, dst)

    def level_table(self, levels):
        return self.hip.level_table(levels)[0]

    def extrema_orient(self, specs, table, peak, corner, cuboid=False):
        """detect_extrema for every octave (one shared candidate buffer, device-side running
        count, a single host sync) followed by assign_orientations on the device-resident
        candidates; the oriented ones are compacted on the device.  specs: [(levels, nx, ny,
        nz_local)].  Returns host arrays: tag and |DoG| value of EVERY candidate (scan order),
        and for the kept ones their position in that list, voxel index and R."""
        torch, hip = self.torch, self.hip
        L = hip.lib()
        wb = max(L.sift3d_hip_extrema_work_bytes(nx, ny, nz, len(lv)) for lv, nx, ny, nz in specs)
        if getattr(self, "_work", None) is None or self._work.numel() < wb:
            self._work = torch.empty(wb, dtype=torch.uint8, device="cuda")
        if getattr(self, "_cap", 0) == 0:
            self._cap = 1 << 18
            self._count = torch.zeros(1, dtype=torch.int32, device="cuda")
        while True:
            if getattr(self, "_cand", None) is None or self._cand.numel() < self._cap * 12:
                self._cand = torch.empty(self._cap * 12, dtype=torch.uint8, device="cuda")
                self._R = torch.empty((self._cap, 9), dtype=torch.float32, device="cuda")
                self._keep = torch.empty(self._cap, dtype=torch.int32, device="cuda")
                # page-locked mirrors: the read-backs run as plain DMA, one sync
                pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)  # noqa: E731
                self._host = dict(tag=pin(self._cap, torch.int32), val=pin(self._cap, torch.int32),
                                  pos=pin(self._cap, torch.int64), idx=pin(self._cap, torch.int32),
                                  R=pin((self._cap, 9), torch.float32))
            self._count.zero_()
            for lv, nx, ny, nz in specs:
                arr = (hip.ExtremaLevel * len(lv))()
                for i, l in enumerate(lv):
                    arr[i] = hip.ExtremaLevel(l["prev"].data_ptr(), l["cur"].data_ptr(),
                                              l["next"].data_ptr(), l["absmax"].data_ptr(),
                                              l["z_lo"], l["z_hi"], l["tag"])
                hip._check(L.sift3d_hip_extrema_mode(arr, len(lv), nx, ny, nz, float(peak),
                                                     int(bool(cuboid)), self._cand.data_ptr(),
                                                     self._cap, self._count.data_ptr(),
                                                     self._work.data_ptr(), self._work.numel(),
                                                     hip.current_stream()),
                           "sift3d_hip_extrema_mode")
            n = int(self._count.item())
            if n <= self._cap:
                break
            self._cap = n + n // 4 + 1024
        if n == 0:
            return (np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros(0, np.int64),
                    np.zeros(0, np.uint32), np.zeros((0, 9), np.float32))
        hip._check(L.sift3d_hip_orient(table.data_ptr(), self._cand.data_ptr(), n, float(corner),
                                       self._R.data_ptr(), self._keep.data_ptr(),
                                       hip.current_stream()), "sift3d_hip_orient")
        rec = self._cand[:n * 12].view(torch.int32).view(n, 3)     # idx, tag, val (sift3d_hip_cand)
        kept = torch.nonzero(self._keep[:n]).squeeze(1)             # ascending = scan order
        nk = int(kept.numel())
        h = self._host
        h["tag"][:n].copy_(rec[:, 1], non_blocking=True)
        h["val"][:n].copy_(rec[:, 2], non_blocking=True)
        h["pos"][:nk].copy_(kept, non_blocking=True)
        h["idx"][:nk].copy_(rec[kept, 0], non_blocking=True)
        h["R"][:nk].copy_(self._R[kept], non_blocking=True)
        torch.cuda.synchronize()
        return (h["tag"][:n].numpy(), h["val"][:n].numpy().view(np.float32), h["pos"][:nk].numpy(),
                h["idx"][:nk].numpy().view(np.uint32), h["R"][:nk].numpy())

    def describe(self, table, kps):
        torch, hip = self.torch, self.hip
        n = len(kps)
        if n == 0:
            return np.zeros((0, 768), np.float32)
        # histograms are stored straight into page-locked host memory (as the C API does)
        if getattr(self, "_hist_host", None) is None or self._hist_host.shape[0] < n:
            cap = n + n // 8 + 64
            self._hist_host = torch.empty((cap, 768), dtype=torch.float32, pin_memory=True)
            self._hist_dev = hip.lib().sift3d_hip_host_device_ptr(self._hist_host.data_ptr())
            if not self._hist_dev:
                raise RuntimeError("pinned descriptor buffer is not visible to the device: %s"
                                   % hip.lib().sift3d_hip_last_error().decode())
        nb = n * kps.dtype.itemsize
        if getattr(self, "_kp_host", None) is None or self._kp_host.numel() < nb:
            self._kp_host = torch.empty(nb + nb // 8, dtype=torch.uint8, pin_memory=True)
            self._kp_dev = torch.empty(nb + nb // 8, dtype=torch.uint8, device="cuda")
        self._kp_host[:nb].numpy()[:] = np.ascontiguousarray(kps).view(np.uint8).reshape(-1)
        self._kp_dev[:nb].copy_(self._kp_host[:nb], non_blocking=True)
        dk = self._kp_dev
        hip._check(hip.lib().sift3d_hip_describe(table.data_ptr(), dk.data_ptr(), n,
                                                 self._hist_dev, hip.current_stream()),
                   "sift3d_hip_describe")
        torch.cuda.synchronize()
        return self._hist_host[:n].numpy()

    def synth(self, t, z_off, seed):
        self.hip.synth_lattice(t, z_off, seed)

    def sync(self):
        self.torch.cuda.synchronize()

    def event(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def elapsed(self, e0, e1):
        e1.synchronize()
        return 1e-3 * e0.elapsed_time(e1)


class _Level:
    """One pyramid level on this rank: planes [off, off + nloc) of nz_glob."""
    __slots__ = ("t", "off", "z0", "z1", "nz_glob")

    def __init__(self, t, off, z0, z1, nz_glob):
        self.t, self.off, self.z0, self.z1, self.nz_glob = t, off, z0, z1, nz_glob

    def own(self):          # local plane range of the owned planes
        return self.z0 - self.off, self.z1 - self.off


class ShardedSift3D:
    def __init__(self, nx, ny, nz, group=None, backend=None, peak_thresh=0.1, corner_thresh=0.4,
                 num_kp_levels=3, sigma_n=1.15, sigma0=1.6, units=(1.0, 1.0, 1.0),
                 cuboid_extrema=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if group is not None or dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.be = backend if backend is not None else HipBackend()
        self.peak, self.corner = float(peak_thresh), float(corner_thresh)
        self.cuboid = bool(cuboid_extrema)   # CUBOID_EXTREMA (sift.c:24) as a run-time option
        self.sigma_n, self.sigma0 = float(sigma_n), float(sigma0)
        self.units = tuple(float(u) for u in units)
        if min(self.units) <= 0:
            raise ValueError("voxel spacing must be positive")
        K = int(num_kp_levels)
        if self.sigma0 * math.pow(2.0, -1.0 / K) < self.sigma_n:       # imutil.c:1582-1588
            raise ValueError("sigma_n too large for these settings")
        # filter bank (make_gss, imutil.c:1360-1409)
        self.filters = []
        for i in range(K + 3):
            s_cur = self.sigma_n if i == 0 else self.sigma0 * math.pow(2.0, (i - 2) / K)
            s_next = self.sigma0 * math.pow(2.0, (i - 1) / K)
            self.filters.append(self.be.gauss_filter(math.sqrt(s_next * s_next - s_cur * s_cur)))
        # Halo planes a slab needs from its neighbours, in LEVEL planes (the same number in every
        # octave: scale and spacing double together):
        #   window: the descriptor window of Gaussian level s (keypoint level s - 1) reaches
        #           rad = 2 * 7.0711 * sd (sift.c:1453-1454) = 14.1422 * sigma0 * 2^((s-1)/K) / uz
        #           planes, + 1 for the gradient, + 1 for ceil of a fractional centre;
        #   blur:   the z pass reads ceil(hw * unit_factor) + 1 planes (imutil.c:756-757), most at
        #           octave 0 where unit_factor = 1 / uz.
        uz = self.units[2]
        self.win_reach = [int(math.ceil(14.1422 * self.sigma0 * 2.0 ** ((s - 1) / K) / uz)) + 2
                          if 1 <= s <= K else 1 for s in range(K + 3)]
        hw_max = max(len(f) // 2 for f in self.filters)
        blur_reach = int(math.ceil(hw_max * float(np.float32(1.0 / uz)))) + 1
        self.halo = max(max(self.win_reach), blur_reach)
        if self.halo > 500:
            raise ValueError("sigma0 / units give a %d-plane window: beyond what the window kernels "
                             "address (1023 voxels per axis)" % self.halo)
        self.g = Geometry(nx, ny, nz, self.world, num_kp_levels, min_slab=self.halo + 8)
        g = self.g
        self._alloc()
        self.t_pyr = 0.0
        self._table = None
        self.ncand = 0
        self.kp = np.zeros(0, KP_DTYPE)
        self.my_kp_idx = np.zeros(0, np.int64)
        self.my_desc = np.zeros((0, 768), np.float32)

    # ---- geometry helpers -------------------------------------------------------------
    def _scale(self, o, s):
        return self.sigma0 * math.pow(2.0, o + float(s) / self.g.K)   # imutil.c:1578-1579

    def _lunits(self, o):
        return tuple(u * math.ldexp(1.0, o) for u in self.units)

    def _alloc(self):
        g, be, r = self.g, self.be, self.rank
        self.G, self.D, self.tmp = [], [], []
        for o in range(g.num_octaves):
            nx, ny, nz = g.dims[o]
            if g.sharded(o):
                z0, z1 = g.own(o, r)
                off, hi = max(0, z0 - self.halo), min(nz, z1 + self.halo)
            else:
                z0, z1 = g.own(o, r)                       # work split of a replicated level
                off, hi = 0, nz
            mk = lambda: _Level(be.empty((hi - off, ny, nx)), off, z0, z1, nz)  # noqa: E731
            self.G.append([mk() for _ in range(g.ngl)])
            self.D.append([mk() for _ in range(g.ndl)])
            self.tmp.append((mk(), mk()))
        z0, z1 = g.own(0, r) if g.sharded(0) else (0, g.dims[0][2])
        self.in_own = (z0, z1)
        self.raw = be.empty((z1 - z0, g.dims[0][1], g.dims[0][0]))
        # per octave one contiguous array of maxima; dogmax[o][s] are 1-element views of it
        self._dogmax_o = [be.scalar(g.ndl) for _ in range(g.num_octaves)]
        self.dogmax = [[self._dogmax_o[o][s:s + 1] for s in range(g.ndl)]
                       for o in range(g.num_octaves)]
        self.inmax = be.scalar()

    # ---- communication ------------------------------------------------------------------
    def _is_gloo(self):
        return self.world > 1 and self.dist.get_backend(self.group) == "gloo"

    def _p2p(self, sends, recvs):
        """sends / recvs: lists of (tensor_view, peer).  Views are contiguous plane ranges."""
        if self.world == 1 or (not sends and not recvs):
            return
        dist, torch = self.dist, self.torch
        stage = self._is_gloo() and self.be.device == "cuda"
        ops, backcopy = [], []
        for t, peer in recvs:
            buf = torch.empty(t.shape, dtype=t.dtype) if stage else t
            if stage:
                backcopy.append((t, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
        for t, peer in sends:
            buf = t.cpu() if stage else t
            ops.append(dist.P2POp(dist.isend, buf, peer, self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for t, buf in backcopy:
            t.copy_(buf)

    def _halo(self, lv, h):
        """Fill the h halo planes on both sides of the owned range of `lv` from the slab
        neighbours (which own at least h planes)."""
        if self.world == 1 or h <= 0:
            return
        r, t = self.rank, lv.t
        a, b = lv.own()
        if h > self.halo or h > b - a:
            raise ValueError("a %d-plane halo does not fit slabs of %d planes with %d halo planes"
                             % (h, b - a, self.halo))
        sends, recvs = [], []
        if r > 0 and lv.z0 > 0:
            sends.append((t[a:a + h], r - 1))
            recvs.append((t[a - h:a], r - 1))
        if r < self.world - 1 and lv.z1 < lv.nz_glob:
            sends.append((t[b - h:b], r + 1))
            recvs.append((t[b:b + h], r + 1))
        self._p2p(sends, recvs)

    def _allreduce_max(self, scalars):
        if self.world == 1:
            return
        torch, dist = self.torch, self.dist
        v = torch.cat([s.reshape(1) for s in scalars])
        if self._is_gloo() and v.is_cuda:
            c = v.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
            v = c.to(scalars[0].device)
        else:
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
        for i, s in enumerate(scalars):
            s.copy_(v[i:i + 1])

    def _allgather_records(self, recs):
        """all-gather-v of a numpy structured array -> list per rank."""
        if self.world == 1:
            return [recs]
        torch, dist = self.torch, self.dist
        dev = "cpu" if self._is_gloo() else self.be.device
        n = torch.tensor([len(recs)], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(counts, n, group=self.group)
        counts = [int(c.item()) for c in counts]
        mx = max(max(counts), 1)
        item = recs.dtype.itemsize
        buf = np.zeros(mx * item, np.uint8)
        buf[:len(recs) * item] = np.ascontiguousarray(recs).view(np.uint8).reshape(-1)
        mine = torch.from_numpy(buf).to(dev)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        return [o.cpu().numpy()[:c * item].view(recs.dtype).copy() for o, c in zip(outs, counts)]

    def _allgather_fixed(self, arr):
        """all-gather of equally shaped numpy arrays -> stacked [world, ...]."""
        if self.world == 1:
            return arr[None]
        torch, dist = self.torch, self.dist
        dev = "cpu" if self._is_gloo() else self.be.device
        mine = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        return np.stack([o.cpu().numpy() for o in outs])

    def _allgather_padded(self, recs, counts):
        """all-gather-v when every rank already knows all lengths -> list per rank."""
        if self.world == 1:
            return [recs]
        torch, dist = self.torch, self.dist
        dev = "cpu" if self._is_gloo() else self.be.device
        item = recs.dtype.itemsize
        buf = np.zeros(max(max(counts), 1) * item, np.uint8)
        buf[:len(recs) * item] = np.ascontiguousarray(recs).view(np.uint8).reshape(-1)
        mine = torch.from_numpy(buf).to(dev)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        return [o.cpu().numpy()[:c * item].view(recs.dtype) for o, c in zip(outs, counts)]

    def _gather_level(self, lv_src, dst):
        """all-gather the owned planes of a sharded level into the replicated tensor dst."""
        torch, dist = self.torch, self.dist
        g = self.g
        a, b = lv_src.own()
        bounds = [(min(v, dst.shape[0])) for v in self._rep_bounds]
        mx = max(bounds[i + 1] - bounds[i] for i in range(self.world))
        plane = dst.shape[1] * dst.shape[2]
        stage = self._is_gloo() and dst.is_cuda
        dev = "cpu" if stage else dst.device
        mine = torch.zeros((mx, dst.shape[1], dst.shape[2]), dtype=dst.dtype, device=dev)
        own = lv_src.t[a:b]
        mine[:own.shape[0]].copy_(own)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        for r in range(self.world):
            n = bounds[r + 1] - bounds[r]
            if n > 0:
                dst[bounds[r]:bounds[r + 1]].copy_(outs[r][:n])
        del plane, g

    # ---- input ----------------------------------------------------------------------------
    def set_local_volume(self, own_planes):
        """own_planes: [z1 - z0, ny, nx] tensor (or numpy array) with this rank's raw slab."""
        t = own_planes
        if isinstance(t, np.ndarray):
            t = self.torch.from_numpy(np.ascontiguousarray(t, np.float32))
        self.raw.copy_(t)

    def synth(self, seed=11):
        self.be.synth(self.raw, self.in_own[0], seed)

    # ---- pyramid --------------------------------------------------------------------------
    def _blur(self, o, src, dst, f):
        """apply_Sep_FIR_filter (imutil.c:1127-1206) on a slab: x, y on the owned planes, halo
        exchange of the y-pass output, z on the owned planes."""
        be = self.be
        ta, tb = self.tmp[o]
        lu = self._lunits(o)
        a, b = dst.own()
        nx, ny, nz = self.g.dims[o]
        ufs = [np.float32(1.0 / lu[k]) for k in range(3)]   # unit = 1.0, sift.c:675
        hw = len(f) // 2
        reach = int(math.ceil(hw * float(ufs[2]))) + 1
        be.fir(src.t, ta.t, 0, f, ufs[0], nx, 0, a, b)
        if ufs[1] == 1.0 and ufs[2] == 1.0 and hasattr(be, "fir_yz"):
            # fused y+z kernel: it re-derives the y pass on the halo planes itself, so the
            # exchanged halo is the x-pass output
            self._halo(ta, reach)
            if be.fir_yz(ta.t, dst.t, f, nz, dst.off, a, b):
                return
        be.fir(ta.t, tb.t, 1, f, ufs[1], ny, 0, a, b)
        if self.g.sharded(o):
            self._halo(tb, reach)
        be.fir(tb.t, dst.t, 2, f, ufs[2], nz, dst.off, a, b)

    def _pyramid(self):
        g, be, r = self.g, self.be, self.rank
        # set_im_SIFT3D: scale by the global max|v| (sift.c:645-649)
        self.inmax.zero_()
        be.absmax(self.raw, self.inmax)
        self._allreduce_max([self.inmax])
        first = self.G[0][0]
        ta = self.tmp[0][0]
        a, b = first.own()
        if g.sharded(0) or self.world == 1:
            im = _Level(ta.t.new_empty(ta.t.shape), ta.off, ta.z0, ta.z1, ta.nz_glob)
            be.scale(self.raw, im.t[a:b], self.inmax)
        else:
            # octave 0 itself is replicated (tiny volume): every rank holds the whole input
            im = _Level(ta.t.new_empty(ta.t.shape), 0, 0, g.dims[0][2], g.dims[0][2])
            be.scale(self.raw, im.t, self.inmax)
        e0 = be.event()
        for o in range(g.num_octaves):
            if o == 0:
                self._blur_full(0, im, self.G[0][0], self.filters[0])
            for s in range(1, g.ngl):
                self._blur_full(o, self.G[o][s - 1], self.G[o][s], self.filters[s])
            if o != g.num_octaves - 1:
                s_end = g.ngl - 2
                ds = max(s_end - 2, -1)                     # sift.c:696-697
                src, dst = self.G[o][ds + 1], self.G[o + 1][0]
                if g.sharded(o) and not g.sharded(o + 1):
                    # transition to the replicated octaves: down-sample the owned planes into a
                    # staging slab, then all-gather it
                    b0 = [min(v >> (o + 1), g.dims[o + 1][2]) for v in self._b0] + []
                    b0[-1] = g.dims[o + 1][2]
                    self._rep_bounds = b0
                    z0, z1 = b0[r], b0[r + 1]
                    stage = _Level(be.empty((max(z1 - z0, 1), g.dims[o + 1][1], g.dims[o + 1][0])),
                                   z0, z0, z1, g.dims[o + 1][2])
                    if z1 > z0:
                        be.downsample2(src.t[2 * z0 - src.off:], stage.t[:z1 - z0])
                    self._gather_level(stage, dst.t)
                else:
                    z0, z1 = (g.own(o + 1, r) if g.sharded(o + 1) else (0, g.dims[o + 1][2]))
                    if z1 > z0:
                        be.downsample2(src.t[2 * z0 - src.off:], dst.t[z0 - dst.off:z1 - dst.off])
        e1 = be.event()
        self._ev = (e0, e1)

    def _blur_full(self, o, src, dst, f):
        if self.g.sharded(o):
            self._blur(o, src, dst, f)
        else:
            # replicated level: plain unsharded blur of the whole level on every rank
            be = self.be
            ta, tb = self.tmp[o]
            lu = self._lunits(o)
            nx, ny, nz = self.g.dims[o]
            ufs = [np.float32(1.0 / lu[k]) for k in range(3)]
            be.fir(src.t, ta.t, 0, f, ufs[0], nx, 0, 0, nz)
            if (ufs[1] == 1.0 and ufs[2] == 1.0 and hasattr(be, "fir_yz") and
                    be.fir_yz(ta.t, dst.t, f, nz, 0, 0, nz)):
                return
            be.fir(ta.t, tb.t, 1, f, ufs[1], ny, 0, 0, nz)
            be.fir(tb.t, dst.t, 2, f, ufs[2], nz, 0, 0, nz)

    # ---- detect -----------------------------------------------------------------------------
    def detect(self):
        g, be, r = self.g, self.be, self.rank
        self._b0 = g.b0
        self._pyramid()
        # window / DoG halos of the sharded octaves
        for o in range(g.o_shard):
            for s in range(g.ngl):
                # keypoint level s-1 (Gaussian index s): the descriptor window reaches
                # 2 * 7.0711 * sigma0 * 2^((s-1)/K) / uz planes (sift.c:1453-1455) + 1 for the
                # gradient; the other levels only feed the DoG / extrema neighbours
                self._halo(self.G[o][s], self.win_reach[s])
        # build_dog (sift.c:713-732) + dogmax (sift.c:821-826)
        scal = []
        stack = getattr(be, "dog_stack", None)
        for o in range(g.num_octaves):
            self._dogmax_o[o].zero_()
            d0 = self.D[o][0]
            if g.sharded(o):
                a, b = d0.own()
                lo, hi = max(a - 1, 0), min(b + 1, d0.t.shape[0])
            else:
                lo, hi = 0, d0.t.shape[0]
            if not (stack and stack([self.G[o][s].t[lo:hi] for s in range(g.ngl)],
                                    [self.D[o][s].t[lo:hi] for s in range(g.ndl)],
                                    self._dogmax_o[o])):
                for s in range(g.ndl):
                    be.subtract_absmax(self.G[o][s].t[lo:hi], self.G[o][s + 1].t[lo:hi],
                                       self.D[o][s].t[lo:hi], self.dogmax[o][s])
            scal.extend(self.dogmax[o])
        self._allreduce_max(scal)
        # detect_extrema (sift.c:735-871) on the owned planes of every octave, then
        # assign_orientations (sift.c:1109-1167) for the local candidates
        specs = []
        for o in range(g.num_octaves):
            nx, ny, nzo = g.dims[o]
            levels = []
            z0, z1 = g.own(o, r)
            off = self.D[o][0].off
            zl, zh = max(z0, 1) - off, min(z1, nzo - 1) - off
            for s in range(g.K):
                levels.append(dict(prev=self.D[o][s].t, cur=self.D[o][s + 1].t,
                                   next=self.D[o][s + 2].t, absmax=self.dogmax[o][s + 1],
                                   z_lo=zl, z_hi=max(zh, zl), tag=o * g.ngl + s + 1))
            specs.append((levels, nx, ny, self.D[o][0].t.shape[0]))
        if self._table is None:
            table_levels = []
            for o in range(g.num_octaves):
                for s in range(g.ngl):
                    lv = self.G[o][s]
                    table_levels.append(dict(data=lv.t, off=lv.off, nz_glob=lv.nz_glob,
                                             units=self._lunits(o), octave=o,
                                             sd=self._scale(o, s - 1)))
            self._table = be.level_table(table_levels)
        tag, val, kpos, kidx, R = be.extrema_orient(specs, self._table, self.peak, self.corner,
                                                    self.cuboid)
        # Exchange (all-gather-v): per-(o,s) counts, the candidates' |DoG| values and
        # the ORIENTED keypoints only -- the rejected candidates' records stay on their rank.
        nkey = g.num_octaves * g.K
        tag = tag.astype(np.int64)
        key = (tag // g.ngl) * g.K + (tag % g.ngl - 1)   # non-decreasing: extrema emit in (o, s) order
        ktag = tag[kpos]
        ok_ = ktag // g.ngl
        cnt = np.stack([np.bincount(key, minlength=nkey), np.bincount(key[kpos], minlength=nkey)], 1)
        dims = np.array(g.dims, np.int64)
        offs = np.array([self.D[o][0].off for o in range(g.num_octaves)], np.int64)
        idx = kidx.astype(np.int64)
        nxv, nyv = dims[ok_, 0], dims[ok_, 1]
        rec = np.zeros(len(kpos), GKP_DTYPE)
        rec["o"] = ok_
        rec["s"] = ktag % g.ngl - 1
        rec["x"] = idx % nxv
        rec["y"] = (idx // nxv) % nyv
        rec["z"] = idx // (nxv * nyv) + offs[ok_]
        rec["R"] = R
        cnts = self._allgather_fixed(cnt.astype(np.int64))          # [world, nkey, 2]
        vals = self._allgather_padded(np.ascontiguousarray(val), cnts[:, :, 0].sum(1))
        recs = self._allgather_padded(rec, cnts[:, :, 1].sum(1))
        # global order: (o, s) major, then ranks in slab order (their z ranges are disjoint and
        # ascending), each rank's list already in (z, y, x) order
        if self.world > 1:
            cc = np.concatenate([np.zeros((self.world, 1), np.int64), np.cumsum(cnts[:, :, 0], 1)], 1)
            ck = np.concatenate([np.zeros((self.world, 1), np.int64), np.cumsum(cnts[:, :, 1], 1)], 1)
            kept = np.concatenate([recs[r][ck[r, k]:ck[r, k + 1]]
                                   for k in range(nkey) for r in range(self.world)])
            # copy_Keypoint omits `strength` (sift.c:372-384): slot j keeps CANDIDATE j's value
            # (Q2), so only the first len(kept) values of the global candidate order matter
            need, got, chunks = len(kept), 0, []
            for k in range(nkey):
                for r in range(self.world):
                    if got >= need:
                        break
                    c = vals[r][cc[r, k]:cc[r, k + 1]]
                    chunks.append(c)
                    got += len(c)
            val_head = np.concatenate(chunks)[:need] if chunks else np.zeros(0, np.float32)
        else:
            kept, val_head = rec, val[:len(rec)]
        self.ncand = int(cnts[:, :, 0].sum())
        kp = np.zeros(len(kept), KP_DTYPE)
        kp["o"], kp["s"] = kept["o"], kept["s"]
        kp["xd"], kp["yd"], kp["zd"] = kept["x"], kept["y"], kept["z"]
        sd_tab = np.array([[self._scale(o, s) for s in range(g.K)] for o in range(g.num_octaves)])
        kp["sd"] = sd_tab[kept["o"], kept["s"]] if len(kept) else 0.0
        kp["R"] = kept["R"].reshape(-1, 3, 3)
        kp["strength"] = val_head
        self.kp = kp
        self.t_pyr = be.elapsed(*self._ev)
        return kp

    # ---- describe -----------------------------------------------------------------------------
    def describe(self, kp=None):
        """Descriptors of the keypoints this rank owns (by z).  Returns (indices into kp,
        hist [n, 768]); the union over ranks covers every keypoint exactly once."""
        g, be, r = self.g, self.be, self.rank
        kp = self.kp if kp is None else kp
        from sift3d_amd.hip import KP_DTYPE as HKP
        if self.world == 1:
            idx = np.arange(len(kp))
        else:
            z0s = np.array([g.own(o, r)[0] for o in range(g.num_octaves)], np.float64)
            z1s = np.array([g.own(o, r)[1] for o in range(g.num_octaves)], np.float64)
            ko = kp["o"]
            idx = np.nonzero((kp["zd"] >= z0s[ko]) & (kp["zd"] < z1s[ko]))[0]
        # launch order: widest windows (largest s) first, each histogram to its own row
        ks = kp["s"][idx]
        order = np.argsort(-ks.astype(np.int64), kind="stable")
        sel = idx[order]
        q = np.zeros(len(idx), HKP)
        q["R"] = kp["R"][sel].reshape(-1, 9)
        q["cx"], q["cy"], q["cz"] = kp["xd"][sel], kp["yd"][sel], kp["zd"][sel]
        q["level"] = kp["o"][sel] * g.ngl + ks[order] + 1
        q["row1"] = order + 1
        q["sd"] = kp["sd"][sel]
        self.my_kp_idx = idx
        self.my_desc = be.describe(self._table, q)
        return idx, self.my_desc

    def gather_descriptors(self):
        """N x 771 matrix (sift3d_descriptor_store_to_mat_rm layout) on every rank."""
        rec = np.zeros(len(self.my_kp_idx), np.dtype([("i", "i8"), ("h", "f4", (768,))]))
        rec["i"], rec["h"] = self.my_kp_idx, self.my_desc
        parts = self._allgather_records(rec)
        out = np.zeros((len(self.kp), 771), np.float32)
        f = np.ldexp(1.0, self.kp["o"])
        out[:, 0], out[:, 1], out[:, 2] = self.kp["xd"] * f, self.kp["yd"] * f, self.kp["zd"] * f
        for p in parts:
            out[p["i"], 3:] = p["h"]
        return out

    # ---- bench hooks ---------------------------------------------------------------------------
    def step(self):
        if hasattr(self.be, "hip"):
            self.be.hip.current_stream(refresh=True)
        self.detect()
        self.describe()

    def stats(self):
        return dict(candidates=int(self.ncand), keypoints=int(len(self.kp)),
                    keypoints_rank0=int(len(self.my_kp_idx)),
                    sharded_octaves=int(self.g.o_shard), octaves=int(self.g.num_octaves))

    def pyramid_seconds(self):
        return self.t_pyr
