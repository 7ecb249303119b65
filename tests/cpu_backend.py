"""CPU compute backend for tests.sharded_py -- TEST INFRASTRUCTURE ONLY.

Implements the backend interface of tests.sharded_py.HipBackend with the oracle
(oracle/sift3d_oracle.c) on torch CPU tensors, so that the Z-slab orchestration (slab
geometry, halo exchange, reductions, gather order) can be exercised with gloo on machines
without a GPU.  The product never imports this module.
"""
import ctypes as C
import time

import numpy as np
import torch

from oracle import sift3d_oracle as so

CAND_DTYPE = np.dtype([("idx", "u4"), ("tag", "i4"), ("val", "f4")])


class OracleBackend:
    device = "cpu"

    def __init__(self):
        self.L = so.lib()
        self.L.orc_orient_slab.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           np.ctypeslib.ndpointer(np.float64), C.c_double, C.c_int,
                                           C.c_int, C.c_int, C.c_double,
                                           np.ctypeslib.ndpointer(np.float32)]
        self.L.orc_describe_slab.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             np.ctypeslib.ndpointer(np.float64), C.c_void_p,
                                             C.c_void_p]

    def gauss_filter(self, sigma):
        return so.gauss_taps(sigma)

    def empty(self, shape):
        return torch.full(tuple(shape), float("nan"), dtype=torch.float32)

    def scalar(self, n=1):
        return torch.zeros(n, dtype=torch.float32)

    def absmax(self, t, out):
        if t.numel():
            out[0] = max(float(out[0]), float(t.abs().max()))

    def scale(self, src, dst, mx):
        m = np.float32(mx[0].item())
        dst.copy_(src if m == 0 else torch.from_numpy(src.numpy() / m))

    def fir(self, src, dst, axis, taps, uf, n_glob, off, z_lo, z_hi):
        s = src.numpy()
        nz, ny, nx = s.shape
        if axis == 2:
            out, r = so.fir_axis(s, taps, 2, uf=uf, n_glob=n_glob, off=off, out_lo=z_lo,
                                 out_hi=z_hi, mode=1)
            assert r == 0, "halo too thin for the z pass"
            dst.numpy()[z_lo:z_hi] = out[z_lo:z_hi]
        else:
            out, r = so.fir_axis(s[z_lo:z_hi], taps, axis, uf=uf, mode=1)
            assert r == 0
            dst.numpy()[z_lo:z_hi] = out

    def subtract_absmax(self, a, b, dst, out):
        d = a.numpy() - b.numpy()
        dst.copy_(torch.from_numpy(d))
        if d.size:
            out[0] = max(float(out[0]), float(np.abs(d).max()))

    def downsample2(self, src, dst):
        mz, my, mx = dst.shape
        dst.copy_(src[0:2 * mz:2, 0:2 * my:2, 0:2 * mx:2])

    def extrema(self, levels, nx, ny, nz, peak, cuboid=False):
        recs = []
        for lv in levels:
            p, c, n = lv["prev"].numpy(), lv["cur"].numpy(), lv["next"].numpy()
            thr = np.float32(peak * float(lv["absmax"][0].item()))   # sift.c:829
            zl, zh = lv["z_lo"], lv["z_hi"]
            if zh <= zl:
                continue
            v = c[zl:zh, 1:-1, 1:-1]
            nb = [p[zl:zh, 1:-1, 1:-1], n[zl:zh, 1:-1, 1:-1], c[zl:zh, 1:-1, 2:], c[zl:zh, 1:-1, :-2],
                  c[zl:zh, 2:, 1:-1], c[zl:zh, :-2, 1:-1], c[zl - 1:zh - 1, 1:-1, 1:-1],
                  c[zl + 1:zh + 1, 1:-1, 1:-1]]
            if cuboid:                                                # sift.c:761-796
                nb = [a[zl + dz:zh + dz, 1 + dy:a.shape[1] - 1 + dy, 1 + dx:a.shape[2] - 1 + dx]
                      for a in (p, c, n) for dz in (-1, 0, 1) for dy in (-1, 0, 1)
                      for dx in (-1, 0, 1) if not (a is c and dz == dy == dx == 0)]
            gt = np.ones(v.shape, bool)
            lt = np.ones(v.shape, bool)
            for q in nb:
                gt &= v > q
                lt &= v < q
            hit = ((v > thr) | (v < -thr)) & (gt | lt)               # sift.c:842-849
            z, y, x = np.nonzero(hit)                                # C order = scan order
            r = np.zeros(len(z), CAND_DTYPE)
            r["idx"] = (x + 1) + nx * ((y + 1) + ny * (z + zl))
            r["tag"] = lv["tag"]
            r["val"] = np.abs(v[hit])
            recs.append(r)
        return np.concatenate(recs) if recs else np.zeros(0, CAND_DTYPE)

    def level_table(self, levels):
        return levels

    def extrema_orient(self, specs, table, peak, corner, cuboid=False):
        recs = [self.extrema(lv, nx, ny, nz, peak, cuboid) for lv, nx, ny, nz in specs]
        local = np.concatenate(recs) if recs else np.zeros(0, CAND_DTYPE)
        R, keep = self.orient(table, local, corner)
        kpos = np.nonzero(keep)[0].astype(np.int64)
        return (local["tag"].astype(np.int32), np.ascontiguousarray(local["val"]), kpos,
                local["idx"][kpos].astype(np.uint32), R[kpos])

    def orient(self, table, cands, corner):
        n = len(cands)
        R = np.zeros((n, 9), np.float32)
        keep = np.zeros(n, np.int32)
        for i, c in enumerate(cands):
            L = table[int(c["tag"])]
            t = L["data"]
            nzl, ny, nx = t.shape
            idx = int(c["idx"])
            x, y, zl = idx % nx, (idx // nx) % ny, idx // (nx * ny)
            r = np.zeros(9, np.float32)
            keep[i] = self.L.orc_orient_slab(t.data_ptr(), nx, ny, nzl, L["off"], L["nz_glob"],
                                             np.array(L["units"], np.float64), L["sd"], x, y,
                                             zl + L["off"], corner, r)
            R[i] = r
        return R, keep

    def describe(self, table, kps):
        n = len(kps)
        out = np.zeros((n, 768), np.float32)
        for i, k in enumerate(kps):
            L = table[int(k["level"])]
            t = L["data"]
            nzl, ny, nx = t.shape
            key = so.Keypoint()
            for j in range(9):
                key.R[j] = float(k["R"][j])
            key.xd, key.yd, key.zd, key.sd = float(k["cx"]), float(k["cy"]), float(k["cz"]), float(k["sd"])
            key.o, key.s = int(L["octave"]), 0
            d = so.Descriptor()
            self.L.orc_describe_slab(t.data_ptr(), nx, ny, nzl, L["off"], L["nz_glob"],
                                     np.array(L["units"], np.float64), C.byref(key), C.byref(d))
            row = int(k["row1"]) - 1 if int(k["row1"]) else i   # as the HIP kernel
            out[row] = np.ctypeslib.as_array(d.hist)
        return out

    def synth(self, t, z_off, seed):
        nz, ny, nx = t.shape
        for z in range(nz):
            for y in range(ny):
                for x in range(nx):
                    t[z, y, x] = self.L.sift3d_amd_synth_lattice_voxel(x, y, z + z_off, seed)

    def sync(self):
        pass

    def event(self):
        return time.perf_counter()

    def elapsed(self, e0, e1):
        return e1 - e0
