#!/usr/bin/env python3
"""desc_trace.py -- bin addresses of real descriptor windows, for lds_f64_atomic.hip.

Restates the window scan of extract_descrip (sift.c:1442-1536) in numpy for NKP keypoints of the
128^3 lattice volume (detected by the oracle), in the kernel's scan order (z, y, x), and writes
per window voxel the base cell of its 2x2x2 block (shifted to stay inside the 4x4x4 grid, as
k_describe does) and the three histogram vertices of its icosahedron face:

    desc_trace.bin : int32 header [nvox, nwin], then nvox records of 4 x uint8
                     (base cell 0..42, vertex 0, vertex 1, vertex 2), then nwin+1 int32 window offsets

The face is taken as the one whose plane normal has the largest product with the gradient (the
face a ray through the direction crosses) -- this is a load generator for a microbenchmark, not a
parity path.  Test/profile infrastructure: imports the oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sift3d_oracle as so  # noqa: E402


def main(out, n=128, nkp=48):
    vol = so.synth_lattice(n, seed=11)
    o = so.Oracle()
    assert o.detect(vol) == 0
    kps = o.keypoints()
    mv, midx = o.mesh()
    nrm = np.cross(mv[:, 1] - mv[:, 0], mv[:, 2] - mv[:, 0]).astype(np.float64)
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    ctr = mv.mean(1)
    nrm *= np.sign((nrm * ctr).sum(1))[:, None]
    sel = np.linspace(0, len(kps) - 1, nkp).astype(int)
    recs, offs = [], [0]
    for k in kps[sel]:
        G, units, _ = o.level(0, int(k["o"]), int(k["s"]))
        u = np.float32(units[0])
        nz, ny, nx = G.shape
        sigma = np.float32(k["sd"] * 7.071067812)
        rad = np.float32(2.0 * float(sigma))
        half_w = np.float32(float(rad) / np.sqrt(2.0))
        bin_f = np.float32(1.0) / (np.float32(2.0) * half_w / np.float32(4.0))
        c = np.array([k["xd"], k["yd"], k["zd"]], np.float32)
        R = k["R"].reshape(3, 3).astype(np.float32)
        lo = np.maximum(np.floor(c - rad / u), 1).astype(int)
        hi = np.minimum(np.ceil(c + rad / u), np.array([nx, ny, nz]) - 2).astype(int)
        z, y, x = np.meshgrid(np.arange(lo[2], hi[2] + 1), np.arange(lo[1], hi[1] + 1),
                              np.arange(lo[0], hi[0] + 1), indexing="ij")
        z, y, x = z.ravel(), y.ravel(), x.ravel()          # scan order z, y, x
        d = np.stack([(x - c[0]) * u, (y - c[1]) * u, (z - c[2]) * u], 1).astype(np.float32)
        sq = (d * d).sum(1)
        vkp = d @ R                                         # Rt * d
        vb = (vkp + half_w) * bin_f
        ok = (sq <= rad * rad) & (vb >= 0).all(1) & (vb < 4).all(1)
        x, y, z, vb = x[ok], y[ok], z[ok], vb[ok]
        g = np.stack([G[z, y, x + 1] - G[z, y, x - 1], G[z, y + 1, x] - G[z, y - 1, x],
                      G[z + 1, y, x] - G[z - 1, y, x]], 1).astype(np.float64)
        gr = g @ R.astype(np.float64)
        live = (gr * gr).sum(1) > 1e-12
        face = np.argmax(gr @ nrm.T, 1)
        base = np.minimum(vb.astype(int), 2)
        cell = base[:, 0] + 4 * base[:, 1] + 16 * base[:, 2]
        r = np.stack([cell, midx[face, 0], midx[face, 1], midx[face, 2]], 1).astype(np.uint8)
        r = r[live]
        recs.append(r)
        offs.append(offs[-1] + len(r))
    recs = np.concatenate(recs)
    with open(out, "wb") as f:
        np.array([len(recs), len(offs) - 1], np.int32).tofile(f)
        recs.tofile(f)
        np.array(offs, np.int32).tofile(f)
    # how often a voxel repeats its predecessor's 24 bins (DESIGN 3.3 quotes 44 %)
    same = (recs[1:] == recs[:-1]).all(1).mean()
    print("%d window voxels of %d windows -> %s; %.1f %% repeat their predecessor's bins"
          % (len(recs), len(offs) - 1, out, 100 * same))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "desc_trace.bin"))
