#!/usr/bin/env python3
"""Phase 1 of the octave-0 extrema stage ALONE on the device (zero the border planes, k_extrema_sweep3g<64, true>,
the two border planes' maxima, refilter, count) on six random 512^3 levels -- in a step the sweep shares the device
with the chains of the smaller octaves.  Run under `rocprofv3 --kernel-trace --stats` for the sweep kernel's own
duration (profiles/scripts/r5_profiles.sh keeps the summary as profiles/r05_sweep_alone.txt):

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o run -- python3 profiles/microbench/sweep_alone.py

Random levels: the sweep's time does not depend on the data (no data-dependent branch); the refilter's does (it
visits the marked voxels: a few hundred thousand on the bench volume, ~10^8 on noise), so only the sweep's line of the
summary means anything."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sift3d_amd import hip  # noqa: E402


def main():
    n = int(os.environ.get("N", "512"))
    lib = hip.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    g = [torch.rand(n, n, n, device=dev) for _ in range(6)]
    ptrs = (C.c_void_p * 6)(*[t.data_ptr() for t in g])
    wb = lib.sift3d_hip_extrema_work_bytes(n, n, n, 3)
    work = torch.zeros(wb, dtype=torch.uint8, device=dev)
    est = torch.full((5,), 0.5, device=dev)
    exact = torch.zeros(5, device=dev)
    count = torch.zeros(4, dtype=torch.int32, device=dev)
    out = torch.zeros(1 << 20, 3, dtype=torch.int32, device=dev)
    f = lib.sift3d_hip_extrema_gauss6_est_phase
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                  C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]

    def run():
        rc = f(ptrs, est.data_ptr(), exact.data_ptr(), n, n, n, 0, 0.1, out.data_ptr(), 1 << 20, count.data_ptr(),
               work.data_ptr(), wb, None, 1)
        assert rc == 0, rc

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    print("sweep_alone: %d^3, 13 launches of phase 1; algorithmic bytes per sweep launch: %d" % (n, 6 * 4 * n ** 3))


if __name__ == "__main__":
    main()
