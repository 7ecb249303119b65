#!/usr/bin/env python3
"""Launch every octave-0 FIR kernel instance a few times at 512^3 (for rocprofv3 --pmc passes).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 profiles/pmc_fir.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 profiles/pmc_fir.py
    python3 profiles/pmc_fir.py --parse gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/traffic.json

Round 5: the HBM bytes of ONE WHOLE pyramid build (every kernel of every octave), for bench.py's
`pyramid_hbm_frac` -- two more passes over `--pyramid` (two builds and nothing else), parsed with two more
directories:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_pyr_fetch -- python3 profiles/pmc_fir.py --pyramid
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_pyr_write -- python3 profiles/pmc_fir.py --pyramid
    python3 profiles/pmc_fir.py --parse <fetch> <write> <pyr fetch> <pyr write> > profiles/traffic.json
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _require_built():
    """These entry points run under rocprofv3, whose preloaded library has already initialised the GPU:
    building from here (a fork + exec of make) is not allowed on this pool.  Build first."""
    lib = os.environ.get("SIFT3D_AMD_LIB") or os.path.join(ROOT, "sift3d_amd", "libsift3d_amd.so")
    if not os.path.exists(lib):
        sys.exit("%s is missing -- build first: python3 -c \"from sift3d_amd import _native; "
                 "_native.build()\"" % lib)

SIG = [0.5387011637869722, 0.9732939207323564, 1.2262734984654078, 1.5450077936447955,
       1.9465878414647133, 2.4525469969308156]


def run_guarded(*a, **k):
    _require_built()
    return run(*a, **k)


def run(n=512, reps=3):
    import torch
    from sift3d_amd import api, hip
    src = torch.empty((n, n, n), device="cuda")
    dst = torch.empty_like(src)
    hip.synth_lattice(src, 0, 11)
    for s in SIG:
        taps = api.gauss_filter(s)
        for ax in range(3):
            for _ in range(reps):
                hip.fir(src, dst, ax, taps)
        for _ in range(reps):
            hip.fir_yz(src, dst, taps)
    torch.cuda.synchronize()


def run_pyramid(n=512, builds=2):
    """`builds` whole Gaussian pyramid builds (sift3d_amd_build_pyramid_device) and nothing else: every
    dispatch of the process belongs to one, so the per-build HBM bytes are the counter sums / builds."""
    import torch
    from sift3d_amd import api, hip
    _require_built()
    vol = torch.empty((n, n, n), device="cuda")
    hip.synth_lattice(vol, 0, 11)
    torch.cuda.synchronize()
    det = api.Detector()
    for _ in range(builds):
        assert det.build_pyramid_device(vol.data_ptr(), n, n, n) == 0
    torch.cuda.synchronize()


def parse_pyramid(fetch_dir, write_dir, n=512, builds=2):
    """HBM bytes of one whole pyramid build: sums over every dispatch of run_pyramid() but the volume's
    synthesis, / builds; FETCH_SIZE doubled as in parse()."""
    def total(d, name):
        tot, by = 0.0, {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != name or "synth" in r["Kernel_Name"]:
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                v = float(r["Counter_Value"])
                tot += v
                by[k] = by.get(k, 0.0) + v
        return tot, by
    fe, fby = total(fetch_dir, "FETCH_SIZE")
    wr, wby = total(write_dir, "WRITE_SIZE")
    num_oct = 0
    m = n
    alg = 0
    while m >= 8:
        alg += 24 * m ** 3 * (6 if num_oct == 0 else 5)
        num_oct += 1
        m //= 2
    hbm = (2.0 * fe + wr) * 1024.0 / builds
    by = {k: round((2.0 * fby.get(k, 0.0) + wby.get(k, 0.0)) * 1024.0 / builds) for k in set(fby) | set(wby)}
    return dict(hbm_bytes=round(hbm), algorithmic_bytes=alg, ratio=round(hbm / alg, 3), builds=builds,
                note="every kernel of sift3d_amd_build_pyramid_device (max|v|, all FIR passes of all octaves, "
                     "down-sampling): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over its dispatches",
                hbm_bytes_by_kernel=dict(sorted(by.items(), key=lambda kv: -kv[1])))


def parse(fetch_dir, write_dir, n=512):
    """Per kernel: mean counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB-ish units of
    1024 B (guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); on gfx950 FETCH_SIZE reports half
    of the bytes of a wide coalesced streaming read, so it is doubled (MI355X_MICROARCH.md, HBM)."""
    def load(d, name):
        out = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != name:
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                gx = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else 0
                out.setdefault((k, gx), []).append(float(r["Counter_Value"]))
        return out
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    for (k, gx), v in fe.items():
        if "fir" not in k:
            continue
        w = wr.get((k, gx), [0.0])
        fetch_b = 2.0 * 1024.0 * sum(v) / len(v)
        write_b = 1024.0 * sum(w) / len(w)
        alg = (16 if "yz" in k else 8) * n ** 3
        res[k] = dict(fetch_bytes=round(fetch_b), write_bytes=round(write_b),
                      hbm_bytes=round(fetch_b + write_b), algorithmic_bytes=alg,
                      ratio=round((fetch_b + write_b) / float(alg), 3), dispatches=len(v))
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--parse":
        # --parse <fetch dir> <write dir> [<pyramid fetch dir> <pyramid write dir>]
        res = parse(sys.argv[2], sys.argv[3])
        if len(sys.argv) > 5:
            res["pyramid_build_512"] = parse_pyramid(sys.argv[4], sys.argv[5])
        print(json.dumps(res, indent=1, sort_keys=True))
    elif len(sys.argv) > 1 and sys.argv[1] == "--pyramid":
        run_pyramid()
    else:
        run_guarded()
