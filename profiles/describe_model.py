#!/usr/bin/env python3
"""Counter totals of k_describe (profiles/describe_model.json, read by bench.py).

    # on the GPU box (cd /tmp && export TMPDIR=/tmp first), three runs of this script:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/pmc_desc/a -o run --output-format csv -- python3 profiles/describe_model.py --run
    rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d gpurun_out/pmc_desc/b -o run --output-format csv -- python3 profiles/describe_model.py --run
    python3 profiles/describe_model.py --count          # diagnostic build: exact window voxels
    python3 profiles/describe_model.py --parse gpurun_out/pmc_desc <window_voxels> > profiles/describe_model.json
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


def run_guarded(*a, **k):
    _require_built()
    return run(*a, **k)


def run(n=512):
    import torch
    from sift3d_amd import api, hip
    vol = torch.empty((n, n, n), device="cuda")
    hip.synth_lattice(vol, 0, 11)
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0
    assert det.extract_descriptors(kp, desc) == 0
    return det, kp, desc


def count():
    """Needs the diagnostic build (make -C sift3d_amd/csrc DIAG=-DSIFT3D_AMD_DIAG, see the Makefile) in place of the library."""
    import ctypes as C
    from sift3d_amd import api
    L = api.lib()
    L.sift3d_amd_diag_desc_voxels.restype = C.c_ulonglong
    L.sift3d_amd_diag_desc_voxels()
    _require_built()
    det, kp, desc = run()
    print(json.dumps(dict(window_voxels=int(L.sift3d_amd_diag_desc_voxels()), keypoints=len(kp))))


def parse(d, voxels):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_describe" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    out = dict(kernel="k_describe", workload="512^3 lattice volume, 42 501 keypoints", window_voxels=int(voxels),
               valu_insts=c.get("SQ_INSTS_VALU"), salu_insts=c.get("SQ_INSTS_SALU"),
               lds_insts=c.get("SQ_INSTS_LDS"), lds_array_cycles=c.get("SQ_LDS_IDX_ACTIVE"),
               lds_bank_conflict_cycles=c.get("SQ_LDS_BANK_CONFLICT"), waves=c.get("SQ_WAVES"),
               wave_quad_cycles=c.get("SQ_WAVE_CYCLES"), wait_any=c.get("SQ_WAIT_ANY"),
               wait_inst_any=c.get("SQ_WAIT_INST_ANY"), wait_inst_lds=c.get("SQ_WAIT_INST_LDS"),
               note="wave-instructions per launch (rocprofv3 --pmc); lds_array_cycles is summed over the 256 CUs; "
                    "bench.py prices both against the kernel's own cycle count of its run (clock probe)")
    if voxels and c.get("SQ_INSTS_VALU"):
        out["valu_insts_per_64_voxels"] = round(64.0 * c["SQ_INSTS_VALU"] / voxels, 1)
        out["lds_insts_per_64_voxels"] = round(64.0 * c.get("SQ_INSTS_LDS", 0) / voxels, 1)
        out["lds_array_cycles_per_64_voxels"] = round(64.0 * c.get("SQ_LDS_IDX_ACTIVE", 0) / voxels, 1)
    # Clock-free fractions (round 5; the 2.4 GHz constant and the LDS issue model of rounds 2-4 are gone: the
    # chip does not hold 2.4 GHz under this kernel, and the model priced more pipe cycles than the waves were
    # alive).  Every wave of the launch is persistent, so SQ_WAVE_CYCLES (units of 4 cycles, summed over the
    # waves) * 4 / waves = the kernel's duration in shader cycles; the LDS array's busy cycles are summed over
    # the 256 CUs, and one wave64 VALU instruction occupies one of the 1024 SIMDs for 2 cycles (guide: SIMD-32).
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAVES"):
        kc = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]
        out["profiled_kernel_cycles"] = round(kc)
        out["lds_array_busy"] = round(c.get("SQ_LDS_IDX_ACTIVE", 0) / (256.0 * kc), 4)
        out["valu_busy"] = round(2.0 * c.get("SQ_INSTS_VALU", 0) / (1024.0 * kc), 4)
        if c.get("GRBM_GUI_ACTIVE"):
            out["grbm_gui_active_cycles"] = c["GRBM_GUI_ACTIVE"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if "--run" in sys.argv:
        run_guarded()
    elif "--count" in sys.argv:
        count()
    elif "--parse" in sys.argv:
        i = sys.argv.index("--parse")
        parse(sys.argv[i + 1], float(sys.argv[i + 2]))
