#!/usr/bin/env python3
"""Issue-rate model of k_describe (profiles/describe_model.json, read by bench.py).

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
               cycles_per_valu=2.5,
               note="wave-instructions per launch (rocprofv3 --pmc); one VALU instruction occupies a "
                    "SIMD for ~2.5 cycles (scratch microbenchmark, plain f32 ops); "
                    "lds_array_cycles is summed over the 256 CUs")
    if voxels and c.get("SQ_INSTS_VALU"):
        out["valu_insts_per_64_voxels"] = round(64.0 * c["SQ_INSTS_VALU"] / voxels, 1)
        out["lds_insts_per_64_voxels"] = round(64.0 * c.get("SQ_INSTS_LDS", 0) / voxels, 1)
        out["lds_array_cycles_per_64_voxels"] = round(64.0 * c.get("SQ_LDS_IDX_ACTIVE", 0) / voxels, 1)
    # LDS instruction-issue model.  profiles/microbench/lds_cost.hip: on gfx950 one LDS instruction
    # occupies the CU's LDS pipe for a time set by its kind and dwords per lane, NOT by the number
    # of active lanes (cycles at 2.4 GHz, 16 waves per CU issuing back to back):
    cost = dict(read_b32=3.1, read_b128=4.6, write_b32=4.35, write2_b32=6.4)
    # k_describe per batch of 64 window voxels (static count of the main loop, sift3d_describe.hip):
    mix = dict(rmw_read_b32=32, rmw_write_b32=32,           # 2 passes x 16 rounds, 2 voxels per round
               record_read_b128=24,                         # 3 fields x 4 chunks x 2 passes
               record_write2_b32=6, record_write_b32=2,     # 7 dwords per lane per pass
               face_read_b128=4, octant_read_b32=1, queue_read_b32=1, queue_write_b32=2)
    out["issue_model_note"] = ("an LDS instruction priced at what the back-to-back microbenchmark measured "
                               "(lds_cost_mi355x.txt); the counters (SQ_LDS_IDX_ACTIVE, SQ_WAIT_INST_LDS) are "
                               "what the hardware reports for this kernel -- bench.py prints both")
    cyc = (mix["rmw_read_b32"] * cost["read_b32"] + mix["rmw_write_b32"] * cost["write_b32"] +
           mix["record_read_b128"] * cost["read_b128"] + mix["record_write2_b32"] * cost["write2_b32"] +
           mix["record_write_b32"] * cost["write_b32"] + mix["face_read_b128"] * cost["read_b128"] +
           (mix["octant_read_b32"] + mix["queue_read_b32"]) * cost["read_b32"] +
           mix["queue_write_b32"] * cost["write_b32"])
    out["lds_issue_model"] = dict(cost_cycles_per_instruction=cost, instructions_per_64_voxels=mix,
                                  lds_pipe_cycles_per_64_voxels=round(cyc, 1), clock_hz=2.4e9, cus=256,
                                  seconds_if_lds_bound=round(voxels / 64.0 / 256.0 * cyc / 2.4e9, 5) if voxels else None,
                                  note="the LDS pipe of a CU is the binding resource of k_describe: "
                                       "frac = seconds_if_lds_bound / measured seconds")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if "--run" in sys.argv:
        run_guarded()
    elif "--count" in sys.argv:
        count()
    elif "--parse" in sys.argv:
        i = sys.argv.index("--parse")
        parse(sys.argv[i + 1], float(sys.argv[i + 2]))
