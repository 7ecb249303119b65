#!/usr/bin/env python3
"""bench.py -- SIFT3D detect+describe throughput on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W            # one GPU, 512^3 (headline)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (sift3d_detect_keypoints + sift3d_extract_descriptors
through the drop-in C API) over one synthetic float32 volume that is already resident in
HBM when the timed region starts.  N = 1: the 512^3 volume BASELINE.json's metric is quoted
on (configs[2]).  N > 1: every rank owns one Z-slab of 512 planes of a 512 x 512 x (512 N)
volume (weak scaling; halo exchange + keypoint gather over RCCL, driven by the C slab driver
sift3d_amd/csrc/sift3d_sharded.c).

    python bench.py --gpus N --strong 1024 ...                # BASELINE configs[3]: ONE 1024^3 volume
                                                              # as N Z-slabs (strong scaling)

Rank 0 prints ONE JSON line: metric/value (Mvoxel/s, whole job), ms_per_step, plus
  roofline     the dominant pyramid kernel: algorithmic bytes per launch (8 B/voxel per 1-D
               pass) / its average launch time measured here with HIP events,
               against the 8 TB/s HBM3E peak; `pyramid` = the same ratio for the whole
               Gaussian pyramid build (21.63 GB algorithmic at 512^3)
  cpu_baseline the unmodified reference (oracle/_ref, kind "reference") or the oracle
               restatement (kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# CPU-baseline threading: one GPU's share of the host (16 cores on the GPU box); must be set
# before libgomp / OpenBLAS are loaded.  The reference's LAPACK is called from every OpenMP
# thread, and SciPy's OpenBLAS supports at most 128 callers.
CPU_THREADS = min(16, os.cpu_count() or 1)
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_THREADS))
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def pyramid_algorithmic_bytes(nx, ny, nz):
    """24 B/voxel/blur (3 passes x (4 B read + 4 B write)); 6 blurs on octave 0, 5 on each
    later octave (SURVEY.md 8d / BASELINE.md section 3)."""
    mn = min(nx, ny, nz)
    num_oct = int(np.log2(mn)) - 3 + 1
    total = 0
    d = [nx, ny, nz]
    for o in range(num_oct):
        n = d[0] * d[1] * d[2]
        total += 24 * n * (6 if o == 0 else 5)
        d = [v // 2 for v in d]
    return total


def kernel_microbench(torch, hip, n, reps=10):
    """Each octave-0 FIR kernel instance at the headline size, timed with HIP events on the
    stream it is launched on (torch's current stream)."""
    from sift3d_amd import api
    sig = [0.5387011637869722, 0.9732939207323564, 1.2262734984654078, 1.5450077936447955,
           1.9465878414647133, 2.4525469969308156]
    src = torch.empty((n, n, n), device="cuda")
    dst = torch.empty_like(src)
    tmp = torch.empty_like(src)
    hip.synth_lattice(src, 0, 11)
    out = []

    def row(nm, hw, ax, nbytes, ms, in_pipeline):
        out.append(dict(kernel=nm, taps=2 * hw + 1, axis=ax, avg_ms=round(ms, 4),
                        in_pipeline=in_pipeline, algorithmic_GB=round(nbytes * n ** 3 / 1e9, 4),
                        achieved_GBs=round(nbytes * n ** 3 / 1e9 / (ms * 1e-3), 1)))

    for s in sig:
        taps = api.gauss_filter(s)          # the detector's own filter bank
        hw = len(taps) // 2
        # What the pipeline launches at octave 0, in the pipeline's own sequence (level -> x pass
        # -> scratch -> fused y+z pass -> next level, ping-pong), so that every launch sees the
        # cache state it sees in a real pyramid build; one event pair per launch.
        a, b = src, dst
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * reps)]
        hip.fir(a, tmp, 0, taps)
        hip.fir_yz(tmp, b, taps)
        torch.cuda.synchronize()
        for r in range(reps):
            ev[3 * r].record()
            hip.fir(a, tmp, 0, taps)
            ev[3 * r + 1].record()
            hip.fir_yz(tmp, b, taps)
            ev[3 * r + 2].record()
            a, b = b, a
        torch.cuda.synchronize()
        x_ms = sum(ev[3 * r].elapsed_time(ev[3 * r + 1]) for r in range(reps)) / reps
        yz_ms = sum(ev[3 * r + 1].elapsed_time(ev[3 * r + 2]) for r in range(reps)) / reps
        row("k_fir_x_u1<%d>" % hw, hw, 0, 8.0, x_ms, True)
        row("k_fir_yz_u1<%d, 32, %d>" % (hw, 32 if n % 128 == 0 else 16), hw, 12, 16.0, yz_ms, True)   # two 1-D passes = 16 B/voxel

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        # the separate y and z kernels are the fallback / slab-halo path
        hip.synth_lattice(src, 0, 11)       # (the ping-pong above blurred it away)
        row("k_fir_sweep_u1<%d, 4> (y)" % hw, hw, 1, 8.0, timed(lambda: hip.fir(src, dst, 1, taps)), False)
        row("k_fir_sweep_u1<%d, 4> (z)" % hw, hw, 2, 8.0, timed(lambda: hip.fir(src, dst, 2, taps)), False)
    del tmp
    del src, dst
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _set_omp_threads(k):
    """libgomp reads OMP_NUM_THREADS once; later changes go through omp_set_num_threads."""
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(k))
        return True
    except OSError:
        return False


def cpu_baseline(n_all=128, n_one=96):
    """The reference's CPU path on bounded samples of the same kind of volume: once on this
    GPU's share of the host cores, once on ONE thread (SURVEY.md 8d asks for both)."""
    threads = int(os.environ.get("OMP_NUM_THREADS", CPU_THREADS))
    from oracle import sift3d_oracle as so
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libsift3d_refprobe.so")
    kind = "port"
    if os.path.exists(ref_lib):
        try:
            from oracle import refprobe
            refprobe.Probe().close()
            kind = "reference"
        except Exception as e:  # the reference build did not load on this host
            sys.stderr.write("cpu_baseline: reference unavailable (%s), using the port\n" % e)

    def run(n):
        vol = so.synth_lattice(n, seed=11)
        if kind == "reference":
            p = refprobe.Probe()
            t0 = time.time()
            assert p.detect_public(vol) == 0
            t1 = time.time()
            assert p.describe() == 0
            t2 = time.time()
            nkp = len(p.keypoints()["strength"])
            p.close()
        else:
            o = so.Oracle()
            t0 = time.time()
            assert o.detect(vol) == 0
            t1 = time.time()
            assert o.describe() == 0
            t2 = time.time()
            nkp = len(o.keypoints())
        return n ** 3 / 1e6 / (t2 - t0), t1 - t0, t2 - t1, nkp

    v_all, d_all, e_all, k_all = run(n_all)
    one = None
    if _set_omp_threads(1):
        v_one, d_one, e_one, k_one = run(n_one)
        _set_omp_threads(threads)
        one = dict(value=round(v_one, 4), cores=1,
                   sample="%d^3 lattice volume: detect %.2f s + describe %.2f s, %d keypoints"
                          % (n_one, d_one, e_one, k_one))
    what = ("unmodified reference libsift3D (OpenMP)" if kind == "reference"
            else "oracle restatement (OpenMP)")
    out = dict(value=round(v_all, 4), unit="Mvoxel/s", cores=threads, kind=kind,
               cpu_model=_cpu_model(),
               sample="%d^3 lattice volume (1/%d of the workload's voxels), %s: detect %.2f s + "
                      "describe %.2f s, %d keypoints" % (n_all, (512 // n_all) ** 3, what, d_all,
                                                         e_all, k_all),
               one_thread=one)
    # context: the reference's own run of THIS workload (the 512^3 fixture, tests/golden/MANIFEST.json,
    # generated in the 8-core build container with the default OpenMP thread count)
    try:
        man = json.load(open(os.path.join(ROOT, "tests", "golden", "MANIFEST.json")))["g5_512"]
        out["reference_512_fixture"] = dict(
            detect_s=man["t_detect"], describe_s=man["t_describe"],
            value=round(512 ** 3 / 1e6 / (man["t_detect"] + man["t_describe"]), 4),
            note="unmodified reference on the whole 512^3 workload when the fixture was made "
                 "(build container, 8 cores, OMP default)")
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512, help="edge of the (per-GPU) volume")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-micro", action="store_true", help="skip the per-kernel microbench")
    ap.add_argument("--strong", type=int, default=0, metavar="EDGE",
                    help="strong scaling: ONE EDGE^3 volume (BASELINE configs[3]: 1024) cut into "
                         "--gpus Z-slabs, instead of the default 512 planes per GPU")
    ap.add_argument("--no-host", action="store_true", help="skip the host-resident (PCIe-inclusive) leg")
    ap.add_argument("--py-driver", action="store_true",
                    help="N > 1 / --sharded: use the Python slab driver instead of the C one")
    ap.add_argument("--sharded", action="store_true",
                    help="N=1 only: run the Z-slab driver (C: sift3d_amd_sharded_*) instead of the "
                         "drop-in API, to measure the driver's own overhead")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sift3d_amd import api, hip

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % a.gpus)
    if not torch.cuda.is_available() or not api.device_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path to measure")
    # SIFT3D_AMD_REHEARSE=1: all ranks share device 0 and talk through gloo -- a rehearsal of
    # the N > 1 code path on a one-GPU box (the numbers mean nothing); the real run is RCCL
    rehearse = bool(os.environ.get("SIFT3D_AMD_REHEARSE")) and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    hip.lib().sift3d_hip_set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n = a.strong if a.strong else a.size
    nz_total = n if a.strong else n * world
    if world == 1 and not a.sharded:
        vol = torch.empty((n, n, n), device="cuda")
        hip.synth_lattice(vol, 0, 11)
        torch.cuda.synchronize()
        det = api.Detector()
        kp, desc = api.KeypointStore(), api.DescriptorStore()

        def step():
            rc = det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp)
            assert rc == 0, "detect failed"
            rc = det.extract_descriptors(kp, desc)
            assert rc == 0, "describe failed"

        stats = lambda: dict(candidates=det.num_candidates(), keypoints=len(kp),  # noqa: E731
                             stage_s={k: round(v, 6) for k, v in det.timings().items()})
        pyr_time = lambda: det.timings()["gauss_dev"]  # noqa: E731
    else:
        # The slab driver in C (sift3d_amd/csrc/sift3d_sharded.c) over the library's own RCCL
        # communicator; rehearsals on one device stage the exchanges through gloo.  The Python
        # driver (sift3d_amd/sharded.py) remains for configurations the C driver refuses.
        job = None
        if not a.py_driver:
            from sift3d_amd import sharded_c
            try:
                if world == 1:
                    tr = None
                elif rehearse:
                    tr = sharded_c.DistTransport()
                else:
                    tr = sharded_c.RcclTransport()
                job = sharded_c.CShardedSift3D(n, n, nz_total, tr)
            except (ValueError, RuntimeError) as e:
                sys.stderr.write("bench: C slab driver unavailable (%s), using the Python driver\n" % e)
                job = None
        if job is None:
            from sift3d_amd import sharded
            job = sharded.ShardedSift3D(n, n, nz_total, dist.group.WORLD if world > 1 else None)
        job.synth(seed=11)
        step = job.step
        stats = job.stats
        pyr_time = job.pyramid_seconds
    voxels_per_step = n * n * nz_total

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pyr = []
    for _ in range(a.steps):
        step()
        pyr.append(pyr_time())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if rehearse else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * dt / a.steps
    value = voxels_per_step / 1e6 / (dt / a.steps)
    if world == 1:
        slabs = ""
    elif a.strong:
        slabs = " cut into %d Z-slabs" % world
    else:
        slabs = " as %d Z-slabs of %d planes" % (world, n)
    out = {
        "metric": "Mvoxel/s detect+describe (float32 volume resident in HBM)",
        "value": round(value, 2), "unit": "Mvoxel/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "strong" if a.strong else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%dx%dx%d float32 lattice-blob volume%s, detect+describe, default "
                               "parameters (sigma0 1.6, sigma_n 1.15, 3 levels/octave)"
                               % (n, n, nz_total, slabs),
                   "parallelism": "single GPU" if world == 1 else "z-slab x%d" % world},
    }
    out.update(stats())
    # pyramid roofline (whole Gaussian pyramid build, per GPU)
    pbytes = pyramid_algorithmic_bytes(n, n, nz_total // world)
    pt = float(np.median(pyr)) if pyr and pyr[0] else None
    pyramid = None
    if pt:
        ach = pbytes / 1e9 / pt
        pyramid = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_GB": round(pbytes / 1e9, 3),
                   "seconds": round(pt, 6)}
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = None
    # host-resident entry (sift3d_detect_keypoints on a sift3d_make_image volume in pageable host
    # memory): the same K steps timed the same way, H2D copy included.  Reported beside `value`,
    # never as `value`.
    if world == 1 and not a.sharded and not a.no_host:
        im = api.Image.from_array(vol.cpu().numpy())
        kp2, desc2 = api.KeypointStore(), api.DescriptorStore()

        def host_step():
            assert det.detect_keypoints(im, kp2) == 0 and det.extract_descriptors(kp2, desc2) == 0

        host_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            host_step()
        torch.cuda.synchronize()
        th = (time.perf_counter() - t0) / a.steps
        assert len(kp2) == len(kp)
        out["value_host_resident"] = {"value": round(voxels_per_step / 1e6 / th, 2), "unit": "Mvoxel/s",
                                      "ms_per_step": round(1e3 * th, 3),
                                      "note": "volume in pageable host memory, PCIe H2D inside the "
                                              "timed region (sift3d_detect_keypoints on a "
                                              "sift3d_image)"}
        del im
    if world == 1 and not a.no_micro and not a.sharded and not a.strong:
        kb = kernel_microbench(torch, hip, n)
        # dominant kernel = the pipeline kernel with the longest launch
        dom = max((k for k in kb if k["in_pipeline"]), key=lambda k: k["avg_ms"])
        tr = None
        if traffic:
            # profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950
            # corrections applied by profiles/pmc_fir.py) is keyed by the kernel symbol
            sym = dom["kernel"].split(" (")[0]
            if sym in traffic:
                tr = traffic[sym]["hbm_bytes"]
        # `achieved` / `frac` price the ALGORITHMIC bytes (8 B per voxel and 1-D
        # pass) against the HBM peak; `hbm_GBs` is what the kernel really moves (measured
        # traffic / launch time): the fused y+z kernel keeps its intermediate on chip, so its
        # HBM rate is about half its algorithmic rate.
        out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBs"],
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(dom["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": tr,
                           "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                                             "WRITE_SIZE passes of this kernel, committed; not "
                                             "measured in this run)",
                           "hbm_GBs": round(tr / 1e9 / (dom["avg_ms"] * 1e-3), 1) if tr else None,
                           "algorithmic_bytes_per_launch": int(dom["algorithmic_GB"] * 1e9),
                           "avg_launch_ms": dom["avg_ms"], "pyramid": pyramid, "kernels": kb}
        # avg_launch_ms is the kernel ALONE on the device (this leg's launches).  Inside the step
        # the same launch -- the last blur of octave 0 -- shares the device with the streams that
        # build octaves >= 1 (HIP events on its own stream, last step of the timed region); the
        # rocprofv3 average in profiles/ is the mix of both kinds of launch.
        yz_last = out.get("stage_s", {}).get("yz_last")
        if yz_last:
            out["roofline"]["avg_launch_ms_in_pipeline"] = round(1e3 * yz_last, 4)
            out["roofline"]["in_pipeline_note"] = ("in the step this launch overlaps the octave >= 1 "
                                                   "streams; avg_launch_ms is the kernel alone")
        dpath = os.path.join(ROOT, "profiles", "describe_model.json")
        if os.path.exists(dpath):
            try:
                dm = json.load(open(dpath))
                dsec = out.get("stage_s", {}).get("describe")
                if dsec and dm.get("valu_insts"):
                    # issue-rate model of the descriptor kernel (VALU-issue / LDS-pipe bound, not HBM)
                    simd_rate = 256 * 4 * 2.4e9 / dm.get("cycles_per_valu", 2.5)
                    dm["achieved_valu_insts_per_s"] = round(dm["valu_insts"] / dsec, 1)
                    dm["peak_valu_insts_per_s"] = round(simd_rate, 1)
                    dm["valu_frac"] = round(dm["valu_insts"] / dsec / simd_rate, 4)
                    dm["seconds"] = dsec
                    lm = dm.get("lds_issue_model") or {}
                    if lm.get("seconds_if_lds_bound"):
                        # the binding resource: LDS instruction issue (profiles/microbench/lds_cost.hip)
                        dm["bound"] = "lds-issue"
                        dm["frac"] = round(lm["seconds_if_lds_bound"] / dsec, 4)
                    else:
                        dm["bound"] = "valu-issue"
                        dm["frac"] = dm["valu_frac"]
                    out["roofline"]["describe"] = dm
            except Exception:
                pass
    else:
        out["roofline"] = dict(pyramid or {}, kernel="Gaussian pyramid (all k_fir_* launches)",
                               traffic=None)
    if not a.no_cpu and world == 1:          # the CPU leg runs at N = 1 only
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
