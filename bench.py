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

    def row(nm, hw, ax, nbytes, times_ms, in_pipeline):
        """One kernel instance: median / min / mean over the reps; a rep slower than 3x the median is
        an outlier (a stall of the box, not the kernel) and is flagged, not hidden."""
        t = np.sort(np.asarray(times_ms, np.float64))
        med = float(np.median(t))
        out.append(dict(kernel=nm, taps=2 * hw + 1, axis=ax, avg_ms=round(med, 4), min_ms=round(float(t[0]), 4),
                        mean_ms=round(float(t.mean()), 4), reps=len(t),
                        outliers=int((t > 3.0 * med).sum()),
                        in_pipeline=in_pipeline, algorithmic_GB=round(nbytes * n ** 3 / 1e9, 4),
                        achieved_GBs=round(nbytes * n ** 3 / 1e9 / (med * 1e-3), 1),
                        frac=round(nbytes * n ** 3 / 1e9 / (med * 1e-3) / HBM_PEAK_GBS, 4)))

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
        x_ms = [ev[3 * r].elapsed_time(ev[3 * r + 1]) for r in range(reps)]
        yz_ms = [ev[3 * r + 1].elapsed_time(ev[3 * r + 2]) for r in range(reps)]
        # (the launchers' choices: sift3d_kernels.hip launch_fir_x_u1, sift3d_fir_yz.hip launch_fir_yz)
        row(("k_fir_x_u1f<%d, false>" if n % 512 == 0 else "k_fir_x_u1<%d>") % hw, hw, 0, 8.0, x_ms, True)
        if n % 64 == 0 and n >= 128:
            yzname = "k_fir_yz_dma<%d, %d>" % (hw, 64 if hw <= 2 else 32)
        else:
            yzname = "k_fir_yz_u1<%d, %s>" % (hw, "32, 32" if n % 128 == 0 else "32, 16")
        row(yzname, hw, 12, 16.0, yz_ms, True)   # two 1-D passes = 16 B/voxel

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            e[0].record()
            for r in range(reps):
                fn()
                e[r + 1].record()
            torch.cuda.synchronize()
            return [e[r].elapsed_time(e[r + 1]) for r in range(reps)]

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


def register_bench(a, torch, api, hip):
    """BASELINE configs[4]: two 512^3 volumes related by a known rigid motion; a step = detect+describe
    both + descriptor matching (sift3d_hip_nn2 on the matrix cores, both directions) + RANSAC affine.
    PARITY UNPINNED (the reference fork removed this stage): validated by recovering the motion."""
    n = a.size
    sy, sz = 5, 9
    vol = torch.empty((n + 24, n + 16, n), device="cuda")
    hip.synth_lattice(vol, 0, 21)
    v1 = vol[0:n, 0:n, :].contiguous()
    v2 = vol[sz:sz + n, sy:sy + n, :].transpose(1, 2).flip(1).contiguous()   # 90 degrees about z + shift
    del vol
    torch.cuda.synchronize()
    dets = [api.Detector(), api.Detector()]
    kps = [api.KeypointStore(), api.KeypointStore()]
    descs = [api.DescriptorStore(), api.DescriptorStore()]
    for dsc in descs:
        dsc.keep_device(True)          # the matcher reads the histograms where the describe kernel left them
    matcher = api.Matcher()
    want = np.array([[0, 1.0, 0, -sy], [-1.0, 0, 0, n - 1], [0, 0, 1.0, -sz]])
    res = {}

    def step():
        for v, det, kp, desc in zip((v1, v2), dets, kps, descs):
            assert det.detect_keypoints_device(v.data_ptr(), n, n, n, kp) == 0
            assert det.extract_descriptors(kp, desc) == 0
        t0 = time.perf_counter()
        m = matcher.match(descs[0], descs[1], 0.8)
        t1 = time.perf_counter()
        hit = np.nonzero(m >= 0)[0]
        p1 = descs[0].xyz()[hit]
        p2 = descs[1].xyz()[m[hit]]
        T, inl = api.ransac_affine(p1, p2, err_thresh=3.0, num_iter=500, seed=5)
        res.update(match_s=t1 - t0, ransac_s=time.perf_counter() - t1, matches=int(len(hit)),
                   inliers=int(inl.sum()), T=T, nn2_s=matcher.seconds())

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stat = []
    for _ in range(a.steps):
        step()
        stat.append((res["match_s"], res["ransac_s"], res["nn2_s"]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    na, nb = len(descs[0]), len(descs[1])
    nn2 = float(np.median([s[2] for s in stat]))
    flops = 2.0 * 2.0 * na * nb * 768                      # both directions, multiply + add
    peak = 157.3                                           # dense f32 MFMA TFLOP/s, MI355X_MICROARCH.md
    err_R = float(np.abs(res["T"][:, :3] - want[:, :3]).max())
    err_t = float(np.abs(res["T"][:, 3] - want[:, 3]).max())
    out = {
        "metric": "Mvoxel/s two-volume detect+describe+match+RANSAC (float32 volumes resident in HBM)",
        "value": round(2 * n ** 3 / 1e6 / dt, 2), "unit": "Mvoxel/s", "n_gpus": 1, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * dt, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "two %d^3 float32 lattice-blob volumes (the second = the first rotated by 90 "
                               "degrees about z and shifted): detect+describe both, mutual nearest-neighbour "
                               "match with ratio 0.8, RANSAC 12-parameter affine" % n,
                   "parallelism": "single GPU"},
        "keypoints": [na, nb], "matches": res["matches"], "inliers": res["inliers"],
        "affine_error": {"linear_part_max": round(err_R, 5), "translation_max_voxels": round(err_t, 4)},
        "stage_s": {"detect_describe": round(dt - float(np.median([s[0] + s[1] for s in stat])), 6),
                    "match": round(float(np.median([s[0] for s in stat])), 6),
                    "ransac": round(float(np.median([s[1] for s in stat])), 6)},
        "roofline": {"bound": "mfma", "kernel": "k_nn2 (v_mfma_f32_32x32x2_f32), both directions",
                     "achieved": round(flops / nn2 / 1e12, 2) if nn2 > 0 else None, "peak": peak,
                     "unit": "TFLOP/s", "frac": round(flops / nn2 / 1e12 / peak, 4) if nn2 > 0 else None,
                     "traffic": None, "flops": flops, "seconds": round(nn2, 6),
                     "note": "device seconds of the two sift3d_hip_nn2 calls (HIP events on their stream); "
                             "parity unpinned: the reference fork removed this stage"},
    }
    print(json.dumps(out))


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this script as child processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run sets them), relay
    their output, return non-zero if any rank fails.  The parent imports neither torch nor the library and
    makes no HIP call: nothing is exec'd from a GPU-initialised process."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    code = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    # a rank died: the others are inside a collective it will never join
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n"
                                     % (procs.index(p), code))
                    for q in pending:
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


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
    ap.add_argument("--rehearse-threads", type=int, default=0, metavar="R",
                    help="N=1 only: R slab drivers as R threads of this process on the one device "
                         "(sharded_c.StreamThreadTransport) -- exercises the N = R code path and geometry where "
                         "R processes may not share a card; the numbers are not a scaling measurement")
    ap.add_argument("--register", action="store_true",
                    help="BASELINE configs[4]: two volumes, detect+describe both, descriptor matching on "
                         "the matrix cores + RANSAC affine; prints the config-5 line (flops roofline of k_nn2)")
    ap.add_argument("--sharded", action="store_true",
                    help="N=1 only: run the Z-slab driver (C: sift3d_amd_sharded_*) instead of the "
                         "drop-in API, to measure the driver's own overhead")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started like the one-GPU run (`python bench.py --gpus N ...`), not under torch.distributed.run:
        # this process starts the N ranks itself and never touches the GPU
        raise SystemExit(launch_ranks(a.gpus))

    import torch
    import torch.distributed as dist
    from sift3d_amd import api, hip

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d, but WORLD_SIZE is %d" % (a.gpus, world))
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
        # Control plane (the RCCL unique id, the barriers around the timed region, the max over ranks of a
        # few host scalars): gloo.  The data plane -- halos, reductions, gathers -- is the library's OWN RCCL
        # communicator (sift3d_amd_rccl_transport); no second RCCL communicator exists in the process.
        dist.init_process_group("gloo")

    n = a.strong if a.strong else a.size
    nz_total = n if a.strong else n * world
    if a.register:
        return register_bench(a, torch, api, hip)
    rt = a.rehearse_threads if world == 1 else 0
    if rt > 1:
        # R ranks as R threads: R C slab drivers on this device, exchanging through ThreadTransport
        import threading
        from sift3d_amd import sharded_c
        nz_total = n if a.strong else n * rt
        grp = sharded_c.StreamThreadGroup(rt)      # stream-ordered exchanges (events only), as over RCCL
        jobs = [None] * rt

        def _threads(fn):
            err = []

            def run(r):
                try:
                    fn(r)
                except Exception as e:  # noqa: BLE001
                    err.append((r, repr(e)))
                    grp.barrier.abort()
            th = [threading.Thread(target=run, args=(r,)) for r in range(rt)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if err:
                raise SystemExit("rehearsal failed: %s" % err)

        def _make(r):
            jobs[r] = sharded_c.CShardedSift3D(n, n, nz_total, sharded_c.StreamThreadTransport(grp, r))
            jobs[r].synth(seed=11)
        _threads(_make)
        step = lambda: _threads(lambda r: jobs[r].step())  # noqa: E731
        stats = lambda: dict(jobs[0].stats(), rehearsal="%d ranks as threads of one process on one device: "  # noqa: E731
                             "code path and geometry of N = %d, not a scaling measurement" % (rt, rt),
                             per_rank_max_s={k: round(max(j.breakdown()[k] for j in jobs), 6)
                                             for k in jobs[0].breakdown()})
        pyr_time = jobs[0].pyramid_seconds
    elif world == 1 and not a.sharded:
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
        pyr_time = lambda: (lambda t: (t["gauss_dev"], t.get("yz_last", 0.0)))(det.timings())  # noqa: E731
    else:
        # The slab driver in C (sift3d_amd/csrc/sift3d_sharded.c) over the library's own RCCL
        # communicator; rehearsals on one device stage the exchanges through gloo.
        from sift3d_amd import sharded_c
        if world == 1:
            tr = None
        elif rehearse:
            tr = sharded_c.DistTransport()
        else:
            tr = sharded_c.RcclTransport()
        job = sharded_c.CShardedSift3D(n, n, nz_total, tr)
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
    pyr, yz_in_step = [], []
    for _ in range(a.steps):
        step()
        pt_ = pyr_time()
        if isinstance(pt_, tuple):
            pyr.append(pt_[0])
            yz_in_step.append(pt_[1])
        else:
            pyr.append(pt_)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    per_rank = None
    if world > 1 and hasattr(job, "breakdown"):
        # stage seconds of the last step, max over ranks (so that a scaling curve can be read)
        bd = job.breakdown()
        keys = sorted(bd)
        t = torch.tensor([bd[k] for k in keys], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank = {k: round(float(v), 6) for k, v in zip(keys, t.tolist())}
    def _finish():
        """Tear the ranks down in step: slab driver, the library's RCCL communicator, the process group."""
        if world > 1:
            job.close()
            dist.barrier()
            if tr is not None:
                tr.close()
            dist.destroy_process_group()

    if rank != 0:
        _finish()
        return

    ms_per_step = 1e3 * dt / a.steps
    value = voxels_per_step / 1e6 / (dt / a.steps)
    if rt > 1:
        slabs = " cut into %d Z-slabs" % rt
    elif world == 1:
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
                   "parallelism": ("z-slab x%d (threads of one process, one device)" % rt) if rt > 1 else
                                  "single GPU" if world == 1 else "z-slab x%d" % world},
    }
    out.update(stats())
    if per_rank:
        out["per_rank_max_s"] = per_rank
        out["transport"] = "gloo rehearsal (ranks share device 0)" if rehearse else "RCCL (ncclSend/Recv, all-reduce, all-gather)"
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
    if world == 1 and not a.sharded and not a.no_host and not rt:
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
        # (scalars: volume in pageable host memory, PCIe H2D inside the timed region --
        # sift3d_detect_keypoints on a sift3d_image; never `value`)
        out["value_host_resident"] = round(voxels_per_step / 1e6 / th, 2)
        out["ms_per_step_host_resident"] = round(1e3 * th, 3)
        del im
    if world == 1 and not a.no_micro and not a.sharded and not a.strong and not rt:
        kb = kernel_microbench(torch, hip, n)
        # dominant kernel = the pipeline kernel with the longest launch
        dom = max((k for k in kb if k["in_pipeline"]), key=lambda k: k["avg_ms"])
        tbytes = None
        if traffic:
            # profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950
            # corrections applied by profiles/pmc_fir.py) is keyed by the kernel symbol
            sym = dom["kernel"].split(" (")[0]
            if sym in traffic:
                tbytes = traffic[sym]["hbm_bytes"]
        # `achieved` / `frac` price the ALGORITHMIC bytes (8 B per voxel and 1-D pass)
        # against the HBM peak, with the launch duration measured IN THE STEP: HIP events around that
        # launch -- the last blur of octave 0 -- on the stream it runs on, mean over the timed steps;
        # there it shares the device with the streams that build octaves >= 1, which is also what the
        # rocprofv3 kernel trace of a --no-micro run averages (profiles/).  `frac_alone` is the same kernel
        # with the device to itself (the microbench leg).  `hbm_GBs` is what the kernel really moves
        # (counter traffic / launch time): the fused y+z kernel keeps its intermediate on chip.
        # (the MEAN over the timed steps: what rocprofv3's AverageNs of the same launches is; the launch shares
        # the device with the other octaves' streams and its duration scatters by tens of per cent -- the
        # median is reported beside it)
        yz_ms = 1e3 * float(np.mean(yz_in_step)) if yz_in_step and min(yz_in_step) > 0 else None
        launch_ms = yz_ms if yz_ms else dom["avg_ms"]
        ach = dom["algorithmic_GB"] / (launch_ms * 1e-3)
        out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(ach, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": tbytes,
                           "avg_launch_ms": round(launch_ms, 4),
                           "launch_timing": "in the step (mean of %d steps)" % len(yz_in_step) if yz_ms
                                            else "kernel alone (no in-step events)",
                           "median_launch_ms": round(1e3 * float(np.median(yz_in_step)), 4) if yz_ms else None,
                           "frac_alone": round(dom["achieved_GBs"] / HBM_PEAK_GBS, 4),
                           "avg_launch_ms_alone": dom["avg_ms"],
                           "algorithmic_bytes_per_launch": int(dom["algorithmic_GB"] * 1e9),
                           "hbm_GBs": round(tbytes / 1e9 / (launch_ms * 1e-3), 1) if tbytes else None,
                           "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                             "passes of this kernel, committed; not measured in this run)"}
        if pyramid:
            # the metric BASELINE.json names: achieved HBM GB/s on the whole Gaussian pyramid build
            out["roofline"].update(pyramid_frac=pyramid["frac"], pyramid_ms=round(1e3 * pyramid["seconds"], 4),
                                   pyramid_GBs=pyramid["achieved"],
                                   pyramid_algorithmic_GB=pyramid["algorithmic_GB"])
        details = {"kernels": kb, "pyramid": pyramid}
        dpath = os.path.join(ROOT, "profiles", "describe_model.json")
        if os.path.exists(dpath):
            try:
                dm = json.load(open(dpath))
                dsec = out.get("stage_s", {}).get("describe")
                if dsec and dm.get("valu_insts"):
                    # k_describe is bound by instruction issue (VALU and the LDS pipe), not by HBM: what the
                    # hardware counters of the committed profile say for this kernel (instruction and
                    # LDS-array-cycle counts are properties of the code and the workload; the seconds are
                    # this run's)
                    cu_cycles = dsec * 2.4e9 * 256
                    simd_rate = 256 * 4 * 2.4e9 / dm.get("cycles_per_valu", 2.5)
                    dm["seconds"] = dsec
                    dm["valu_frac"] = round(dm["valu_insts"] / dsec / simd_rate, 4)
                    dm["lds_array_frac"] = round(dm.get("lds_array_cycles", 0) / cu_cycles, 4)
                    if dm.get("wave_quad_cycles"):
                        w = dm["wave_quad_cycles"]
                        dm["wave_time_split"] = {"waiting_at_s_waitcnt": round(dm.get("wait_any", 0) / w, 3),
                                                 "issue_stalled": round(dm.get("wait_inst_any", 0) / w, 3),
                                                 "issue_stalled_on_lds": round(dm.get("wait_inst_lds", 0) / w, 3)}
                    dm["frac"] = max(dm["valu_frac"], dm["lds_array_frac"])
                    details["describe"] = dm
                    out["roofline"].update(describe_ms=round(1e3 * dsec, 3), describe_valu_frac=dm["valu_frac"],
                                           describe_lds_array_frac=dm["lds_array_frac"])
            except Exception:
                pass
        out["details"] = details
    else:
        out["roofline"] = dict(pyramid or {}, kernel="Gaussian pyramid (all k_fir_* launches)",
                               traffic=None)
        if pyramid:
            out["roofline"].update(pyramid_frac=pyramid["frac"], pyramid_ms=round(1e3 * pyramid["seconds"], 4),
                                   pyramid_GBs=pyramid["achieved"])
        if yz_in_step and min(yz_in_step) > 0:
            # (no microbench leg: the in-step launch of the dominant kernel, as the default run reports it)
            yz_ms = 1e3 * float(np.mean(yz_in_step))
            out["roofline"].update(dominant_kernel="k_fir_yz_dma<8, 32> (last blur of octave 0)",
                                   dominant_avg_launch_ms=round(yz_ms, 4),
                                   dominant_frac=round(16.0 * n ** 3 / 1e9 / (yz_ms * 1e-3) / HBM_PEAK_GBS, 4))
    if not a.no_cpu and world == 1:          # the CPU leg runs at N = 1 only
        out["cpu_baseline"] = cpu_baseline()
    if "details" in out:                     # the long per-kernel lists go last
        out["details"] = out.pop("details")
    print(json.dumps(out))
    sys.stdout.flush()
    _finish()


if __name__ == "__main__":
    main()
