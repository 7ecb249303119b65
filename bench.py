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


# candidate -> keypoint counts of the bench workloads (lattice volume, seed 11, default parameters), as the
# reference (512^3: tests/golden/g5_512.npz) and the single-GPU path / the 8-rank rehearsals of rounds 3-4
# (profiles/r04_rehearsal_8ranks_weak.json, r04_rehearsal_8ranks_1024.json) produced them.  Every run checks
# its own counts against these and exits non-zero on a mismatch: the first multi-GPU run verifies itself.
KNOWN_COUNTS = {
    ("weak", 512, 1): (158924, 42501), ("weak", 512, 2): (316228, 84538),
    ("weak", 512, 4): (619908, 166329), ("weak", 512, 8): (1222989, 328589),
}
KNOWN_STRONG = {1024: (1249357, 332413), 512: (158924, 42501)}


def expected_counts(strong, n, ranks):
    return KNOWN_STRONG.get(n) if strong else KNOWN_COUNTS.get(("weak", n, ranks))


def fir_kernel_names(api, n):
    """(x kernel, y+z kernel, half width) of every octave-0 blur, as the launchers choose them
    (sift3d_kernels.hip launch_fir_x_u1, sift3d_fir_yz.hip launch_fir_yz) -- blur 0 scales on the fly."""
    sig = [0.5387011637869722, 0.9732939207323564, 1.2262734984654078, 1.5450077936447955,
           1.9465878414647133, 2.4525469969308156]
    out = []
    for b, s in enumerate(sig):
        hw = len(api.gauss_filter(s)) // 2
        xk = ("k_fir_x_u1f<%d, %s>" % (hw, "true" if b == 0 else "false")) if n % 512 == 0 else "k_fir_x_u1<%d>" % hw
        if n % 64 == 0 and n >= 128:
            yk = "k_fir_yz_dma<%d, %d>" % (hw, 64 if hw <= 2 else 32)
        else:
            yk = "k_fir_yz_u1<%d, %s>" % (hw, "32, 32" if n % 128 == 0 else "32, 16")
        out.append((xk, yk, hw))
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
                    help="strong scaling as the MAIN line: ONE EDGE^3 volume (BASELINE configs[3]: 1024) cut "
                         "into --gpus Z-slabs, instead of the default 512 planes per GPU")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="skip the strong_1024 sub-record (BASELINE configs[3]: one 1024^3 volume as N Z-slabs) "
                         "that every default run carries beside its main line")
    ap.add_argument("--no-host", action="store_true", help="skip the host-resident (PCIe-inclusive) leg")
    ap.add_argument("--no-pyramid-leg", action="store_true",
                    help="skip the pyramid-only leg (a kernel trace of the run then holds in-step launches only)")
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

    if a.register:
        return register_bench(a, torch, api, hip)
    rt = a.rehearse_threads if world == 1 else 0
    ranks = rt if rt > 1 else world

    # one transport for every leg of the run (the library's RCCL communicator is opened once)
    tr = None
    if world > 1:
        from sift3d_amd import sharded_c
        tr = sharded_c.DistTransport() if rehearse else sharded_c.RcclTransport()

    def run_leg(n, nz_total, steps, warmup, use_api):
        """K timed steps of one workload (barrier + synchronize on both sides, max over ranks).  Returns a dict:
        seconds per step, stats, the pyramid's device seconds per step, octave 0's in-step launch timings."""
        leg = {}
        if rt > 1:
            import threading
            from sift3d_amd import sharded_c
            grp = sharded_c.StreamThreadGroup(rt)      # stream-ordered exchanges (events only), as over RCCL
            jobs = [None] * rt
            trs = [None] * rt

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
                trs[r] = sharded_c.StreamThreadTransport(grp, r)
                jobs[r] = sharded_c.CShardedSift3D(n, n, nz_total, trs[r])
                jobs[r].synth(seed=11)
            _threads(_make)
            step = lambda: _threads(lambda r: jobs[r].step())  # noqa: E731
            stats = lambda: dict(jobs[0].stats(), rehearsal="%d ranks as threads of one process on one device: "  # noqa: E731
                                 "code path and geometry of N = %d, not a scaling measurement" % (rt, rt),
                                 per_rank_max_s={k: round(max(j.breakdown()[k] for j in jobs), 6)
                                                 for k in jobs[0].breakdown()})
            pyr_time = jobs[0].pyramid_seconds
            closer = lambda: ([j.close() for j in jobs], [t.close() for t in trs], grp.close())  # noqa: E731
            job = None
        elif use_api:
            vol = torch.empty((nz_total, n, n), device="cuda")
            hip.synth_lattice(vol, 0, 11)
            torch.cuda.synchronize()
            det = api.Detector()
            kp, desc = api.KeypointStore(), api.DescriptorStore()

            def step():
                rc = det.detect_keypoints_device(vol.data_ptr(), n, n, nz_total, kp)
                assert rc == 0, "detect failed"
                rc = det.extract_descriptors(kp, desc)
                assert rc == 0, "describe failed"

            stats = lambda: dict(candidates=det.num_candidates(), keypoints=len(kp),  # noqa: E731
                                 stage_s={k: round(v, 6) for k, v in det.timings().items()})
            pyr_time = lambda: det.timings()["gauss_dev"]  # noqa: E731
            leg.update(det=det, kp=kp, desc=desc, vol=vol)
            closer = lambda: None  # noqa: E731
            job = None
        else:
            # The slab driver in C (sift3d_amd/csrc/sift3d_sharded.c) over the library's own RCCL
            # communicator; rehearsals on one device stage the exchanges through gloo.
            from sift3d_amd import sharded_c
            job = sharded_c.CShardedSift3D(n, n, nz_total, tr)
            job.synth(seed=11)
            step = job.step
            stats = job.stats
            pyr_time = job.pyramid_seconds
            closer = job.close
        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pyr, launches, dclk = [], [], []
        for _ in range(steps):
            step()
            pyr.append(pyr_time())
            if "det" in leg:
                launches.append(leg["det"].launch_timings())
                dclk.append(leg["det"].describe_clock())
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per_rank = None
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            if job is not None:
                # stage seconds of the last step, max over ranks (so that a scaling curve can be read)
                bd = job.breakdown()
                keys = sorted(bd)
                t = torch.tensor([bd[k] for k in keys] + [float(np.median(pyr))], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                per_rank = {k: round(float(v), 6) for k, v in zip(keys, t.tolist())}
                pyr = [float(t[-1])]                 # the slowest rank's pyramid
        leg.update(sec_per_step=dt / steps, stats=stats(), pyr=pyr, launches=launches, dclk=dclk,
                   per_rank=per_rank, close=closer, voxels=n * n * nz_total)
        return leg

    def pyramid_record(n, nz_rank, seconds):
        pbytes = pyramid_algorithmic_bytes(n, n, nz_rank)
        ach = pbytes / 1e9 / seconds
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_GB": round(pbytes / 1e9, 3),
                "seconds": round(seconds, 6)}

    n = a.strong if a.strong else a.size
    nz_total = n if a.strong else n * ranks
    use_api = world == 1 and not a.sharded and not rt
    main_leg = run_leg(n, nz_total, a.steps, a.warmup, use_api)
    failed = []

    def check(tag, strong, edge, st):
        want = expected_counts(strong, edge, ranks)
        got = (int(st["candidates"]), int(st["keypoints"]))
        if want is None:
            return None
        if got != want:
            failed.append("%s: %d candidates -> %d keypoints, expected %d -> %d" % ((tag,) + got + want))
        return got == want

    main_ok = check("main line", bool(a.strong), n, main_leg["stats"])
    if not use_api:
        main_leg["close"]()

    # BASELINE configs[3] beside the main line: ONE 1024^3 volume as N Z-slabs, so that the driver's
    # N = 1, 2, 4, 8 records hold the strong-scaling curve north_star names
    strong_rec = None
    if not a.strong and not a.no_strong_leg and a.size == 512:
        sl = run_leg(1024, 1024, min(a.steps, 3), 1, use_api)
        st = sl["stats"]
        pt = float(np.median(sl["pyr"])) if sl["pyr"] and sl["pyr"][0] else None
        strong_rec = {"workload": "1024x1024x1024 float32 lattice-blob volume%s, detect+describe, default parameters"
                                  % ("" if ranks == 1 else " cut into %d Z-slabs" % ranks),
                      "scaling": "strong", "n_gpus": world, "ranks": ranks, "steps": min(a.steps, 3), "warmup": 1,
                      "ms_per_step": round(1e3 * sl["sec_per_step"], 3),
                      "value": round(sl["voxels"] / 1e6 / sl["sec_per_step"], 2), "unit": "Mvoxel/s",
                      "candidates": int(st["candidates"]), "keypoints": int(st["keypoints"]),
                      "counts_ok": check("strong_1024", True, 1024, st)}
        if pt:
            pr = pyramid_record(1024, 1024 // ranks, pt)
            strong_rec.update(pyramid_ms_per_gpu=round(1e3 * pt, 4), pyramid_frac_per_gpu=pr["frac"],
                              pyramid_GBs_per_gpu=pr["achieved"])
        if sl["per_rank"]:
            strong_rec["per_rank_max_s"] = sl["per_rank"]
        elif "per_rank_max_s" in st:
            strong_rec["per_rank_max_s"] = st["per_rank_max_s"]
        elif "stage_s" in st:
            strong_rec["stage_s"] = st["stage_s"]
        sl["close"]()
        for k in ("det", "kp", "desc", "vol"):
            sl.pop(k, None)
        del sl
        torch.cuda.empty_cache()

    def _finish():
        """Tear the ranks down in step: the library's RCCL communicator, the process group."""
        if world > 1:
            dist.barrier()
            if tr is not None:
                tr.close()
            dist.destroy_process_group()

    if rank != 0:
        _finish()
        if failed:
            raise SystemExit(3)
        return

    dt = main_leg["sec_per_step"]
    voxels_per_step = main_leg["voxels"]
    ms_per_step = 1e3 * dt
    value = voxels_per_step / 1e6 / dt
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
    out.update(main_leg["stats"])
    out["counts_ok"] = main_ok
    if main_leg["per_rank"]:
        out["per_rank_max_s"] = main_leg["per_rank"]
        out["transport"] = "gloo rehearsal (ranks share device 0)" if rehearse else "RCCL (ncclSend/Recv, all-reduce, all-gather)"
    if strong_rec:
        out["strong_1024"] = strong_rec
    # pyramid roofline (whole Gaussian pyramid build, per GPU), ALGORITHMIC separable bytes
    pyr = main_leg["pyr"]
    pt = float(np.median(pyr)) if pyr and pyr[0] else None
    pyramid = pyramid_record(n, nz_total // ranks, pt) if pt else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = None
    if use_api:
        det, kp, vol = main_leg["det"], main_leg["kp"], main_leg["vol"]
    # host-resident entry (sift3d_detect_keypoints on a sift3d_make_image volume in host memory): the same K
    # steps timed the same way, H2D copy included.  Reported beside `value`, never as `value`.
    if use_api and not a.no_host:
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
        # SURVEY.md 8d's metric (1) taken literally: host-resident float32 input -> host-resident stores,
        # the PCIe H2D copy inside the timed region (sift3d_detect_keypoints on a sift3d_image); never `value`
        out["value_host_resident"] = round(voxels_per_step / 1e6 / th, 2)
        out["ms_per_step_host_resident"] = round(1e3 * th, 3)
        out["config"]["value_host_resident_Mvoxel_s"] = out["value_host_resident"]
        out["config"]["ms_per_step_host_resident"] = out["ms_per_step_host_resident"]
        del im
    # pyramid-only leg: the Gaussian pyramid with the device to itself (inside the step its last launches
    # share the device with octave 0's extrema sweep, which starts as soon as octave 0 is complete)
    pyr_alone = None
    if use_api and not a.no_pyramid_leg:
        ts = []
        for i in range(a.steps + 1):
            assert det.build_pyramid_device(vol.data_ptr(), n, n, nz_total) == 0
            if i:
                ts.append(det.timings()["gauss_dev"])
        pyr_alone = pyramid_record(n, nz_total, float(np.median(ts)))
    if use_api and not a.strong:
        kb = kernel_microbench(torch, hip, n) if not a.no_micro else []
        alone = {k["kernel"]: k for k in kb if k["in_pipeline"]}
        # every octave-0 pyramid launch of the timed steps, HIP events around each on the stream it runs on
        # (sift3d_amd_timings [10..]): mean / median over the steps.  roofline.kernel = the LONGEST of them.
        names = fir_kernel_names(api, n)
        L = np.asarray(main_leg["launches"], np.float64)            # [steps, blurs, (x, yz)]
        in_step = []
        for b, (xk, yk, hw) in enumerate(names):
            for j, (kn, nb) in enumerate(((xk, 8.0), (yk, 16.0))):
                t = L[:, b, j]
                if not (t > 0).all():
                    continue
                mean_ms, med_ms = 1e3 * float(t.mean()), 1e3 * float(np.median(t))
                gb = nb * n ** 3 / 1e9
                tb = traffic.get(kn.replace("true", "false"), {}).get("hbm_bytes") if traffic else None
                al = alone.get(kn.replace("true", "false"))
                in_step.append(dict(kernel=kn, blur=b, taps=2 * hw + 1, mean_ms=round(mean_ms, 4),
                                    median_ms=round(med_ms, 4), min_ms=round(1e3 * float(t.min()), 4),
                                    max_ms=round(1e3 * float(t.max()), 4), algorithmic_GB=round(gb, 4),
                                    frac=round(gb / (mean_ms * 1e-3) / HBM_PEAK_GBS, 4),
                                    alone_ms=al["avg_ms"] if al else None,
                                    frac_alone=al["frac"] if al else None,
                                    hbm_bytes=tb,
                                    hbm_frac=round(tb / 1e9 / (mean_ms * 1e-3) / HBM_PEAK_GBS, 4) if tb else None))
        if in_step:
            dom = max(in_step, key=lambda k: k["mean_ms"])
            out["roofline"] = {
                "bound": "hbm", "kernel": "%s (blur %d of octave 0)" % (dom["kernel"], dom["blur"]),
                "achieved": round(dom["algorithmic_GB"] / (dom["mean_ms"] * 1e-3), 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": dom["frac"], "traffic": dom["hbm_bytes"],
                "avg_launch_ms": dom["mean_ms"], "median_launch_ms": dom["median_ms"],
                "launch_timing": "the LONGEST octave-0 pyramid launch in the step: HIP events around every x and "
                                 "fused y+z launch on its stream, mean over the %d timed steps (what rocprofv3's "
                                 "average of the same launches is)" % len(L),
                "frac_alone": dom["frac_alone"], "avg_launch_ms_alone": dom["alone_ms"],
                "algorithmic_bytes_per_launch": int(dom["algorithmic_GB"] * 1e9),
                "hbm_GBs": round(dom["hbm_bytes"] / 1e9 / (dom["mean_ms"] * 1e-3), 1) if dom["hbm_bytes"] else None,
                "hbm_frac": dom["hbm_frac"],
                "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "kernel, committed; not measured in this run)"}
        else:
            out["roofline"] = dict(pyramid or {}, kernel="Gaussian pyramid (all k_fir_* launches)", traffic=None)
        details = {"kernels": kb, "in_step_launches": in_step, "pyramid": pyramid, "pyramid_alone": pyr_alone}
        dpath = os.path.join(ROOT, "profiles", "describe_model.json")
        if os.path.exists(dpath):
            try:
                dm = json.load(open(dpath))
                dsec = out.get("stage_s", {}).get("describe")
                clk = [c for c in main_leg["dclk"] if c]
                if dsec and dm.get("valu_insts") and clk:
                    # k_describe is bound by instruction issue (VALU and the LDS pipe), not by HBM.  Clock-free:
                    # the kernel's own shader-cycle count (its first, persistent wave: s_memtime at entry and
                    # exit, measured in THIS run) against the committed counter totals per launch (instruction
                    # and LDS-array-cycle counts are properties of the code and the workload).  One wave64 VALU
                    # instruction occupies a SIMD for 2 cycles (MI355X_MICROARCH.md: SIMD-32).
                    cyc = float(np.median([c[0] for c in clk]))
                    sec = float(np.median([c[1] for c in clk]))
                    cus, simds = 256, 1024
                    dm = {k: v for k, v in dm.items() if k not in ("lds_issue_model", "issue_model_note")}
                    dm.update(seconds=dsec, kernel_cycles=cyc, kernel_seconds_100MHz_counter=round(sec, 6),
                              clock_GHz=round(cyc / sec / 1e9, 3),
                              valu_busy=round(dm["valu_insts"] * 2.0 / (simds * cyc), 4),
                              lds_array_busy=round(dm.get("lds_array_cycles", 0) / (cus * cyc), 4))
                    if dm.get("wave_quad_cycles") and dm.get("waves"):
                        w = dm["wave_quad_cycles"]
                        dm["profiled_kernel_cycles"] = round(4.0 * w / dm["waves"])
                        dm["wave_time_split"] = {"waiting_at_s_waitcnt": round(dm.get("wait_any", 0) / w, 3),
                                                 "issue_stalled": round(dm.get("wait_inst_any", 0) / w, 3),
                                                 "issue_stalled_on_lds": round(dm.get("wait_inst_lds", 0) / w, 3)}
                    dm["frac"] = max(dm["valu_busy"], dm["lds_array_busy"])
                    details["describe"] = dm
                    out["roofline"].update(describe_ms=round(1e3 * dsec, 3), describe_clock_GHz=dm["clock_GHz"],
                                           describe_valu_busy=dm["valu_busy"],
                                           describe_lds_array_busy=dm["lds_array_busy"])
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("bench.py: describe model skipped (%r)\n" % (e,))
        out["details"] = details
    else:
        out["roofline"] = dict(pyramid or {}, kernel="Gaussian pyramid (all k_fir_* launches)", traffic=None)
    if pyramid:
        # the metric BASELINE.json names: achieved HBM GB/s on the whole Gaussian pyramid build.
        #   pyramid_frac      ALGORITHMIC separable bytes (24 B per voxel and blur) / in-step time
        #   pyramid_hbm_frac  bytes that really reach HBM (rocprofv3 FETCH_SIZE / WRITE_SIZE of one whole pyramid
        #                     build, profiles/traffic.json "pyramid_build") / the same time: the fused y+z pass
        #                     keeps its intermediate on chip, so this is the smaller number
        out["roofline"].update(pyramid_frac=pyramid["frac"], pyramid_ms=round(1e3 * pyramid["seconds"], 4),
                               pyramid_GBs=pyramid["achieved"], pyramid_algorithmic_GB=pyramid["algorithmic_GB"])
        pb = (traffic or {}).get("pyramid_build_%d" % n) if ranks == 1 and nz_total == n else None
        if pb:
            out["roofline"].update(pyramid_hbm_GB=round(pb["hbm_bytes"] / 1e9, 3),
                                   pyramid_hbm_frac=round(pb["hbm_bytes"] / 1e9 / pyramid["seconds"] / HBM_PEAK_GBS, 4))
        if pyr_alone:
            out["roofline"].update(pyramid_alone_ms=round(1e3 * pyr_alone["seconds"], 4),
                                   pyramid_alone_frac=pyr_alone["frac"])
            if pb:
                out["roofline"]["pyramid_alone_hbm_frac"] = round(
                    pb["hbm_bytes"] / 1e9 / pyr_alone["seconds"] / HBM_PEAK_GBS, 4)
    if not a.no_cpu and world == 1:          # the CPU leg runs at N = 1 only
        out["cpu_baseline"] = cpu_baseline()
    if "details" in out:                     # the long per-kernel lists go last
        out["details"] = out.pop("details")
    if failed:
        out["counts_error"] = failed
    print(json.dumps(out))
    sys.stdout.flush()
    _finish()
    if failed:
        sys.stderr.write("bench.py: WRONG RESULT: %s\n" % "; ".join(failed))
        raise SystemExit(3)


if __name__ == "__main__":
    main()
