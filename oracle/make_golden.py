#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference -- TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs oracle/_ref/, i.e. /root/reference):

    make -C oracle ref && OMP_NUM_THREADS=1 python oracle/make_golden.py [--big]

Inputs are produced by this repository's own seeded generators (sift3d_amd/csrc/synth.c)
or numpy's PCG64, so only OUTPUTS of the reference are stored.  The fixtures pin the CPU
restatement (oracle/sift3d_oracle.c) and, through it and directly, the HIP path.

Fixture families:
  g1_filters   Gaussian taps for the default bank and a sweep of sigmas
  g2_fir       1-D interpolating FIR per axis, three widths x units 1/2/4 (+ anisotropic)
  g3_*         end-to-end dumps: level digests, small levels in full, dogmax-free
               candidate list, keypoints (R, sd, stale strength), descriptors, sort order
  g5_*         larger volumes: candidate/keypoint lists, R, descriptor subset + row sums
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import refprobe  # noqa: E402
from oracle import sift3d_oracle as so  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def desc_projection(h):
    """Every histogram row projected on 8 seeded random unit vectors (N x 8 float64).  Pins ALL
    rows of a large fixture -- which bin the mass of a row sits in, not only its sum
    (reference binning: sift3d/sift.c:1355-1373, 1514-1526) -- in 64 bytes per row.  The
    columns have unit 2-norm, so an elementwise relative error e of the row moves a projection
    by at most e * |row|_2."""
    P = np.random.default_rng(20240768).standard_normal((768, 8))
    P /= np.linalg.norm(P, axis=0, keepdims=True)
    return np.ascontiguousarray(h, np.float64) @ P


def digest(a):
    """sha1 over float32 values with -0.0 folded into +0.0."""
    a = np.ascontiguousarray(a, np.float32) + np.float32(0.0)
    return hashlib.sha1(a.tobytes()).hexdigest()


def g1_filters():
    sigmas = [0.3, 0.5387011637869722, 0.8, 0.9732939207323564, 1.0, 1.2262734984654078,
              1.5450077936447955, 1.9465878414647133, 2.0, 2.4525469969308156, 3.3, 5.0,
              0.0]
    d = {"sigmas": np.array(sigmas)}
    for i, s in enumerate(sigmas):
        d["taps_%d" % i] = refprobe.gauss_filter(s)
    p = refprobe.Probe()
    vol = so.synth_survey(16, nblob=4)
    assert p.detect(vol) == 0
    bank = p.gss()
    d["bank_sigma"] = np.array([s for s, _ in bank])
    for i, (_, t) in enumerate(bank):
        d["bank_%d" % i] = t
    p.close()
    np.savez_compressed(os.path.join(OUT, "g1_filters.npz"), **d)
    return {"g1_filters": len(sigmas)}


def g2_fir():
    rng = np.random.Generator(np.random.PCG64(20240917))
    dims = (21, 13, 9)  # nx, ny, nz ; nz = 9 with the 17-tap filter is the hw ~ n corner
    vol = rng.standard_normal((dims[2], dims[1], dims[0])).astype(np.float32)
    d = {"vol": vol}
    widths = {5: 0.5387011637869722, 9: 1.2262734984654078, 17: 2.4525469969308156}
    cases = []
    for w, sg in widths.items():
        taps = refprobe.gauss_filter(sg)
        assert len(taps) == w
        d["taps_w%d" % w] = taps
        for u in (1.0, 2.0, 4.0):
            for ax in range(3):
                key = "axis%d_w%d_u%g" % (ax, w, u)
                d[key] = refprobe.fir_axis(vol, taps, ax, units=(u, u, u), unit=1.0)
                cases.append(key)
    # asymmetric (non-Gaussian) taps: catches a reversed tap order
    at = rng.standard_normal(7).astype(np.float32)
    d["taps_asym"] = at
    for u in (1.0, 2.0):
        for ax in range(3):
            key = "axis%d_asym_u%g" % (ax, u)
            d[key] = refprobe.fir_axis(vol, at, ax, units=(u, u, u), unit=1.0)
            cases.append(key)
    # full 3-axis application, incl. anisotropic (non-dyadic unit factors) and unit=-1
    vol2 = rng.standard_normal((17, 20, 24)).astype(np.float32)
    d["vol2"] = vol2
    for name, units, unit in (("iso1", (1, 1, 1), 1.0), ("iso2", (2, 2, 2), 1.0),
                              ("iso4", (4, 4, 4), 1.0), ("aniso", (1.0, 1.5, 0.7), 1.0),
                              ("aniso3", (3.0, 0.9, 6.0), 1.0), ("default", (2, 2, 2), -1.0)):
        for w in (5, 17):
            key = "blur_%s_w%d" % (name, w)
            d[key] = refprobe.apply_sep_fir(vol2, d["taps_w%d" % w], units=units, unit=unit)
            cases.append(key)
    d["units_aniso"] = np.array([1.0, 1.5, 0.7])
    d["units_aniso3"] = np.array([3.0, 0.9, 6.0])
    d["down"] = refprobe.downsample(vol2)
    d["down_odd"] = refprobe.downsample(vol)
    # eigen: a few SPD matrices
    As, Qs, Ls = [], [], []
    for _ in range(16):
        m = rng.standard_normal((3, 3))
        A = m @ m.T
        Q, L = refprobe.eigen3(A)
        As.append(A)
        Qs.append(Q)
        Ls.append(L)
    d["eig_A"] = np.array(As)
    d["eig_Q"] = np.array(Qs)
    d["eig_L"] = np.array(Ls)
    np.savez_compressed(os.path.join(OUT, "g2_fir.npz"), **d)
    return {"g2_fir": len(cases)}


def end_to_end(name, vol, units=(1, 1, 1), full_levels_below=17, desc_stride=1, params=None,
               store_levels=True, input_spec=None):
    params = params or {}
    p = refprobe.Probe(**params)
    t0 = time.time()
    assert p.detect(vol, units) == 0
    t_detect = time.time() - t0
    K = params.get("num_kp_levels") or 3
    d = {"units": np.array(units, np.float64), "dims": np.array(vol.shape[::-1], np.int32),
         "input_digest": np.array(digest(vol)), "num_octaves": np.array(p.num_octaves)}
    for k, v in params.items():
        d["param_" + k] = np.array(v)
    dig = {}
    for o in range(p.num_octaves):
        for which, n in ((0, K + 3), (1, K + 2)):
            for s in range(-1, n - 1):
                a, u, sc = p.level(which, o, s)
                key = "%s_o%d_s%d" % ("G" if which == 0 else "D", o, s)
                dig[key] = digest(a)
                d["scale_" + key] = np.array(sc)
                d["lunits_" + key] = u
                if store_levels and max(a.shape) < full_levels_below:
                    d["level_" + key] = a
    a, _, _ = p.level(2, 0, 0)
    dig["IM"] = digest(a)
    d["digests"] = np.array(json.dumps(dig))
    c = p.candidates()
    d["cand_osxyz"] = c["osxyz"]
    d["cand_strength"] = c["strength"]
    d["cand_sd"] = c["sd"]
    k = p.keypoints()
    d["kp_os"] = k["os"]
    d["kp_xyzsd"] = k["xyzsd"]
    d["kp_strength"] = k["strength"]
    d["kp_R"] = k["R"]
    t_desc = 0.0
    if len(k["strength"]):
        d["kp_mat"] = p.kp_mat()
        t0 = time.time()
        assert p.describe() == 0
        t_desc = time.time() - t0
        h, x = p.descriptors()
        d["desc_idx"] = np.arange(0, len(h), desc_stride, dtype=np.int32)
        d["desc_hist"] = h[::desc_stride]
        d["desc_xyzsd"] = x
        d["desc_rowsum"] = h.astype(np.float64).sum(axis=1)
        d["desc_rowsumsq"] = (h.astype(np.float64) ** 2).sum(axis=1)
        d["desc_proj"] = desc_projection(h)
        # sort_by_strength(limit): resulting order expressed as (o,s,x,y,z,strength) rows
        for lim in (0, 10):
            q = refprobe.Probe(**params)
            assert q.detect(vol, units) == 0
            q.sort(lim)
            kk = q.keypoints()
            d["sort%d_os" % lim] = kk["os"]
            d["sort%d_xyzsd" % lim] = kk["xyzsd"]
            d["sort%d_strength" % lim] = kk["strength"]
            q.close()
    if input_spec:
        d["input_spec"] = np.array(json.dumps(input_spec))
    d["ref_time_detect_s"] = np.array(t_detect)
    d["ref_time_describe_s"] = np.array(t_desc)
    p.close()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    return {name: dict(cand=int(len(c["sd"])), kp=int(len(k["strength"])),
                       t_detect=round(t_detect, 2), t_describe=round(t_desc, 2))}


def end_to_end_digest(name, vol, stride=97, input_spec=None):
    """Full-size case: the reference's results as sha1 digests plus strided samples (small file)."""
    p = refprobe.Probe()
    t0 = time.time()
    assert p.detect(vol, (1, 1, 1)) == 0
    t_detect = time.time() - t0
    d = {"dims": np.array(vol.shape[::-1], np.int32), "input_digest": np.array(digest(vol)),
         "num_octaves": np.array(p.num_octaves), "stride": np.array(stride)}
    dig = {}
    for o in range(p.num_octaves):
        for which, n in ((0, 6), (1, 5)):
            for s in range(-1, n - 1):
                a, _, _ = p.level(which, o, s)
                dig["%s_o%d_s%d" % ("G" if which == 0 else "D", o, s)] = digest(a)
    c = p.candidates()
    k = p.keypoints()
    d["ncand"] = np.array(len(c["sd"]))
    d["nkp"] = np.array(len(k["strength"]))
    dig["cand_osxyz"] = digest(np.ascontiguousarray(c["osxyz"]))
    dig["kp_os"] = digest(np.ascontiguousarray(k["os"]))
    dig["kp_xyzsd"] = digest(np.ascontiguousarray(k["xyzsd"]))
    dig["kp_strength"] = digest(np.ascontiguousarray(k["strength"]))
    dig["kp_R"] = digest(np.ascontiguousarray(k["R"]))
    d["kp_idx"] = np.arange(0, len(k["strength"]), stride, dtype=np.int32)
    d["kp_os_s"] = k["os"][::stride]
    d["kp_xyzsd_s"] = k["xyzsd"][::stride]
    d["kp_strength_s"] = k["strength"][::stride]
    d["kp_R_s"] = k["R"][::stride]
    t0 = time.time()
    assert p.describe() == 0
    t_desc = time.time() - t0
    h, x = p.descriptors()
    dig["desc_hist"] = digest(np.ascontiguousarray(h))
    d["desc_hist_s"] = h[::stride]
    d["desc_rowsum"] = h.astype(np.float64).sum(axis=1).astype(np.float32)
    d["desc_proj"] = desc_projection(h)
    d["desc_rownorm"] = np.sqrt((h.astype(np.float64) ** 2).sum(axis=1))
    d["digests"] = np.array(json.dumps(dig))
    if input_spec:
        d["input_spec"] = np.array(json.dumps(input_spec))
    d["ref_time_detect_s"] = np.array(t_detect)
    d["ref_time_describe_s"] = np.array(t_desc)
    p.close()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    return {name: dict(cand=int(d["ncand"]), kp=int(d["nkp"]), t_detect=round(t_detect, 2),
                       t_describe=round(t_desc, 2))}


def g4_csv():
    """Files written by the reference's own sift3d_keypoint_store_save /
    sift3d_descriptor_store_save (sift.c:1741-1830 via write_Mat_rm, imutil.c:405-479), plain and
    gzip, for the g3_64 volume -- as DATA: the whole keypoint file (a few KB of numbers), sha1 +
    size + first / last rows of the descriptor file, and the records they were written from."""
    import gzip
    import hashlib
    import tempfile
    vol = so.synth_survey(64)
    p = refprobe.Probe()
    assert p.detect(vol) == 0 and p.describe() == 0
    k = p.keypoints()
    h, x = p.descriptors()
    d = {"kp_os": k["os"], "kp_xyzsd": k["xyzsd"], "kp_strength": k["strength"], "kp_R": k["R"],
         "desc_hist": h, "desc_xyzsd": x, "dims": np.array(vol.shape[::-1], np.int32)}
    with tempfile.TemporaryDirectory() as t:
        kp, dp = os.path.join(t, "kp.csv"), os.path.join(t, "desc.csv")
        assert p.save(kp, dp) == 0
        assert p.save(kp + ".gz", dp + ".gz") == 0
        ktxt, dtxt = open(kp, "rb").read(), open(dp, "rb").read()
        assert gzip.open(kp + ".gz", "rb").read() == ktxt and gzip.open(dp + ".gz", "rb").read() == dtxt
    rows = dtxt.split(b"\n")
    d["kp_csv"] = np.frombuffer(ktxt, np.uint8)
    d["desc_csv_sha1"] = np.array(hashlib.sha1(dtxt).hexdigest())
    d["desc_csv_bytes"] = np.array(len(dtxt))
    d["desc_csv_first_row"] = np.frombuffer(rows[0], np.uint8)
    d["desc_csv_last_row"] = np.frombuffer(rows[-2] if rows[-1] == b"" else rows[-1], np.uint8)
    d["desc_csv_ends_with_newline"] = np.array(dtxt.endswith(b"\n"))
    p.close()
    np.savez_compressed(os.path.join(OUT, "g4_csv.npz"), **d)
    return {"g4_csv": dict(kp=int(len(k["strength"])), kp_csv_bytes=len(ktxt), desc_csv_bytes=len(dtxt))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also 128^3 and 256^3 (minutes)")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    assert refprobe.available(), "run `make -C oracle ref` first (build container only)"
    os.makedirs(OUT, exist_ok=True)
    info = {"omp_threads": os.environ.get("OMP_NUM_THREADS", "unset"),
            "reference": "fatimp/SIFT3D v2.0 @ /root/reference, gcc -O3 -DNDEBUG -fopenmp"}
    jobs = {
        "g1": g1_filters,
        "g2": g2_fir,
        "g3_64": lambda: end_to_end("g3_64", so.synth_survey(64),
                                    input_spec=dict(gen="survey", n=64, nblob=200)),
        "g3_nc": lambda: end_to_end("g3_70x50x41", so.synth_survey((70, 50, 41)),
                                    input_spec=dict(gen="survey", n=[70, 50, 41])),
        "g3_an": lambda: end_to_end("g3_aniso", so.synth_survey((40, 33, 47)),
                                    units=(1.0, 1.5, 0.7),
                                    input_spec=dict(gen="survey", n=[40, 33, 47])),
        "g3_pp": lambda: end_to_end("g3_params", so.synth_survey(32),
                                    params=dict(num_kp_levels=2, sigma0=2.0, sigma_n=1.0,
                                                peak_thresh=0.05, corner_thresh=0.3),
                                    input_spec=dict(gen="survey", n=32)),
        "g3_cub": lambda: end_to_end("g3_cuboid64", so.synth_survey(64),
                                     params=dict(cuboid_extrema=True),
                                     input_spec=dict(gen="survey", n=64, nblob=200)),
        "g3_cubp": lambda: end_to_end("g3_cuboid_params", so.synth_lattice((50, 44, 40), seed=9),
                                      params=dict(cuboid_extrema=True, peak_thresh=0.03,
                                                  corner_thresh=0.2),
                                      input_spec=dict(gen="lattice", n=[50, 44, 40], seed=9)),
        "g3_lat": lambda: end_to_end("g3_lattice48", so.synth_lattice(48, seed=7),
                                     input_spec=dict(gen="lattice", n=48, seed=7)),
        "g4_csv": g4_csv,
    }
    if a.big:
        jobs["g5_128"] = lambda: end_to_end(
            "g5_128", so.synth_survey(128), desc_stride=8, store_levels=False,
            input_spec=dict(gen="survey", n=128, nblob=1600))
        jobs["g5_256"] = lambda: end_to_end(
            "g5_256", so.synth_survey(256), desc_stride=64, store_levels=False,
            input_spec=dict(gen="survey", n=256, nblob=12800))
    # descriptors at large sigma0 (sift.c:1453-1456: the window grows with sd^3, a bin receives ~30x
    # the terms it gets at the default 1.6) -- only on request (the reference needs minutes)
    if a.only and "g3_sigma3" in a.only.split(","):
        jobs["g3_sigma3"] = lambda: end_to_end(
            "g3_sigma3", so.synth_lattice(96, seed=13), store_levels=False,
            params=dict(sigma0=3.0, peak_thresh=0.03, corner_thresh=0.3),
            input_spec=dict(gen="lattice", n=96, seed=13))
    if a.only and "g3_sigma5" in a.only.split(","):
        jobs["g3_sigma5"] = lambda: end_to_end(
            "g3_sigma5", so.synth_survey(96), store_levels=False,
            params=dict(sigma0=5.0, peak_thresh=0.02, corner_thresh=0.2),
            input_spec=dict(gen="survey", n=96))
    # Round 5: the descriptor kernel's fast / reference-order switch (sift3d_host.c exact_desc_first_level:
    # windows of more than 1.9e5 voxels take the reference-order kernel).  sigma0 = 2.85: the keypoints of
    # level s = 0 have windows of 1.85e5 voxels -- the LARGEST the fast commit is ever used for --, those of
    # s = 1, 2 take the reference-order kernel.  Anisotropic units (1, 0.8, 0.8) at the default sigma0: the
    # unit product 0.64 pushes the windows of s = 2 (1.31e5 / 0.64 = 2.05e5 voxels) over the switch, s = 0, 1
    # stay below (sift.c:1453-1456, window radius in voxels = rad / unit: sift.c:96-108)
    if a.only and "g3_switch" in a.only.split(","):
        jobs["g3_switch"] = lambda: end_to_end(
            "g3_switch285", so.synth_lattice(96, seed=17), store_levels=False,
            params=dict(sigma0=2.85, peak_thresh=0.03, corner_thresh=0.3),
            input_spec=dict(gen="lattice", n=96, seed=17))
    if a.only and "g3_switch_aniso" in a.only.split(","):
        jobs["g3_switch_aniso"] = lambda: end_to_end(
            "g3_switch_aniso", so.synth_lattice((80, 112, 112), seed=19), units=(1.0, 0.8, 0.8),
            store_levels=False, params=dict(peak_thresh=0.05, corner_thresh=0.3),
            input_spec=dict(gen="lattice", n=[80, 112, 112], seed=19))
    # Round 5: sigma0 = 8 -- the octave filters are 31 ... 75 taps wide (half width ceil(3 sigma),
    # imutil.c:1275-1277), beyond the 65-tap tables of the device's fast kernels: the reference accepts any
    # sigma0 >= 0 (sift.c:553-565) and so must the drop-in
    if a.only and "g3_sigma8" in a.only.split(","):
        jobs["g3_sigma8"] = lambda: end_to_end(
            "g3_sigma8", so.synth_survey(96), store_levels=False,
            params=dict(sigma0=8.0, peak_thresh=0.01, corner_thresh=0.1),
            input_spec=dict(gen="survey", n=96))
    if a.only and "g5_slab8" in a.only.split(","):
        # BASELINE configs[3]'s slab geometry at 1/16 of its voxels: 256 x 256 x 1024 (eight 128-plane
        # Z-slabs, o_shard = 2) -- the sharded GPU tests compare with THIS, not with the single-GPU API
        jobs["g5_slab8"] = lambda: end_to_end_digest(
            "g5_256x256x1024", so.synth_lattice((256, 256, 1024), seed=11), stride=53,
            input_spec=dict(gen="lattice", n=[256, 256, 1024], seed=11))
    if a.only and "g5_512" in a.only.split(","):
        # BASELINE configs[2] (the bench workload): ~10 min of reference CPU time, ~10 GB
        jobs["g5_512"] = lambda: end_to_end_digest(
            "g5_512", so.synth_lattice(512, seed=11), input_spec=dict(gen="lattice", n=512, seed=11))
    for k, fn in jobs.items():
        if a.only and k not in a.only.split(","):
            continue
        t0 = time.time()
        r = fn()
        print(k, r, "%.1fs" % (time.time() - t0), flush=True)
        info.update(r)
    mpath = os.path.join(OUT, "MANIFEST.json")
    old = {}
    if os.path.exists(mpath):
        old = json.load(open(mpath))
    old.update(info)
    json.dump(old, open(mpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
