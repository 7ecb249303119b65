"""The drop-in boundary as a C consumer sees it: a program written against the reference's
public headers (tests/c_program.c) is compiled with gcc against include/sift3d and linked with
libsift3d_amd.so -- no Python in the loop.  Round 5: the same under the reference library's own
name -- libsift3D.so.2, the soname the reference installs (/root/reference/sift3d/CMakeLists.txt:11-14)
-- and the reference's own consumer, cli/kpSift3D.c, compiled UNCHANGED against include/sift3d and
linked with -lsift3D (/root/reference/cli/CMakeLists.txt:3-4, cli/kpSift3D.c:96-146)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_program.c")


REF_CLI = "/root/reference/cli/kpSift3D.c"
CLI_BIN = os.path.join(ROOT, "oracle", "_ref", "kpSift3D")   # built by `make -C oracle ref` (build container)


def _needed(path):
    out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
    return [ln.split("[")[1].split("]")[0] for ln in out.splitlines() if "(NEEDED)" in ln]


def _build(tmp_path, lib="-lsift3d_amd"):
    from sift3d_amd import _native
    _native.load()                       # make sure the library exists
    exe = str(tmp_path / "c_program")
    libdir = os.path.join(ROOT, "sift3d_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", SRC, "-I" + os.path.join(ROOT, "include"),
                    "-L" + libdir, lib, "-Wl,-rpath," + libdir, "-o", exe], check=True)
    return exe


def test_reference_soname_artifact(tmp_path):
    """sift3d_amd/libsift3D.so.2 carries the reference's soname and exports its 27 names; a program linked
    against it records NEEDED libsift3D.so.2, exactly as one linked against the reference does."""
    from sift3d_amd import _native
    _native.load()
    lib = os.path.join(ROOT, "sift3d_amd", "libsift3D.so.2")
    assert os.path.exists(lib), "make -C sift3d_amd/csrc builds it beside libsift3d_amd.so"
    out = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True, check=True).stdout
    assert "Library soname: [libsift3D.so.2]" in out
    sym = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1] for ln in sym.splitlines() if ln.strip()}
    import re
    decl = set()
    for h in ("sift.h", "imutil.h"):
        decl |= set(re.findall(r"\b(sift3d_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "sift3d", h)).read()))
    assert len(decl) == 27 and decl <= names, sorted(decl - names)
    exe = _build(tmp_path, "-l:libsift3D.so.2")
    assert "libsift3D.so.2" in _needed(exe) and "libsift3d_amd.so" not in _needed(exe)


@pytest.mark.refprobe
def test_reference_cli_compiles_unchanged_and_links(tmp_path):
    """cli/kpSift3D.c -- the reference's own consumer of the API -- compiled as it is against include/sift3d
    and linked with -lsift3D; without a device it reports the failure the way it would with the reference."""
    import torch
    if not os.path.exists(REF_CLI):
        pytest.skip("needs /root/reference (build container only)")
    from sift3d_amd import _native
    _native.load()
    libdir = os.path.join(ROOT, "sift3d_amd")
    exe = str(tmp_path / "kpSift3D")
    subprocess.run(["gcc", "-Wall", "-O2", "-I" + os.path.join(ROOT, "include", "sift3d"), REF_CLI, "-o", exe,
                    "-L" + libdir, "-lsift3D", "-Wl,-rpath," + libdir], check=True)
    assert "libsift3D.so.2" in _needed(exe)
    r = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "kpSift3D" in r.stdout
    if torch.cuda.is_available():
        return
    from tests.test_host_api import _write_nii
    vol = np.random.default_rng(3).random((24, 24, 24)).astype(np.float32)
    _write_nii(tmp_path / "v.nii", vol)
    r = subprocess.run([exe, "--keys", str(tmp_path / "k.csv"), str(tmp_path / "v.nii")], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 1 and "Failed to detect keypoints." in r.stderr      # cli/kpSift3D.c:116-120


def test_c_program_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe, "24", "24", "24", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rc1, rc2, rows, cols, _ = r.stdout.split()
    assert int(rc1) == -1 and int(rc2) == -2 and int(rows) == 0   # no device: no CPU fallback
    assert r.stderr.strip() != ""                                  # ... and it says so


@pytest.mark.gpu
def test_c_program_equals_python_api(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from sift3d_amd import api
    exe = _build(tmp_path)
    r = subprocess.run([exe, "48", "40", "36", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rc1, rc2, rows, cols, chk = r.stdout.split()
    assert int(rc1) == 0 and int(rc2) == 0 and int(cols) == 771 and int(rows) > 0
    # the same volume through the ctypes mirror
    s = 7
    n = 48 * 40 * 36
    vol = np.zeros(n, np.float32)
    M = (1 << 64) - 1
    for k in range(n):
        s ^= (s << 13) & M; s ^= s >> 7; s ^= (s << 17) & M
        vol[k] = np.float32((s >> 11) * (1.0 / 9007199254740992.0))
    for _ in range(40):
        s ^= (s << 13) & M; s ^= s >> 7; s ^= (s << 17) & M
        vol[s % n] += np.float32(25.0)
    det, kp, desc = api.Detector(peak_thresh=0.05), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.from_array(vol.reshape(36, 40, 48)), kp) == 0
    kp.sort_by_strength(100)
    assert det.extract_descriptors(kp, desc) == 0
    m = desc.to_mat_rm()
    assert m.shape == (int(rows), 771)
    want = float((m.reshape(-1).astype(np.float64) * (1 + np.arange(m.size) % 7)).sum())
    assert abs(want - float(chk)) <= 1e-9 * abs(want)


@pytest.mark.gpu
def test_c_program_through_the_reference_soname(tmp_path):
    """tests/c_program.c linked against libsift3D.so.2 (not libsift3d_amd.so): same output as through the
    library's own name."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    a = _build(tmp_path, "-l:libsift3D.so.2")
    assert "libsift3D.so.2" in _needed(a)
    ra = subprocess.run([a, "48", "40", "36", "7"], capture_output=True, text=True, timeout=300)
    (tmp_path / "b").mkdir()
    b = _build(tmp_path / "b")
    rb = subprocess.run([b, "48", "40", "36", "7"], capture_output=True, text=True, timeout=300)
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr, rb.stderr)
    assert ra.stdout == rb.stdout and int(ra.stdout.split()[2]) > 0


@pytest.mark.gpu
def test_reference_cli_binary_runs_on_this_library(tmp_path):
    """oracle/_ref/kpSift3D = the reference's cli/kpSift3D.c, compiled unchanged in the build container against
    include/sift3d and linked with -lsift3D (oracle/Makefile): read a NIfTI volume, detect, sort_by_strength(100),
    describe, write both CSV files (cli/kpSift3D.c:96-146) -- byte for byte the files the ctypes mirror of the
    same calls writes."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    if not os.path.exists(CLI_BIN):
        pytest.skip("oracle/_ref/kpSift3D was not built (needs /root/reference at build time)")
    assert "libsift3D.so.2" in _needed(CLI_BIN)
    from sift3d_amd import api
    from oracle import sift3d_oracle as so
    from tests.test_host_api import _write_nii
    vol = so.synth_lattice((72, 64, 80), seed=23)
    _write_nii(tmp_path / "v.nii.gz", vol, pixdim=(1.0, 1.0, 1.0))
    kf, df = tmp_path / "keys.csv", tmp_path / "desc.csv.gz"
    r = subprocess.run([CLI_BIN, "--keys", str(kf), "--desc", str(df), str(tmp_path / "v.nii.gz")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
    assert det.detect_keypoints(api.Image.read(str(tmp_path / "v.nii.gz")), kp) == 0
    kp.sort_by_strength(100)
    assert det.extract_descriptors(kp, desc) == 0
    assert len(kp) == 100
    assert kp.save(str(tmp_path / "k2.csv")) == 0 and desc.save(str(tmp_path / "d2.csv.gz")) == 0
    import gzip
    assert open(kf, "rb").read() == open(tmp_path / "k2.csv", "rb").read()
    assert gzip.open(df, "rb").read() == gzip.open(tmp_path / "d2.csv.gz", "rb").read()


@pytest.mark.gpu
def test_c_slab_driver_program(tmp_path):
    """The multi-GPU driver's C ABI from a C host program (tests/c_sharded_program.c): three ranks as threads
    over the library's stream-ordered thread transport -- create, upload, the collective detect / describe /
    descriptor gather -- and every rank compares its global keypoint matrix and the gathered N x 771
    descriptors with the drop-in single-GPU API, in C, bit for bit."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from sift3d_amd import _native
    _native.load()
    exe = str(tmp_path / "c_sharded_program")
    libdir = os.path.join(ROOT, "sift3d_amd")
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror", "-O1", "-pthread",
                    os.path.join(ROOT, "tests", "c_sharded_program.c"), "-I" + os.path.join(ROOT, "include"),
                    "-L" + libdir, "-lsift3d_amd", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe, "3", "64", "72", "288", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    tag, nkp, ncand = r.stdout.split()
    assert tag == "ok" and int(nkp) > 50 and int(ncand) >= int(nkp)
