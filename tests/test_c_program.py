"""The drop-in boundary as a C consumer sees it: a program written against the reference's
public headers (tests/c_program.c) is compiled with gcc against include/sift3d and linked with
libsift3d_amd.so -- no Python in the loop."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_program.c")


def _build(tmp_path):
    from sift3d_amd import _native
    _native.load()                       # make sure the library exists
    exe = str(tmp_path / "c_program")
    libdir = os.path.join(ROOT, "sift3d_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O1", SRC, "-I" + os.path.join(ROOT, "include"),
                    "-L" + libdir, "-lsift3d_amd", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    return exe


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
