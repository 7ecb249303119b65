"""Build and load the native library (sift3d_amd/libsift3d_amd.so).

The library is the product: HIP kernels + C host code behind a C ABI.  There is no
Python or CPU implementation to fall back to -- if the library cannot be built or
loaded, importing the bindings raises.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# SIFT3D_AMD_LIB: another build of the same library (the sanitizer build, `make -C sift3d_amd/csrc asan`)
LIB_PATH = os.environ.get("SIFT3D_AMD_LIB") or os.path.join(HERE, "libsift3d_amd.so")
CSRC = os.path.join(HERE, "csrc")

_lib = None


def build(force=False):
    """Compile the HIP kernels for gfx950 and the C host code (in-tree, via make)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])      # four device translation units
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce %s" % LIB_PATH)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # the .so is git-ignored; build it where a toolchain exists
            build()
        # torch ships its own libamdhip64.so.7; importing it first makes the dynamic linker
        # resolve our NEEDED libamdhip64.so.7 to that already-loaded runtime, so torch tensors
        # and our kernels share ONE HIP runtime (streams, device pointers, events).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)  # RTLD_LOCAL: the reference build in oracle/_ref exports the same 27 names
    return _lib
