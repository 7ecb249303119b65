"""sift3d_amd -- MI355X-native SIFT3D detect+describe (drop-in for fatimp/SIFT3D's C API).

  sift3d_amd.api      Python mirror of the reference C API (ctypes over libsift3d_amd.so)
  sift3d_amd.hip      device-level stage ABI on torch tensors
  sift3d_amd.sharded_c  bindings of the C Z-slab multi-GPU driver (sift3d_amd_sharded_*, RCCL)

The compute lives in sift3d_amd/csrc (HIP kernels + C host code); nothing here computes.
"""
from . import _native  # noqa: F401

__all__ = ["api", "hip", "_native"]
