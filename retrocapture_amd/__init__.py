"""retrocapture_amd - MI355X-native shader chain for RetroCapture presets.

The product is the C-ABI library ``librcshaderchain.so`` (HIP kernels + C++ host,
``retrocapture_amd/csrc``, header ``include/rc_shaderchain.h``).  This package is a thin
ctypes mirror of the reference's ``ShaderEngine`` class for tests, benchmarks and Python
callers; it contains no compute and has no CPU fallback.
"""
from .engine import ShaderEngine, ShaderParameter, load_library, library_path, RcError  # noqa: F401

__all__ = ["ShaderEngine", "ShaderParameter", "load_library", "library_path", "RcError"]
