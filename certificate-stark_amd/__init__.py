"""MI355X-native prover backend for the Topos state-transition AIR (hot path of toposware/certificate-stark).

Layout: csrc/ (HIP kernels + C ABI), _lib.py (ctypes binding of include/cstark.h), backend.py (device
plumbing on torch tensors/streams), prover.py (host-side mirror of the reference's prover interface).
"""
from . import _lib  # noqa: F401
from ._lib import CstarkError  # noqa: F401

__all__ = ["_lib", "CstarkError"]
