"""
MI355X-native SwiFTly (streaming distributed Fourier transform) -- the
facet<->subgrid hot path of ``ska_sdp_exec_swiftly`` 1.0.0 as hand-written
HIP kernels behind the reference's own API.

Public interface mirrors reference src/ska_sdp_exec_swiftly/__init__.py:4-35.
"""
from .core_hip import SwiftlyCoreHip, calculate_pswf

__all__ = ["SwiftlyCoreHip", "calculate_pswf"]
