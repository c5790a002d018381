"""
MI355X-native SwiFTly (streaming distributed Fourier transform) -- the
facet<->subgrid hot path of ``ska_sdp_exec_swiftly`` 1.0.0 as hand-written
HIP kernels behind the reference's own API.

Public interface mirrors reference src/ska_sdp_exec_swiftly/__init__.py:4-35.
"""
from . import api  # noqa: F401  (submodule access: sw.api.preferred_wave_axis, ...)
from .api import (
    DeviceTask,
    FacetConfig,
    SubgridConfig,
    SwiftlyBackward,
    SwiftlyConfig,
    SwiftlyForward,
    make_full_facet_cover,
    make_full_subgrid_cover,
)
from .api_helper import (
    check_facet,
    check_subgrid,
    make_facet,
    make_facet_from_sources,
    make_subgrid,
    make_subgrid_from_sources,
)
from .core_hip import SwiftlyCoreHip, calculate_pswf
from .swift_configs import SWIFT_CONFIGS

__all__ = [
    "DeviceTask",
    "FacetConfig",
    "SubgridConfig",
    "SwiftlyConfig",
    "SwiftlyForward",
    "SwiftlyBackward",
    "SwiftlyCoreHip",
    "SWIFT_CONFIGS",
    "calculate_pswf",
    "check_facet",
    "check_subgrid",
    "make_subgrid",
    "make_facet",
    "make_full_facet_cover",
    "make_full_subgrid_cover",
    "make_facet_from_sources",
    "make_subgrid_from_sources",
]
