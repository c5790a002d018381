"""
Configuration side of the streaming API: facet / subgrid chunk descriptions with their masks, cover helpers and
``SwiftlyConfig`` -- signatures and semantics of the reference (src/ska_sdp_exec_swiftly/api.py:39-214, 593-612;
api_helper.py:243-253).
"""
import numpy

from .core_hip import SwiftlyCoreHip

def make_mask_from_slice(slice_list, mask_size):
    """0/1 float vector that is 1 inside the given slices (reference
    api_helper.py:243-253)."""
    mask = numpy.zeros((mask_size,))
    for piece in slice_list:
        mask[piece] = 1
    return mask


class _ChunkConfig:
    """Offsets, size and masks of one facet or subgrid (reference
    api.py:39-104).  A mask may be an array, ``None`` or ``[[slices], size]``."""

    def __init__(self, off0, off1, size, mask0=None, mask1=None):
        self.off0 = off0
        self.off1 = off1
        self.size = size
        self._mask0 = mask0
        self._mask1 = mask1

    @staticmethod
    def _expand(mask):
        if isinstance(mask, list):
            return make_mask_from_slice(mask[0], mask[1])
        return mask

    @property
    def mask0(self):
        """mask along axis 0"""
        return self._expand(self._mask0)

    @property
    def mask1(self):
        """mask along axis 1"""
        return self._expand(self._mask1)


class FacetConfig(_ChunkConfig):
    """Facet configuration (reference api.py:39-70)"""


class SubgridConfig(_ChunkConfig):
    """Subgrid configuration (reference api.py:73-104)"""


def make_full_cover_config(N, chunk_size, class_name):
    """Cover the N x N plane with ``chunk_size`` pieces at multiples of
    ``chunk_size``; where neighbours overlap (also across the wrap-around) the
    masks hand each pixel to exactly one piece by cutting half way between the
    two offsets (reference api_helper.py:213-240)."""
    count = -(-N // chunk_size)
    offsets = [chunk_size * i for i in range(count)]
    cuts = [(offsets[i] + (offsets[i + 1] if i + 1 < count else N + offsets[0])) // 2 for i in range(count)]
    spans = []
    for i, off in enumerate(offsets):
        lo = (cuts[i - 1] - off + chunk_size // 2) % N
        hi = cuts[i] - off + chunk_size // 2
        spans.append((lo, hi))
    return [
        class_name(o0, o1, chunk_size, [[slice(*spans[i0])], chunk_size], [[slice(*spans[i1])], chunk_size])
        for i0, o0 in enumerate(offsets)
        for i1, o1 in enumerate(offsets)
    ]


def make_full_subgrid_cover(swiftlyconfig):
    """Subgrid configs covering the whole grid (reference api.py:593-601)"""
    return make_full_cover_config(swiftlyconfig.image_size, swiftlyconfig.max_subgrid_size, SubgridConfig)


def make_full_facet_cover(swiftlyconfig):
    """Facet configs covering the whole image (reference api.py:604-612)"""
    return make_full_cover_config(swiftlyconfig.image_size, swiftlyconfig.max_facet_size, FacetConfig)


class SwiftlyConfig:
    """SwiFTly parameters + the core that implements them (reference
    api.py:107-214)."""

    # pylint: disable=too-many-arguments,too-many-instance-attributes
    def __init__(
        self, W, fov, N, yB_size, yN_size, xA_size, xM_size, dask_client=None, backend="hip", column_precision=None,
        axis1_first=False, **_other_args
    ):
        self._W = W
        self._fov = fov
        self._N = N
        self._yB_size = yB_size
        self._yN_size = yN_size
        self._xA_size = xA_size
        self._xM_size = xM_size
        self.dask_client = dask_client  # unused: there is no Dask in this backend
        if backend == "hip":
            # column_precision=64 / axis1_first=True: the two opt-in accuracy modes of the complex64 band pipeline
            # (float64 arithmetic in the column passes; the contiguous axis finished before the strided one -- in the
            # epilogue of K1 where the configuration allows, axis1_first="rows": always by a row pass per wave)
            self._core = SwiftlyCoreHip(W, N, xM_size, yN_size, column_precision=column_precision, axis1_first=axis1_first)
        elif backend in ("numpy", "ska_sdp_func"):
            # reference api.py:137-141 -- those cores live in the reference package; this one is GPU only
            raise ValueError(
                f"SwiFTly backend {backend!r} is provided by ska_sdp_exec_swiftly itself; "
                "ska_sdp_exec_swiftly_amd only implements backend='hip' (no CPU fallback)"
            )
        else:
            raise ValueError(f"Unknown SwiFTly backend: {backend}")
        # the reference wraps a scattered core in dask.delayed (api.py:145-147)
        self.core_task = self._core

    @property
    def core(self):
        """the SwiftlyCoreHip instance"""
        return self._core

    @property
    def image_size(self):
        """Size of the entire (virtual) image in pixels"""
        return self._N

    @property
    def max_facet_size(self):
        """Maximum size of a facet in pixels"""
        return self._yB_size

    @property
    def max_subgrid_size(self):
        """Maximum size of a subgrid in pixels"""
        return self._xA_size

    @property
    def pswf_parameter(self):
        """PSWF window parameter W"""
        return self._W

    @property
    def internal_facet_size(self):
        """Padded facet size used internally"""
        return self._yN_size

    @property
    def internal_subgrid_size(self):
        """Padded subgrid size used internally"""
        return self._xM_size

    @property
    def facet_off_step(self):
        """All facet offsets must be divisible by this"""
        return self._core.facet_off_step

    @property
    def subgrid_off_step(self):
        """All subgrid offsets must be divisible by this"""
        return self._core.subgrid_off_step
