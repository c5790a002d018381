"""
Host-side helpers with the names and semantics of the reference's
``api_helper`` / ``fourier_algorithm`` modules that users of the streaming API
rely on: synthetic facet / subgrid generation from point sources, and RMSE
checks (reference api_helper.py:15-70, fourier_algorithm.py:218-315).
These run on the CPU with numpy -- they produce test *inputs* and *truths*,
they are not part of the accelerated path.
"""
import numpy

__all__ = [
    "make_facet_from_sources",
    "make_subgrid_from_sources",
    "make_facet",
    "make_subgrid",
    "check_facet",
    "check_subgrid",
    "check_residual",
]


def _along(vec, ndim, axis):
    idx = [numpy.newaxis] * ndim
    idx[axis] = slice(None)
    return numpy.asarray(vec)[tuple(idx)]


def _to_numpy(arr):
    if hasattr(arr, "detach"):
        arr = arr.detach().cpu().numpy()
    return numpy.asarray(arr)


def make_facet_from_sources(sources, image_size, facet_size, facet_offsets, facet_masks=None):
    """Image-space facet holding the given point sources ``(intensity,
    *coords)``; coordinates are relative to the image centre and wrap modulo
    ``image_size`` (reference fourier_algorithm.py:218-264)."""
    ndim = len(facet_offsets)
    facet = numpy.zeros(ndim * [facet_size], dtype=complex)
    corner = numpy.array(facet_offsets, dtype=int) - facet_size // 2
    for intensity, *coord in sources:
        pixel = (numpy.array(coord, dtype=int) - corner) % image_size
        if (pixel < facet_size).all():
            facet[tuple(pixel)] += intensity
    for axis, mask in enumerate(facet_masks or []):
        if mask is not None:
            facet *= _along(mask, ndim, axis)
    return facet


def make_subgrid_from_sources(sources, image_size, subgrid_size, subgrid_offsets, subgrid_masks=None):
    """Grid-space subgrid of the given point sources by direct Fourier sum,
    normalised by ``image_size**ndim`` (reference
    fourier_algorithm.py:267-315)."""
    ndim = len(subgrid_offsets)
    subgrid = numpy.zeros(ndim * [subgrid_size], dtype=complex)
    lo, hi = subgrid_size // 2, (subgrid_size + 1) // 2
    for intensity, *coord in sources:
        wave = numpy.ones(ndim * [1], dtype=complex) * (intensity / image_size**ndim)
        for axis in range(ndim):
            uv = numpy.arange(subgrid_offsets[axis] - lo, subgrid_offsets[axis] + hi)
            wave = wave * _along(numpy.exp(2j * numpy.pi * coord[axis] * uv / image_size), ndim, axis)
        subgrid += wave
    for axis, mask in enumerate(subgrid_masks or []):
        if mask is not None:
            subgrid *= _along(mask, ndim, axis)
    return subgrid


def make_subgrid(image_size, sg_config, sources):
    """reference api_helper.py:15-24"""
    return make_subgrid_from_sources(
        sources, image_size, sg_config.size, [sg_config.off0, sg_config.off1], [sg_config.mask0, sg_config.mask1]
    )


def make_facet(image_size, facet_config, sources):
    """reference api_helper.py:27-36"""
    return make_facet_from_sources(
        sources,
        image_size,
        facet_config.size,
        [facet_config.off0, facet_config.off1],
        [facet_config.mask0, facet_config.mask1],
    )


def _rms(arr):
    return numpy.sqrt(numpy.mean(numpy.abs(arr) ** 2))


def check_facet(image_size, facet_config, approx_facet, sources):
    """RMSE between a computed facet and the one generated from the sources
    (reference api_helper.py:39-48)"""
    return _rms(make_facet(image_size, facet_config, sources) - _to_numpy(approx_facet))


def check_residual(residual_facet):
    """RMS of a residual image (reference api_helper.py:51-55)"""
    return _rms(_to_numpy(residual_facet))


def check_subgrid(image_size, sg_config, approx_subgrid, sources):
    """RMSE between a computed subgrid and the direct Fourier sum (reference
    api_helper.py:58-70)"""
    approx = _to_numpy(approx_subgrid)
    truth = make_subgrid_from_sources(
        sources, image_size, approx.shape[0], [sg_config.off0, sg_config.off1], [sg_config.mask0, sg_config.mask1]
    )
    return _rms(truth - approx)
