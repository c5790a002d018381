"""
ctypes binding of libswiftly_hip.so (C ABI: include/swiftly_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``csrc/Makefile``.  There is deliberately NO fallback: if the shared object is
missing, or no HIP device is visible when a core is constructed, the product
path raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("SWIFTLY_HIP_LIB", "libswiftly_hip.so"))  # env: A/B builds

C64, C128 = 0, 1
ERR_PARAM, ERR_UNSUPPORTED, ERR_HIP = 1, 2, 3

_lib = None


class SwiftlyHipError(RuntimeError):
    """HIP runtime / launch failure reported by libswiftly_hip.so"""


def _declare(lib):
    i64, vp = c_int64, c_void_p
    lib.swiftly_hip_last_error.restype = c_char_p
    lib.swiftly_hip_last_error.argtypes = []
    lib.swiftly_hip_version.restype = c_int
    lib.swiftly_hip_device_count.restype = c_int
    lib.swiftly_hip_create.restype = c_int
    lib.swiftly_hip_create.argtypes = [POINTER(vp), i64, i64, i64, c_double, POINTER(c_double), c_int]
    lib.swiftly_hip_destroy.restype = None
    lib.swiftly_hip_destroy.argtypes = [vp]
    lib.swiftly_hip_contribution_size.restype = i64
    lib.swiftly_hip_contribution_size.argtypes = [vp]
    lib.swiftly_hip_build_id.restype = ctypes.c_char_p
    lib.swiftly_hip_build_id.argtypes = []
    lib.swiftly_hip_chain_chunk_streams.restype = None
    lib.swiftly_hip_chain_chunk_streams.argtypes = [ctypes.c_int]
    lib.swiftly_hip_set_column_precision.restype = ctypes.c_int
    lib.swiftly_hip_set_column_precision.argtypes = [vp, ctypes.c_int]
    lib.swiftly_hip_get_column_precision.restype = ctypes.c_int
    lib.swiftly_hip_get_column_precision.argtypes = [vp]
    # (h, dtype, in, rows, [size,] in_rs, in_cs, out, out_rs, out_cs, off, [size, mask,] stream)
    sized_in = [vp, c_int, vp, i64, i64, i64, i64, vp, i64, i64, i64, vp]
    plain = [vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, vp]
    finish = [vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, i64, vp, vp]
    for name, args in [
        ("prepare_facet", sized_in),
        ("extract_from_facet", plain),
        ("add_to_subgrid", plain),
        ("finish_subgrid", finish),
        ("prepare_subgrid", sized_in),
        ("extract_from_subgrid", plain),
        ("add_to_facet", plain),
        ("finish_facet", finish),
    ]:
        fn = getattr(lib, "swiftly_hip_" + name)
        fn.restype = c_int
        fn.argtypes = args
    # the 2-D single-call forms of the native shim (core.py:752-778, 837-855)
    lib.swiftly_hip_add_to_subgrid_2d.restype = c_int
    lib.swiftly_hip_add_to_subgrid_2d.argtypes = [vp, c_int, vp, i64, i64, vp, i64, i64, i64, i64, vp]
    lib.swiftly_hip_prepare_subgrid_inplace.restype = c_int
    lib.swiftly_hip_prepare_subgrid_inplace.argtypes = [vp, c_int, vp, i64, i64, i64, i64, vp]
    lib.swiftly_hip_prepare_subgrid_inplace_2d.restype = c_int
    lib.swiftly_hip_prepare_subgrid_inplace_2d.argtypes = [vp, c_int, vp, i64, i64, i64, i64, vp]
    pi64 = POINTER(i64)
    batch = [i64, i64, i64, pi64]  # nbatch, in_bs, out_bs, offs
    for name, args in [
        ("extract_column", [vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, i64, vp]),
        ("extract_column_rows", [vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, i64, vp, c_int, vp]),
        ("prepare_facet_rows", sized_in[:-1] + [vp, c_int, vp]),
        ("extract_from_facet_batch", plain[:-1] + batch + [vp]),
        ("add_to_subgrid_batch", plain[:-1] + batch + [vp]),
        ("finish_subgrid_batch", finish[:-1] + batch + [i64, vp]),
        ("prepare_subgrid_batch", sized_in[:-1] + batch + [vp]),
        ("extract_from_subgrid_batch", plain[:-1] + batch + [vp]),
        ("add_to_facet_batch", plain[:-1] + batch + [vp]),
        ("finish_facet_batch", finish[:-1] + batch + [i64, vp]),
    ]:
        fn = getattr(lib, "swiftly_hip_" + name)
        fn.restype = c_int
        fn.argtypes = args
    lib.swiftly_hip_sum_finish_rows.restype = c_int
    lib.swiftly_hip_sum_finish_rows.argtypes = [vp, c_int, vp, i64, i64, i64, i64, pi64, vp, i64, i64, pi64, i64, vp, i64, i64, vp]
    lib.swiftly_hip_add_to_subgrid_from_columns.restype = c_int
    lib.swiftly_hip_add_to_subgrid_from_columns.argtypes = [vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, i64, pi64, vp]
    lib.swiftly_hip_band_columns.restype = i64
    lib.swiftly_hip_band_columns.argtypes = [i64]
    lib.swiftly_hip_band_columns_for.restype = i64
    lib.swiftly_hip_band_columns_for.argtypes = [vp, i64]
    lib.swiftly_hip_prepare_facet_band.restype = c_int
    lib.swiftly_hip_prepare_facet_band.argtypes = [vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, i64, c_int, vp]
    lib.swiftly_hip_prepare_facet_band_rows.restype = c_int
    lib.swiftly_hip_prepare_facet_band_rows.argtypes = [vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, i64, i64, i64, vp]
    lib.swiftly_hip_prepare_facet_columns.restype = c_int
    lib.swiftly_hip_prepare_facet_columns.argtypes = [vp, c_int, vp, i64, i64, i64, i64, pi64, i64, i64, i64, vp, i64, i64, vp, vp]
    lib.swiftly_hip_transform_contributions.restype = c_int
    lib.swiftly_hip_transform_contributions.argtypes = [
        vp, c_int, vp, c_int, i64, i64, i64, vp, i64, i64, i64, pi64, i64, pi64, vp, i64, i64, vp,
    ]
    lib.swiftly_hip_sum_finish_facets.restype = c_int
    lib.swiftly_hip_sum_finish_facets.argtypes = [
        vp, c_int, vp, i64, i64, i64, i64, pi64, pi64, vp, i64, i64, pi64, i64, vp, i64, i64, vp,
    ]
    lib.swiftly_hip_wave_facet_side.restype = c_int
    lib.swiftly_hip_wave_facet_side.argtypes = [
        vp, c_int, vp, i64, i64, i64, i64, pi64, i64, i64, i64, vp, i64, vp, i64, c_int, i64, pi64, vp, i64, i64, pi64, pi64,
        vp, i64, vp,
    ]
    lib.swiftly_hip_prepare_facet_columns_waves.restype = c_int
    lib.swiftly_hip_prepare_facet_columns_waves.argtypes = [
        vp, c_int, vp, i64, i64, i64, i64, pi64, i64, i64, i64, pi64, vp, i64, i64, i64, vp, i64, vp, i64, vp,
    ]
    lib.swiftly_hip_wave_subgrid_side.restype = c_int
    lib.swiftly_hip_wave_subgrid_side.argtypes = [
        vp, c_int, vp, i64, i64, i64, pi64, pi64, i64, pi64, pi64, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp,
    ]
    lib.swiftly_hip_wave_subgrid_side_placed.restype = c_int
    lib.swiftly_hip_wave_subgrid_side_placed.argtypes = list(lib.swiftly_hip_wave_subgrid_side.argtypes)
    lib.swiftly_hip_prepare_facet_window_rows.restype = c_int
    lib.swiftly_hip_prepare_facet_window_rows.argtypes = [
        vp, c_int, vp, i64, i64, i64, vp, i64, i64, i64, i64, i64, i64, vp, i64, i64, vp,
    ]
    lib.swiftly_hip_finish_axis1_rows.restype = c_int
    lib.swiftly_hip_finish_axis1_rows.argtypes = [vp, c_int, vp, i64, i64, i64, i64, pi64, i64, i64, i64, vp, i64, i64, vp]
    lib.swiftly_hip_accumulate_facet_columns.restype = c_int
    lib.swiftly_hip_accumulate_facet_columns.argtypes = [
        vp, c_int, vp, i64, i64, pi64, pi64, vp, i64, pi64, i64, vp, i64, vp, i64, i64, i64, i64, vp, vp, i64, vp,
    ]
    lib.swiftly_hip_split_prepare_facets.restype = c_int
    lib.swiftly_hip_split_prepare_facets.argtypes = [vp, c_int, vp, i64, i64, i64, i64, pi64, i64, pi64, pi64, vp, i64, i64, vp]
    lib.swiftly_hip_wave_split_subgrids.restype = c_int
    lib.swiftly_hip_wave_split_subgrids.argtypes = [vp, c_int, vp, i64, i64, pi64, pi64, i64, pi64, pi64, vp, i64, vp, i64, i64, vp]
    lib.swiftly_hip_band_zero_untouched.restype = c_int
    lib.swiftly_hip_band_zero_untouched.argtypes = [vp, c_int, vp, i64, i64, i64, vp, vp]
    lib.swiftly_hip_finish_facet_band.restype = c_int
    lib.swiftly_hip_finish_facet_band.argtypes = [vp, c_int, vp, i64, i64, i64, i64, vp, i64, i64, i64, vp, vp]
    lib.swiftly_hip_malloc.restype = c_int
    lib.swiftly_hip_malloc.argtypes = [POINTER(vp), c_size_t]
    lib.swiftly_hip_free.restype = c_int
    lib.swiftly_hip_free.argtypes = [vp]
    lib.swiftly_hip_memset_async.restype = c_int
    lib.swiftly_hip_memset_async.argtypes = [vp, c_int, c_size_t, vp]
    for name in ("swiftly_hip_memcpy_h2d", "swiftly_hip_memcpy_d2h"):
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = [vp, vp, c_size_t, vp]
    lib.swiftly_hip_stream_synchronize.restype = c_int
    lib.swiftly_hip_stream_synchronize.argtypes = [vp]
    lib.swiftly_hip_stream_create_cu_mask.restype = c_int
    lib.swiftly_hip_stream_create_cu_mask.argtypes = [POINTER(vp), POINTER(ctypes.c_uint32), c_int]
    lib.swiftly_hip_stream_destroy.restype = c_int
    lib.swiftly_hip_stream_destroy.argtypes = [vp]
    lib.swiftly_hip_cu_census.restype = c_int
    lib.swiftly_hip_cu_census.argtypes = [vp, c_int, vp]


def load():
    """Load (once) and return the ctypes library object."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ska-sdp-distributed-fourier-transform_amd/csrc`. "
            "The HIP backend has no CPU fallback."
        )
    # torch first: its bundled libamdhip64.so.7 then serves this library too, so
    # device pointers and streams are shared with torch tensors.
    import torch  # noqa: F401  pylint: disable=import-outside-toplevel,unused-import

    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    _declare(lib)
    _lib = lib
    return lib


def check(rc):
    """Translate a non-zero ABI status into the exception the reference's
    backends raise for the same condition."""
    if rc == 0:
        return
    msg = load().swiftly_hip_last_error().decode("utf-8", "replace")
    if rc == ERR_PARAM:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise SwiftlyHipError(msg)


def build_info():
    """Identity of the running native build: ``src_hash`` compiled into the library (``swiftly_hip_build_id``: hash of
    the kernel sources), ``so_sha256`` of the loaded file, and the git commit the objects were built at (stamped by
    the Makefile beside them; ``None`` when absent)."""
    import hashlib  # pylint: disable=import-outside-toplevel

    lib = load()
    with open(LIB_PATH, "rb") as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()
    head = None
    try:
        with open(os.path.join(os.path.dirname(_HERE), "csrc", "build", "GIT_HEAD"), encoding="utf-8") as fh:
            head = "".join(fh.read().split())
    except OSError:
        pass
    return dict(src_hash=lib.swiftly_hip_build_id().decode(), so_sha256=sha[:16], git_head=head, library=os.path.basename(LIB_PATH))
