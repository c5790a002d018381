"""
``SwiftlyCoreHip`` -- the MI355X sibling of the reference's ``SwiftlyCore`` /
``SwiftlyCoreFunc`` (reference src/ska_sdp_exec_swiftly/fourier_transform/
core.py:20-484 and 487-929).

It exposes the same duck-typed surface (constructor ``(W, N, xM_size,
yN_size)``, attributes ``W N xM_size yN_size xM_yN_size``, properties
``subgrid_off_step facet_off_step``, the eight primitives plus
``add_to_subgrid_2d``, the ``out=`` convention of core.py:152-186 and the
``ValueError`` behaviour), so it drops into ``api_helper``-style task bodies
and into the reference's tests unchanged.  All arithmetic runs in hand-written
HIP kernels behind the C ABI of ``libswiftly_hip.so`` (include/swiftly_hip.h);
this file only marshals arrays.  There is no CPU path: constructing a core
without a visible GPU raises.

Array types.  numpy in -> numpy out (host<->device copies through torch,
convenient and exactly what the reference's callers expect); torch CUDA tensor
in -> torch CUDA tensor out, zero copy, asynchronous on the current torch
stream (what the streaming classes in ``api.py`` use).  complex64 stays
complex64, complex128 stays complex128, real input is promoted to the
matching complex type (core.py:581-585).
"""
import ctypes
import functools

import numpy
import scipy.special

from . import _lib

__all__ = ["SwiftlyCoreHip", "calculate_pswf"]


def calculate_pswf(W, yN_size):
    """Host-side PSWF samples, as reference core.py:119-150: the zeroth-order
    prolate spheroidal angular function of parameter ``pi*W/2`` at
    ``2*(k - yN//2)/yN`` with element 0 zeroed.  scipy's ``pro_ang1`` is fed in
    slices of 500 like the reference does (scipy crashes on long inputs), so
    the constants are bit-identical to the reference's."""
    half = yN_size // 2
    hi = half if yN_size % 2 == 0 else half + 1
    xs = 2.0 * numpy.arange(-half, hi, dtype=float) / yN_size
    vals = numpy.empty(yN_size, dtype=float)
    for start in range(1, yN_size, 500):
        stop = start + 500
        vals[start:stop] = scipy.special.pro_ang1(0, 0, numpy.pi * W / 2, xs[start:stop])[0]
    vals[0] = 0.0
    return vals


def band_range(N, yN, m, subgrid_offs, align=32):
    """Smallest cyclic range ``(start, length)`` of padded-facet columns that contains the ``m`` window
    (core.py:243-253) of every subgrid offset; ``(0, yN)`` when that is everything (pure numpy, unit-tested)."""
    return _band_range_cached(int(N), int(yN), int(m), tuple(sorted(set(int(o) for o in subgrid_offs))), int(align))


@functools.lru_cache(maxsize=256)
def _band_range_cached(N, yN, m, subgrid_offs, align):
    # (memoised, r6: the streaming classes are rebuilt every pass and ask for the same few bands -- 30 us of numpy each,
    # on the enqueue path of a multi-GPU rank)
    keep = numpy.zeros(yN, dtype=bool)
    for off in subgrid_offs:
        s = off * yN // N
        keep[(yN // 2 - m // 2 + numpy.arange(m) + s) % yN] = True
    if keep.all() or not keep.any():
        return 0, yN
    # largest run of unused indices on the ring; the band is its complement
    idx = numpy.flatnonzero(keep)
    gaps = numpy.diff(numpy.concatenate([idx, [idx[0] + yN]]))
    g = int(numpy.argmax(gaps))
    start = int(idx[(g + 1) % idx.size])
    length = int(yN - (gaps[g] - 1))
    # Start the band a few columns early so that (window origin - start) is a multiple of 32: a 64-column tile of
    # the column pass then reads two 256-byte runs of the parity-split layout that both begin on a 128-byte line for
    # EVERY wave (the window rotation and the window offset cancel modulo 32 when m is a multiple of 32).  Measured
    # (r3, rocprofv3 FETCH_SIZE of K2 pass A per wave): 0.83 GB algorithmic, 1.04 GB with one run off a line, 1.26 GB
    # with both.
    if align > 1:
        shift = (start - (yN // 2 - m // 2)) % align
        if length + shift <= yN:
            start = (start - shift) % yN
            length += shift
    return (0, yN) if length >= yN else (start, length)


def mixed_factor(n):
    """``(Q, k)`` when ``n = Q * 2^k`` with Q in {3, 5, 7, 9} and ``2^k >= 8`` -- the lengths that run through one
    radix-Q pass in front of the power-of-two kernels (csrc/swiftly_mixed.h; the same rule as ``mixed_factor`` in
    csrc/swiftly_abi.hip) -- else None (powers of two included: they need no pass)."""
    n = int(n)
    for q in (3, 5, 7, 9):
        if n > 0 and n % q == 0:
            r = n // q
            if r >= 8 and r & (r - 1) == 0:
                return q, r.bit_length() - 1
    return None


def build_row_sources(N, yN, m, sub_off0s, locations, max_chunks=16):
    """Host tables of the gather-sum load (see :py:meth:`SwiftlyCoreHip.column_row_sources`; pure numpy, unit-tested
    on CPU against the oracle's ``add_to_facet``): list of ``(subgrid indices, int32 table [2, yN])``.  Row ``big``
    of the padded axis is the sum of the contribution rows named by ``table[0, big]`` and ``table[1, big]``
    (negative = none), each ``chunk << 20 | (block * m + k)``: the transpose of ``extract_from_facet``
    (core.py:243-253), i.e. ``add_to_facet`` (core.py:441-478) as a gather."""
    k = numpy.arange(m)
    groups = []
    tab = cnt = members = None
    for b, off in enumerate(sub_off0s):
        s = int(off) * yN // N
        big = (yN // 2 - m // 2 + s + ((k - s) % m)) % yN
        chunk, blk = locations[b]
        if blk * m + m > (1 << 20) or chunk >= max_chunks:
            raise ValueError("too many contribution rows / chunks for one accumulate_facet_columns call")
        if tab is None or (cnt[big] >= 2).any():
            tab = numpy.full((2, yN), -1, dtype=numpy.int32)
            cnt = numpy.zeros(yN, dtype=numpy.int64)
            members = []
            groups.append((members, tab))
        tab[cnt[big], big] = (chunk << 20) | (blk * m + k)
        cnt[big] += 1
        members.append(b)
    return groups


def _torch():
    import torch  # pylint: disable=import-outside-toplevel

    return torch


class SwiftlyCoreHip:
    """SwiFTly primitives on one MI355X.

    :param W: PSWF parameter (grid-space support)
    :param N: total image size
    :param xM_size: padded subgrid size
    :param yN_size: padded facet size
    :param device: HIP device index (default: torch's current device)
    """

    # pylint: disable=too-many-public-methods,too-many-arguments

    def __init__(self, W, N, xM_size, yN_size, device=None, column_precision=None, axis1_first=False):
        # (r6) axis-1-first forward band pipeline (finish_axis1_rows): the streaming classes read this switch
        # False / True / "rows" (SwiftlyConfig): the axis-1-first order of the band pipeline -- True: with the contiguous-axis
        # finish fused into K1 where the configuration allows (else as "rows"); "rows": always a row pass per wave
        self.axis1_first = "rows" if axis1_first == "rows" else bool(axis1_first)
        self.W = W
        self.N = N
        self.xM_size = xM_size
        self.yN_size = yN_size
        self.check_params()
        self.xM_yN_size = self.xM_size * self.yN_size // self.N
        self._handle = None
        self._lib = _lib.load()
        torch = _torch()
        if self._lib.swiftly_hip_device_count() <= 0 or not torch.cuda.is_available():
            raise RuntimeError(
                "SwiftlyCoreHip needs a HIP device (MI355X); none is visible and there is no CPU fallback"
            )
        self._device_index = torch.cuda.current_device() if device is None else int(device)
        self._device = torch.device("cuda", self._device_index)
        pswf = numpy.ascontiguousarray(calculate_pswf(W, yN_size))
        handle = ctypes.c_void_p()
        _lib.check(
            self._lib.swiftly_hip_create(
                ctypes.byref(handle),
                N,
                yN_size,
                xM_size,
                float(W),
                pswf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                self._device_index,
            )
        )
        self._handle = handle
        if column_precision is not None:
            self.column_precision = column_precision

    @property
    def column_precision(self):
        """Arithmetic of the column passes of the band pipelines on complex64 data (include/swiftly_hip.h,
        swiftly_hip_set_column_precision): 32 (default, or 64 when SWIFTLY_COL_F64=1) | 64 = float64 butterflies
        between complex64 loads and stores: 3.7x smaller end-to-end error, 1.5x the pass time."""
        return int(self._lib.swiftly_hip_get_column_precision(self._handle))

    @column_precision.setter
    def column_precision(self, bits):
        _lib.check(self._lib.swiftly_hip_set_column_precision(self._handle, int(bits)))

    def __del__(self):
        handle = getattr(self, "_handle", None)
        if handle:
            try:
                self._lib.swiftly_hip_destroy(handle)
            except Exception:  # pylint: disable=broad-except
                pass
            self._handle = None

    # Pickle support like SwiftlyCoreFunc (core.py:512-525): only the four
    # parameters travel, the native handle is rebuilt on the receiving side.
    def __getstate__(self):
        # no device pointers, no caches
        return {"W": self.W, "N": self.N, "xM_size": self.xM_size, "yN_size": self.yN_size,
                "column_precision": self.column_precision, "axis1_first": self.axis1_first}

    def __setstate__(self, state):
        self.__init__(**state)

    def check_params(self):
        """Validate sizes (core.py:55-74)."""
        if self.N % self.yN_size != 0:
            raise ValueError(f"Image size {self.N} not divisible by facet size {self.yN_size}!")
        if self.N % self.xM_size != 0:
            raise ValueError(f"Image size {self.N} not divisible by subgrid size {self.xM_size}!")
        if (self.xM_size * self.yN_size) % self.N != 0:
            raise ValueError(
                f"Contribution size not integer with image size {self.N}, "
                f"subgrid size {self.xM_size} and facet size {self.yN_size}!"
            )

    @property
    def subgrid_off_step(self):
        """All subgrid offsets must be divisible by this (core.py:76-83)."""
        return self.N // self.yN_size

    @property
    def facet_off_step(self):
        """All facet offsets must be divisible by this (core.py:85-92)."""
        return self.N // self.xM_size

    @property
    def device(self):
        """torch device the core computes on"""
        return self._device

    def __repr__(self):
        return (
            f"{self.__class__.__name__}(W={self.W}, N={self.N}, "
            f"xM_size={self.xM_size}, yN_size={self.yN_size})"
        )

    # ------------------------------------------------------------------ marshalling
    def _as_device(self, arr, like_dtype=None):
        """Return (complex CUDA tensor, was_numpy)."""
        torch = _torch()
        was_numpy = not isinstance(arr, torch.Tensor)
        if was_numpy:
            arr = numpy.asarray(arr)
            if not numpy.iscomplexobj(arr):
                arr = arr.astype(numpy.complex64 if arr.dtype == numpy.float32 else numpy.complex128)
            elif arr.dtype not in (numpy.complex64, numpy.complex128):
                arr = arr.astype(numpy.complex128)
            ten = torch.from_numpy(numpy.ascontiguousarray(arr)).to(self._device)
        else:
            ten = arr
            if not ten.is_complex():
                ten = ten.to(torch.complex64 if ten.dtype == torch.float32 else torch.complex128)
            if ten.device != self._device:
                ten = ten.to(self._device)
        if like_dtype is not None and ten.dtype != like_dtype:
            ten = ten.to(like_dtype)
        return ten, was_numpy

    @staticmethod
    def _code(ten):
        torch = _torch()
        return _lib.C64 if ten.dtype == torch.complex64 else _lib.C128

    def _raw_stream(self):
        """raw handle (integer) of the calling thread's current stream on this core's device.  (``torch.cuda.current_stream``
        builds a Stream object on every call -- 5 us, ~200 calls per pass of a multi-GPU rank; the private accessor is
        what torch's own kernel launchers use.)"""
        fast = getattr(_torch()._C, "_cuda_getCurrentRawStream", None)  # pylint: disable=protected-access
        if fast is not None:
            return int(fast(self._device_index))
        return int(_torch().cuda.current_stream(self._device).cuda_stream)

    def _stream(self):
        return ctypes.c_void_p(self._raw_stream())

    def _real_vec(self, vec, cdtype, size):
        """Optional real window (mask) as a device vector of the precision that
        matches the complex dtype."""
        if vec is None:
            return None
        torch = _torch()
        rdtype = torch.float32 if cdtype == torch.complex64 else torch.float64
        if not isinstance(vec, torch.Tensor):
            host = numpy.ascontiguousarray(numpy.asarray(vec, dtype=float))
            if host.size != size:
                raise ValueError(f"Mask has {host.size} elements, expected {size}!")
            # cached by content: a pageable upload is ordered behind all queued work (one bubble per call otherwise;
            # the streaming classes pass the same few facet / subgrid masks over and over)
            key = (str(rdtype), host.tobytes())
            cache = self.__dict__.setdefault("_mask_cache", {})
            hit = cache.get(key)
            if hit is None:
                if len(cache) >= 1024:
                    cache.pop(next(iter(cache)))
                hit = cache[key] = torch.from_numpy(host).to(device=self._device, dtype=rdtype).contiguous()
            return hit
        vec = vec.to(device=self._device, dtype=rdtype).contiguous()
        if vec.numel() != size:
            raise ValueError(f"Mask has {vec.numel()} elements, expected {size}!")
        return vec

    def _axis_call(self, fname, arr, in_size, out_size, axis, out, accumulate, off, size_arg=None, mask=None):
        """Run one single-axis primitive.

        ``in_size`` (or None) is the length the transform axis must have,
        ``size_arg`` the facet / subgrid size parameter some entry points take.
        """
        torch = _torch()
        out_dtype = None
        if out is not None:
            out_dtype = (
                out.dtype
                if isinstance(out, torch.Tensor)
                else {numpy.dtype(numpy.complex64): torch.complex64}.get(numpy.asarray(out).dtype, torch.complex128)
            )
        xin, was_numpy = self._as_device(arr, out_dtype)
        dims = xin.dim()
        if dims == 1:
            x2 = xin.unsqueeze(0)
            ax = 1
        elif dims == 2:
            if axis not in (0, 1):
                raise ValueError(f"Invalid axis {axis} for shape {tuple(xin.shape)}!")
            x2 = xin
            ax = axis
        else:
            raise ValueError(f"Invalid number of dimensions in input array: {dims}")
        if in_size is not None and x2.shape[ax] != in_size:
            raise ValueError(f"Input has size {x2.shape[ax]} along axis {axis}, expected {in_size}!")
        shape2 = list(x2.shape)
        shape2[ax] = out_size
        shape = tuple(shape2) if dims == 2 else (out_size,)

        out_numpy = None
        if out is None:
            o_t = (torch.zeros if accumulate else torch.empty)(shape, dtype=xin.dtype, device=self._device)
        else:
            if tuple(out.shape) != shape:
                raise ValueError(f"Output array has shape {tuple(out.shape)}, expected {shape}!")
            if isinstance(out, torch.Tensor):
                if out.device != self._device or not out.is_complex():
                    raise ValueError("out= tensor must be a complex tensor on the core's device")
                o_t = out
            else:
                out_numpy = out
                o_t = (
                    torch.from_numpy(numpy.ascontiguousarray(out)).to(self._device)
                    if accumulate
                    else torch.empty(shape, dtype=xin.dtype, device=self._device)
                )
        o2 = o_t.unsqueeze(0) if dims == 1 else o_t
        if any(s < 0 for s in x2.stride()) or any(s < 0 for s in o2.stride()):
            raise ValueError("negative strides are not supported")
        other = 1 - ax
        rows = x2.shape[other]
        args = [self._handle, self._code(xin), ctypes.c_void_p(x2.data_ptr()), rows]
        if fname in ("prepare_facet", "prepare_subgrid"):
            args.append(size_arg)
        args += [x2.stride(other), x2.stride(ax), ctypes.c_void_p(o2.data_ptr()), o2.stride(other), o2.stride(ax), int(off)]
        keep = None
        if fname in ("finish_subgrid", "finish_facet"):
            keep = self._real_vec(mask, xin.dtype, size_arg)
            args += [size_arg, ctypes.c_void_p(keep.data_ptr()) if keep is not None else None]
        args.append(self._stream())
        _lib.check(getattr(self._lib, "swiftly_hip_" + fname)(*args))
        del keep
        if out_numpy is not None:
            out_numpy[...] = o_t.cpu().numpy()
            return out_numpy
        if was_numpy and out is None:
            return o_t.cpu().numpy()
        return o_t


    # ------------------------------------------------------------------ raw batched launches
    def launch(
        self, fname, src, rows, in_rs, in_cs, dst, out_rs, out_cs, off=0, *, size=None, mask=None,
        nbatch=1, in_bs=0, out_bs=0, offs=None, mask_bs=0,
    ):
        """Enqueue one (batched) primitive on device tensors -- the zero-copy
        path of the streaming classes.  ``fname`` is the ABI entry point without
        the ``swiftly_hip_`` prefix and ``_batch`` suffix; strides are in
        complex elements; ``offs`` is a sequence of per-item offsets or None.
        See include/swiftly_hip.h for the batch contract."""
        cvp = ctypes.c_void_p
        args = [self._handle, self._code(src), cvp(src.data_ptr()), int(rows)]
        if fname == "prepare_subgrid":
            args.append(int(size))
        args += [int(in_rs), int(in_cs), cvp(dst.data_ptr()), int(out_rs), int(out_cs), int(off)]
        if fname in ("finish_subgrid", "finish_facet"):
            args += [int(size), cvp(mask.data_ptr()) if mask is not None else None]
        offs_arr = None
        if offs is not None:
            offs_arr = (ctypes.c_int64 * int(nbatch))(*[int(o) for o in offs])
        args += [int(nbatch), int(in_bs), int(out_bs), offs_arr]
        if fname in ("finish_subgrid", "finish_facet"):
            args.append(int(mask_bs))
        args.append(self._stream())
        _lib.check(getattr(self._lib, f"swiftly_hip_{fname}_batch")(*args))

    def add_to_subgrid_from_columns(self, cols, facet_off0, colacc, subgrid_off1s):
        """K3 + K4a fused: ``colacc[f, b] += add_to_subgrid(extract_from_facet(cols[f], off1_b, axis=1),
        facet_off0, axis=0)`` for the facets ``cols[F', m, yN]`` (all with the same ``off0``) and the
        subgrids of a wave; ``colacc[F', S, xM, m]``.  NotImplementedError where unavailable."""
        F, S = cols.shape[0], colacc.shape[1]
        offs = (ctypes.c_int64 * S)(*[int(o) for o in subgrid_off1s])
        _lib.check(
            self._lib.swiftly_hip_add_to_subgrid_from_columns(
                self._handle, self._code(cols), ctypes.c_void_p(cols.data_ptr()), cols.stride(1), cols.stride(0), F,
                ctypes.c_void_p(colacc.data_ptr()), colacc.stride(2), colacc.stride(1), int(facet_off0), S, offs,
                self._stream(),
            )
        )
        return colacc

    def sum_finish_rows(self, colacc, group_off1s, out, subgrid_off1s, subgrid_size, mask=None):
        """Fused axis-1 half of ``sum_and_finish_subgrid`` (reference
        api_helper.py:96-112) for a wave: ``colacc[G, S, xM, m]`` (per off1
        group, summed along axis 0 already) -> ``out[S, xM, subgrid_size]``;
        raises NotImplementedError where the fused kernel is not available."""
        G, S = colacc.shape[0], colacc.shape[1]
        goffs = (ctypes.c_int64 * G)(*[int(o) for o in group_off1s])
        soffs = (ctypes.c_int64 * S)(*[int(o) for o in subgrid_off1s])
        _lib.check(
            self._lib.swiftly_hip_sum_finish_rows(
                self._handle, self._code(colacc), ctypes.c_void_p(colacc.data_ptr()), G, colacc.stride(0),
                colacc.stride(1), colacc.stride(2), goffs, ctypes.c_void_p(out.data_ptr()), out.stride(0),
                out.stride(1), soffs, int(subgrid_size),
                ctypes.c_void_p(mask.data_ptr()) if mask is not None else None,
                mask.stride(0) if mask is not None else 0, S, self._stream(),
            )
        )
        return out

    def extract_column(self, BF_F, subgrid_off0, facet_off1, out=None, rowmap=None, prewindowed=False):
        """``prepare_facet(extract_from_facet(BF_F, subgrid_off0, axis=0),
        facet_off1, axis=1)`` (reference api_helper.py:200-210) as one kernel on
        a device tensor ``BF_F[yN_size, facet_size]`` -> ``[xM_yN_size,
        yN_size]``.  With ``rowmap`` (int32 device tensor of length yN_size,
        see :py:meth:`prepare_facet_rows`) ``BF_F`` is the row-compacted form."""
        torch = _torch()
        if not isinstance(BF_F, torch.Tensor):
            res = self.extract_column(self._as_device(BF_F)[0], subgrid_off0, facet_off1)
            return res.cpu().numpy()
        if BF_F.dim() != 2 or (rowmap is None and BF_F.shape[0] != self.yN_size):  # compacted when rowmap is given
            raise ValueError(f"BF_F must have shape [{self.yN_size}, facet_size], got {tuple(BF_F.shape)}")
        if out is None:
            out = torch.empty((self.xM_yN_size, self.yN_size), dtype=BF_F.dtype, device=self._device)
        elif tuple(out.shape) != (self.xM_yN_size, self.yN_size):
            raise ValueError(f"Output array has shape {tuple(out.shape)}, expected {(self.xM_yN_size, self.yN_size)}!")
        args = [
            self._handle, self._code(BF_F), ctypes.c_void_p(BF_F.data_ptr()), int(BF_F.shape[1]),
            BF_F.stride(0), BF_F.stride(1), ctypes.c_void_p(out.data_ptr()), out.stride(0), out.stride(1),
            int(subgrid_off0), int(facet_off1),
        ]
        if rowmap is None and not prewindowed:
            _lib.check(self._lib.swiftly_hip_extract_column(*args, self._stream()))
        else:
            _lib.check(
                self._lib.swiftly_hip_extract_column_rows(
                    *args, ctypes.c_void_p(rowmap.data_ptr()) if rowmap is not None else None, int(bool(prewindowed)),
                    self._stream(),
                )
            )
        return out

    def subgrid_column_rows(self, subgrid_off0s):
        """Row map for a sparse set of subgrid columns: which rows of
        ``BF_F = prepare_facet(facet, off0, axis=0)`` does
        ``extract_from_facet(., off0, axis=0)`` read for the given subgrid
        ``off0`` values (core.py:243-253).  Returns ``(rowmap int32 device
        tensor [yN_size] with -1 for unused rows, number of rows kept)``."""
        torch = _torch()
        yN, m = self.yN_size, self.xM_yN_size
        # cached per core: the upload is a pageable host-to-device copy, which is ordered behind everything already
        # queued on the device (a pipeline bubble per wave if it were repeated every pass)
        key = tuple(sorted(set(int(o) for o in subgrid_off0s)))
        cache = self.__dict__.setdefault("_rowmap_cache", {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        keep = numpy.zeros(yN, dtype=bool)
        for off0 in key:
            s = off0 * yN // self.N
            keep[(yN // 2 - m // 2 + numpy.arange(m) + s) % yN] = True
        rowmap = numpy.full(yN, -1, dtype=numpy.int32)
        rowmap[keep] = numpy.arange(int(keep.sum()), dtype=numpy.int32)
        if len(cache) >= 4096:
            cache.pop(next(iter(cache)))
        cache[key] = (torch.from_numpy(rowmap).to(self._device), int(keep.sum()))
        return cache[key]

    def prepare_facet_rows(self, facet, facet_off, rowmap, n_rows, out=None, fold_axis1_window=False):
        """``prepare_facet(facet, facet_off, axis=0)`` keeping only the rows
        ``rowmap`` selects (device tensors only): output ``[n_rows,
        facet.shape[1]]``; rows no requested subgrid column reads are never
        written, which removes their share of the HBM traffic and footprint.
        ``rowmap=None`` keeps all ``yN_size`` rows.  ``fold_axis1_window`` also
        multiplies column c by the 1/PSWF window of ``prepare_facet(., axis=1)``
        (windows commute with the transform along the other axis), to be paired
        with ``extract_column(..., prewindowed=True)``."""
        torch = _torch()
        if facet.dim() != 2:
            raise ValueError("prepare_facet_rows needs a 2-D device tensor")
        if out is None:
            out = torch.empty((n_rows, facet.shape[1]), dtype=facet.dtype, device=self._device)
        _lib.check(
            self._lib.swiftly_hip_prepare_facet_rows(
                self._handle, self._code(facet), ctypes.c_void_p(facet.data_ptr()), int(facet.shape[1]),
                int(facet.shape[0]), facet.stride(1), facet.stride(0), ctypes.c_void_p(out.data_ptr()),
                out.stride(1), out.stride(0), int(facet_off),
                ctypes.c_void_p(rowmap.data_ptr()) if rowmap is not None else None, int(bool(fold_axis1_window)),
                self._stream(),
            )
        )
        return out

    # ------------------------------------------------------------------ contiguous-axis-first pipeline
    def _logs(self, need=("yN", "xM", "m")):
        """log2 of the transform lengths named in ``need``; None unless all of THOSE are powers of two"""
        logs = {}
        for name, n in (("yN", self.yN_size), ("xM", self.xM_size), ("m", self.xM_yN_size)):
            if name not in need:
                continue
            if n <= 0 or n & (n - 1):
                return None
            logs[name] = n.bit_length() - 1
        return logs

    def _mixed_yN(self):
        """``(Q, k)`` when ``yN_size = Q * 2^k`` with Q in {3, 5, 7, 9} (:py:func:`mixed_factor`), else None"""
        return mixed_factor(self.yN_size)

    MAX_FUSED_FACETS = 64  # kSumFinishMaxFacets (csrc/swiftly_sumfinish.h): facets summed by one sum_finish_facets call

    def supports_fused_subgrid(self, dtype=None, n_facets=None):
        """True when transform_contributions + sum_finish_facets (include/swiftly_hip.h) exist for these sizes
        (and, when given, for ``n_facets`` facets: the facet sum runs inside one kernel)."""
        torch = _torch()
        logs = self._logs(("xM", "m"))
        if logs is None or (dtype is not None and dtype != torch.complex64):
            return False
        if n_facets is not None and n_facets > self.MAX_FUSED_FACETS:
            return False
        pairs = {(7, 8), (7, 10), (8, 9), (8, 10), (9, 10), (9, 11), (10, 11), (10, 12)}  # sum_finish instances
        return logs["m"] <= 10 and (logs["m"], logs["xM"]) in pairs  # m: single-pass column transform

    def supports_band_pipeline(self, dtype=None, n_facets=None):
        """True when the contiguous-axis-first forward kernels (include/swiftly_hip.h) exist for these sizes."""
        # K1: the two-workgroup band kernel for yN = 16384 .. 65536 (band-pruned output), the generic contiguous-axis
        # transform below that (whole padded axis kept)
        # yN = Q * 2^k (r3): the radix-Q pass in front of the same kernels, whole padded axis kept, forward only
        if not self.supports_fused_subgrid(dtype, n_facets) or self._logs(("m",))["m"] < 6:
            return False
        logs = self._logs(("yN",))
        if logs is not None:
            return 6 <= logs["yN"] <= 16
        mixed = self._mixed_yN()
        return mixed is not None and 6 <= mixed[1] <= 15

    def supports_backward_band(self, dtype=None):
        """True when accumulate_facet_columns / finish_facet_band (include/swiftly_hip.h) exist for these sizes."""
        torch = _torch()
        if dtype is not None and dtype != torch.complex64:
            return False
        logs = self._logs()
        if logs is not None:
            return 2 <= logs["yN"] <= 18
        # yN = Q * 2^k (r3): radix-Q pass with the gather-sum load + column-tile sub-transforms, plain band layout
        mixed = self._mixed_yN()
        return self._logs(("xM", "m")) is not None and mixed is not None and 6 <= mixed[1] <= 15

    def band_for_offsets(self, subgrid_offs):
        """Smallest cyclic range ``(start, length)`` of centred indices of the padded facet axis that contains
        the ``xM_yN_size`` window of every given subgrid offset (core.py:243-253); ``(0, yN_size)`` = all."""
        logs = self._logs(("yN",))
        if logs is None or not 14 <= logs["yN"] <= 16:
            return 0, self.yN_size  # short / non-power-of-two padded facets keep the whole axis (plain band layout)
        return band_range(self.N, self.yN_size, self.xM_yN_size, subgrid_offs)

    def band_columns(self, band):
        """physical columns of a band buffer"""
        return int(self._lib.swiftly_hip_band_columns_for(self._handle, int(band[1])))

    def prepare_facet_band(self, facet, facet_off, band, out=None, fold_other_axis_window=True, rows_of=None):
        """K1 (contiguous axis first): ``prepare_facet(facet, facet_off, axis=1)`` for every row of a row-major
        device facet, keeping only the band of output columns (parity-split layout, see include/swiftly_hip.h),
        times the 1/PSWF window of axis 0 when ``fold_other_axis_window``.  ``rows_of=(size, row0)``: ``facet`` is
        the block of rows ``[row0, row0 + facet.shape[0])`` of a facet with ``size`` rows -- the axis-0 window is that
        facet's (``swiftly_hip_prepare_facet_band_rows``)."""
        torch = _torch()
        if facet.dim() != 2 or facet.stride(1) != 1:
            raise ValueError("prepare_facet_band needs a row-major 2-D device tensor")
        ncols = self.band_columns(band)
        if out is None:
            out = torch.empty((facet.shape[0], ncols), dtype=facet.dtype, device=self._device)
        elif tuple(out.shape) != (facet.shape[0], ncols) or out.stride(1) != 1:
            raise ValueError(f"Output array has shape {tuple(out.shape)}, expected {(facet.shape[0], ncols)}!")
        if rows_of is not None:
            _lib.check(
                self._lib.swiftly_hip_prepare_facet_band_rows(
                    self._handle, self._code(facet), ctypes.c_void_p(facet.data_ptr()), int(facet.shape[0]),
                    int(facet.shape[1]), facet.stride(0), ctypes.c_void_p(out.data_ptr()), out.stride(0), int(facet_off),
                    int(band[0]), int(band[1]), int(rows_of[0]) if fold_other_axis_window else 0, int(rows_of[1]),
                    self._stream(),
                )
            )
            return out
        _lib.check(
            self._lib.swiftly_hip_prepare_facet_band(
                self._handle, self._code(facet), ctypes.c_void_p(facet.data_ptr()), int(facet.shape[0]),
                int(facet.shape[1]), facet.stride(0), ctypes.c_void_p(out.data_ptr()), out.stride(0), int(facet_off),
                int(band[0]), int(band[1]), int(bool(fold_other_axis_window)), self._stream(),
            )
        )
        return out

    #: physical band columns the window-rows epilogue of the whole-row K1 can stage (row_whole.hip, row_pass_whole_stage_columns)
    WINDOW_ROWS_STAGE_COLUMNS = 12800

    def window_starts(self, band, wave_off1s):
        """``(first logical column of the contribution window of wave off1 - band start) mod yN`` for every wave: the
        ``window_starts`` table of ``swiftly_hip_prepare_facet_window_rows`` (host list)."""
        m, yN = self.xM_yN_size, self.yN_size
        return [((yN // 2 - m // 2 + int(o) * yN // self.N) - int(band[0])) % yN for o in wave_off1s]

    def supports_window_rows(self, band, facet_size, facet_off1s, n_windows=1):
        """can K1 finish the contiguous axis for every planned window in its epilogue (``prepare_facet_window_rows``)?
        (yN = 32768, m = 512, xM <= 2048 for the placed subgrid side, the band fits the LDS stage, at most 256 windows,
        16-byte loads possible)"""
        return (
            0 < int(n_windows) <= 256 and self.yN_size == 32768 and self.xM_yN_size == 512 and self.xM_size <= 2048 and
            self.band_columns(band) <= self.WINDOW_ROWS_STAGE_COLUMNS and int(band[1]) < self.yN_size and
            int(facet_size) % 2 == 0 and all(int(o) % 2 == 0 for o in facet_off1s)
        )

    def prepare_facet_window_rows(self, facet, facet_off, band, window_starts, out, fold_other_axis_window=True,
                                  rows_of=None):
        """K1 of the axis-1-first pipeline with the COMPLETE contiguous-axis finish in its epilogue
        (``swiftly_hip_prepare_facet_window_rows``, whole-row kernel): ``out[w, row, :]`` = what ``finish_axis1_rows`` gives
        for wave ``w`` (the parity-split window band of ``Fn * cfft_m``).  ``window_starts``: int32 DEVICE tensor
        (``window_starts(band, off1s)``); ``out``: ``[nwin, rows, m]`` (wave-major: K2 of a wave reads one contiguous block)
        or ``[rows, nwin * m]`` (the windows of a row side by side).  Hand ``out[w]`` (or the column block) to K2 with the
        band ``(window start, m)`` and run ``wave_subgrid_side(..., placed=True)``."""
        m = self.xM_yN_size
        nwin = int(window_starts.numel())
        if facet.dim() != 2 or facet.stride(1) != 1 or out.stride(-1) != 1:
            raise ValueError("prepare_facet_window_rows needs row-major device tensors")
        if out.dim() == 3 and tuple(out.shape) == (nwin, facet.shape[0], m):
            row_stride, win_stride = out.stride(1), out.stride(0)
        elif out.dim() == 2 and tuple(out.shape) == (facet.shape[0], nwin * m):
            row_stride, win_stride = out.stride(0), m
        else:
            raise ValueError(f"Output array has shape {tuple(out.shape)}, expected {(nwin, facet.shape[0], m)} or "
                             f"{(facet.shape[0], nwin * m)}!")
        size, row0 = rows_of if rows_of is not None else (facet.shape[0], 0)
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_prepare_facet_window_rows(
                self._handle, self._code(facet), cvp(facet.data_ptr()), int(facet.shape[0]), int(facet.shape[1]),
                facet.stride(0), cvp(out.data_ptr()), row_stride, int(facet_off), int(band[0]), int(band[1]),
                int(size) if fold_other_axis_window else 0, int(row0), cvp(window_starts.data_ptr()), nwin, win_stride,
                self._stream(),
            )
        )
        return out

    def prepare_facet_columns(self, bands, facet_off0s, band, subgrid_off1, rowmap=None, n_rows=None, out=None):
        """K2 (contiguous axis first): for every facet ``f``, gather the window of subgrid offset
        ``subgrid_off1`` from the band buffer ``bands[f]`` (``[F, yB, band columns]``) and run
        ``prepare_facet(., facet_off0s[f], axis=0)`` without its (pre-applied) window on it; keeps the rows
        ``rowmap`` selects -> ``[F, n_rows, xM_yN_size]``."""
        torch = _torch()
        F, yB = bands.shape[0], bands.shape[1]
        m = self.xM_yN_size
        n_rows = self.yN_size if rowmap is None else int(n_rows)
        if out is None:
            out = torch.empty((F, n_rows, m), dtype=bands.dtype, device=self._device)
        # one wave through the multi-wave entry point: it takes the caller-owned four-step scratch (a stream-ordered
        # allocation per call costs host time, see swiftly_hip.h)
        scr = self.scratch("k2", self._k2_scratch_bytes(F))
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_prepare_facet_columns_waves(
                self._handle, self._code(bands), cvp(bands.data_ptr()), int(yB), bands.stride(1), bands.stride(0), F,
                self._i64(facet_off0s), int(band[0]), int(band[1]), 1, self._i64([subgrid_off1]), cvp(out.data_ptr()),
                out.stride(1), out.stride(0), 0, cvp(rowmap.data_ptr()) if rowmap is not None else None, 0,
                cvp(scr.data_ptr()), scr.numel(), self._stream(),
            )
        )
        return out

    def transform_contributions(self, src, layout, facet_off0s, subgrid_offs, out=None, rowmap=None, band=None, nsub=None):
        """K3 + K4a: ``out[f, b] = Fn * cfft_m(contribution_{f,b}, axis 0)`` rotated by the facet's ``off0``
        (add_to_subgrid along axis 0 without the placement) with the contribution gathered on load from
        ``src`` -- layout 0: ``[F, m, yN|band]`` column buffers, windows by subgrid ``off1``; layout 1:
        ``[F, rows, m]`` (prepare_facet_columns), windows by subgrid ``off0`` through ``rowmap``; layout 2:
        ``[F, S, m, m]`` materialised contributions."""
        torch = _torch()
        F = src.shape[0]
        m = self.xM_yN_size
        S = int(nsub) if layout == 2 else len(subgrid_offs)
        if out is None:
            out = torch.empty((F, S, m, m), dtype=src.dtype, device=self._device)
        foffs = (ctypes.c_int64 * F)(*[int(o) for o in facet_off0s])
        soffs = (ctypes.c_int64 * S)(*[int(o) for o in subgrid_offs]) if layout != 2 else None
        if layout == 2:
            row_stride, facet_stride, sub_stride = src.stride(2), src.stride(0), src.stride(1)
        else:
            row_stride, facet_stride, sub_stride = src.stride(1), src.stride(0), 0
        band = band or (0, 0)
        _lib.check(
            self._lib.swiftly_hip_transform_contributions(
                self._handle, self._code(src), ctypes.c_void_p(src.data_ptr()), int(layout), row_stride, facet_stride,
                sub_stride, ctypes.c_void_p(rowmap.data_ptr()) if rowmap is not None else None, int(band[0]),
                int(band[1]), F, foffs, S, soffs, ctypes.c_void_p(out.data_ptr()), out.stride(0), out.stride(1),
                self._stream(),
            )
        )
        return out

    _I64_CACHE = {}

    @classmethod
    def _i64(cls, values):
        """ctypes int64 array of ``values`` (memoised by content: the per-wave offset lists repeat every pass; the native
        side only reads them)"""
        key = tuple(values)
        arr = cls._I64_CACHE.get(key)
        if arr is None:
            if len(cls._I64_CACHE) >= 8192:
                cls._I64_CACHE.clear()
            arr = cls._I64_CACHE[key] = (ctypes.c_int64 * len(key))(*[int(v) for v in key])
        return arr

    def _k2_scratch_bytes(self, F):
        """four-step scratch of K2 for F facets; yN = Q * 2^k also holds the output of the radix-Q pass"""
        n = F * self.yN_size * self.xM_yN_size * 8
        return n if self._mixed_yN() is None else 2 * n + 4096

    SCRATCH_TAIL_BYTES = 1 << 16

    def scratch(self, name, nbytes):
        """Grow-only persistent device scratch of this core, by name (four-step intermediates of the native wave
        calls; stream-ordered allocations of changing size cost ~2 ms of host time each)."""
        torch = _torch()
        pool = self.__dict__.setdefault("_scratch_pool", {})
        # one buffer per HIP stream: calls on one stream are ordered, calls on different streams (other host
        # threads, side streams) must not share a scratch
        key = (name, self._raw_stream())
        # + room for the arrival counters of the fused four-step launches (one word per batch item and column tile)
        nbytes = int(nbytes) + self.SCRATCH_TAIL_BYTES
        buf = pool.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = pool[key] = torch.empty((nbytes,), dtype=torch.uint8, device=self._device)
        return buf

    def wave_facet_side(self, bands, facet_off0s, band, wave_off1, rowmap, n_rows, Q, compute_q, sub_off0s, g_out,
                        g_layout=None):
        """K2 + K3 + K4a of one wave natively (``swiftly_hip_wave_facet_side``): ``bands[F, yB, band columns]`` ->
        workspace ``Q[F, n_rows, m]`` (computed when ``compute_q``) -> blocks written into ``g_out``: a
        ``[F, S, m, m]`` tensor, or a flat send buffer with ``g_layout = (offsets[S], facet_strides[S])``."""
        F, S = Q.shape[0], len(sub_off0s)
        cvp = ctypes.c_void_p
        scr = self.scratch("k2", self._k2_scratch_bytes(F)) if compute_q else None
        if g_layout is None:
            fs, ss, offs, fstr = g_out.stride(0), g_out.stride(1), None, None
        else:
            fs, ss, offs, fstr = 0, 0, self._i64(g_layout[0]), self._i64(g_layout[1])
        _lib.check(
            self._lib.swiftly_hip_wave_facet_side(
                self._handle, self._code(Q), cvp(bands.data_ptr()) if bands is not None else None,
                int(bands.shape[1]) if bands is not None else 0, bands.stride(1) if bands is not None else 0,
                bands.stride(0) if bands is not None else 0, F, self._i64(facet_off0s), int(band[0]), int(band[1]),
                int(wave_off1), cvp(rowmap.data_ptr()) if rowmap is not None else None, int(n_rows),
                cvp(Q.data_ptr()), Q.stride(0), int(bool(compute_q)), S, self._i64(sub_off0s), cvp(g_out.data_ptr()), fs,
                ss, offs, fstr, cvp(scr.data_ptr()) if scr is not None else None, scr.numel() if scr is not None else 0,
                self._stream(),
            )
        )

    def chain_chunk_streams(self, chain):
        """``swiftly_hip_chain_chunk_streams`` for the calling thread: the next chunked strided-axis transforms skip the
        fork of their two internal streams (the caller vouches that their inputs were complete before an earlier,
        forking call; include/swiftly_hip.h)"""
        self._lib.swiftly_hip_chain_chunk_streams(1 if chain else 0)

    def side_stream(self):
        """second HIP stream of this core (bandwidth-bound work issued next to an issue-bound kernel)"""
        st = self.__dict__.get("_side_stream")
        if st is None:
            # (default priority: a high-priority side stream was measured slower, 39.5-39.8 -> 40.0-40.2 ms per pass, r4)
            st = self.__dict__["_side_stream"] = _torch().cuda.Stream(device=self._device)
        return st

    def stacked_rowmaps(self, key, maps):
        """the row maps of several waves as one device tensor ``[W, yN]`` (cached per key)"""
        cache = self.__dict__.setdefault("_rowmap_stack_cache", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 64:
                cache.pop(next(iter(cache)))
            hit = cache[key] = _torch().stack(list(maps)).contiguous()
        return hit

    def prepare_facet_columns_waves(self, bands, facet_off0s, band, wave_off1s, out, rowmaps=None, workspace=None):
        """K2 for the facets ``bands[F', yB, band columns]`` and ALL given waves at once:
        ``out[F', W, rows, m]`` (``rows`` = the largest kept-row count), row maps ``rowmaps[W, yN]`` int32 or None."""
        F, W = bands.shape[0], len(wave_off1s)
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_prepare_facet_columns_waves(
                self._handle, self._code(bands), cvp(bands.data_ptr()), int(bands.shape[1]), bands.stride(1),
                bands.stride(0), F, self._i64(facet_off0s), int(band[0]), int(band[1]), W, self._i64(wave_off1s),
                cvp(out.data_ptr()), out.stride(2), out.stride(0), out.stride(1),
                cvp(rowmaps.data_ptr()) if rowmaps is not None else None,
                rowmaps.stride(0) if rowmaps is not None else 0,
                cvp(workspace.data_ptr()) if workspace is not None else None,
                workspace.numel() * workspace.element_size() if workspace is not None else 0, self._stream(),
            )
        )
        return out

    def wave_subgrid_side(self, G, facet_off0s, facet_off1s, sub_off0s, sub_off1s, subgrid_size, mask0, mask1, tmp, out,
                          placed=False):
        """K4b + K5 of one wave natively (``swiftly_hip_wave_subgrid_side``): ``G[F, S, m, m]`` -> ``out[S, xA, xA]``
        through the workspace ``tmp[S, xM, xA]``.  ``placed``: the blocks come from the axis-1-first pipeline
        (``finish_axis1_rows`` ran before K2): their rows already are ``Fn * cfft_m`` along the contiguous axis."""
        F, S = G.shape[0], G.shape[1]
        cvp = ctypes.c_void_p
        scr = self.scratch("k5b", min(S, 64) * self.xM_size * int(subgrid_size) * 8)
        # axis-1-first pipeline: the rows of G already are Fn * cfft_m along the contiguous axis (finish_axis1_rows)
        entry = self._lib.swiftly_hip_wave_subgrid_side_placed if placed else self._lib.swiftly_hip_wave_subgrid_side
        _lib.check(
            entry(
                self._handle, self._code(G), cvp(G.data_ptr()), F, G.stride(0), G.stride(1), self._i64(facet_off0s),
                self._i64(facet_off1s), S, self._i64(sub_off0s), self._i64(sub_off1s), int(subgrid_size),
                cvp(mask0.data_ptr()) if mask0 is not None else None, mask0.stride(0) if mask0 is not None else 0,
                cvp(mask1.data_ptr()) if mask1 is not None else None, mask1.stride(0) if mask1 is not None else 0,
                cvp(tmp.data_ptr()), cvp(out.data_ptr()), cvp(scr.data_ptr()), scr.numel(), self._stream(),
            )
        )
        return out

    def finish_axis1_rows(self, bands, facet_off1s, band, wave_off1, out=None):
        """Step R of the axis-1-first pipeline (``swiftly_hip_finish_axis1_rows``): the contiguous-axis half of
        ``add_to_subgrid`` (reference core.py:255-285) on every row of the K1 band buffers ``bands[F, yB, band columns]``
        for the wave ``wave_off1``, BEFORE the strided-axis transforms.  Returns ``(W[F, yB, m], window band)``: hand both
        to ``prepare_facet_columns`` / ``wave_facet_side`` in place of the band buffers and their band."""
        torch = _torch()
        F, rows = bands.shape[0], bands.shape[1]
        m, yN = self.xM_yN_size, self.yN_size
        if out is None:
            out = torch.empty((F, rows, m), dtype=bands.dtype, device=bands.device)
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_finish_axis1_rows(
                self._handle, self._code(bands), cvp(bands.data_ptr()), rows, bands.stride(1), bands.stride(0), F,
                self._i64(facet_off1s), int(band[0]), int(band[1]), int(wave_off1), cvp(out.data_ptr()), out.stride(1),
                out.stride(0), self._stream(),
            )
        )
        s = int(wave_off1) * yN // self.N
        return out, ((yN // 2 - m // 2 + s) % yN, m)

    def sum_finish_facets(self, G, facet_off0s, facet_off1s, out, subgrid_off1s, subgrid_size, mask=None):
        """K4b + K5a: sum over facets + axis-1 finish of ``G[F, S, m, m]`` (transform_contributions) ->
        ``out[S, xM, subgrid_size]`` (see include/swiftly_hip.h)."""
        F, S = G.shape[0], G.shape[1]
        f0 = (ctypes.c_int64 * F)(*[int(o) for o in facet_off0s])
        f1 = (ctypes.c_int64 * F)(*[int(o) for o in facet_off1s])
        so = (ctypes.c_int64 * S)(*[int(o) for o in subgrid_off1s])
        _lib.check(
            self._lib.swiftly_hip_sum_finish_facets(
                self._handle, self._code(G), ctypes.c_void_p(G.data_ptr()), F, G.stride(0), G.stride(1), G.stride(2),
                f0, f1, ctypes.c_void_p(out.data_ptr()), out.stride(0), out.stride(1), so, int(subgrid_size),
                ctypes.c_void_p(mask.data_ptr()) if mask is not None else None,
                mask.stride(0) if mask is not None else 0, S, self._stream(),
            )
        )
        return out

    def split_prepare_facets(self, tmp, subgrid_off1s, facet_off0s, facet_off1s, out):
        """``tmp[S, xM, xA]`` (prepare_subgrid along axis 0) -> contributions ``out[F, S, m, m]`` to every facet
        (``swiftly_hip_split_prepare_facets``: the rest of api_helper.prepare_and_split_subgrid)."""
        S, F = tmp.shape[0], out.shape[0]
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_split_prepare_facets(
                self._handle, self._code(tmp), cvp(tmp.data_ptr()), tmp.stride(0), tmp.stride(1), int(tmp.shape[2]), S,
                self._i64(subgrid_off1s), F, self._i64(facet_off0s), self._i64(facet_off1s), cvp(out.data_ptr()),
                out.stride(0), out.stride(1), self._stream(),
            )
        )
        return out

    def wave_split_subgrids(self, sub, sub_off0s, sub_off1s, facet_off0s, facet_off1s, work, out):
        """``sub[S, xA, xA]`` (contiguous) -> contributions ``out[F, S, m, m]`` to every facet in one native call
        (``swiftly_hip_wave_split_subgrids``); ``work``: complex workspace of >= ``2 * S * xM * xA`` elements."""
        S, F = sub.shape[0], out.shape[0]
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_wave_split_subgrids(
                self._handle, self._code(sub), cvp(sub.data_ptr()), int(sub.shape[1]), S, self._i64(sub_off0s),
                self._i64(sub_off1s), F, self._i64(facet_off0s), self._i64(facet_off1s), cvp(work.data_ptr()),
                work.numel(), cvp(out.data_ptr()), out.stride(0), out.stride(1), self._stream(),
            )
        )
        return out

    # ------------------------------------------------------------------ backward, contiguous axis last
    GS_MAX_CHUNKS = 16

    def column_row_sources(self, sub_off0s, locations=None):
        """Row tables of the gather-sum load of :py:meth:`accumulate_facet_columns` for a wave of subgrids with
        axis-0 offsets ``sub_off0s`` (``add_to_facet`` along axis 0, core.py:441-478, as a gather): a list of
        ``(subgrid indices, table)`` -- normally one entry; subgrids are split into several groups when more than
        two of them overlap in a padded row.  ``locations[b] = (chunk, block index inside the chunk)`` says where
        the ``[m, m]`` block of subgrid ``b`` sits (default: chunk 0, block ``b``).  Tables are cached on device."""
        torch = _torch()
        offs = tuple(int(o) for o in sub_off0s)
        locs = tuple((0, b) for b in range(len(offs))) if locations is None else tuple((int(c), int(i)) for c, i in locations)
        key = (offs, locs)
        cache = self.__dict__.setdefault("_row_source_cache", {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        groups = build_row_sources(self.N, self.yN_size, self.xM_yN_size, offs, locs, self.GS_MAX_CHUNKS)
        out = [(list(mem), torch.from_numpy(t).to(self._device)) for mem, t in groups]
        if len(cache) >= 512:
            cache.clear()
        cache[key] = out
        return out

    def accumulate_facet_columns(self, parts, part_row_stride, chunk_offsets, chunk_facet_strides, table, facet_off0s,
                                 facet_size, masks, subgrid_off1, bands, band, workspace=None, touched=None):
        """``bands[f] += mask0_f * finish_facet_axis0(sum_b add_to_facet_axis0(C[f][b]))`` placed at the band columns
        of ``subgrid_off1`` (``swiftly_hip_accumulate_facet_columns``, include/swiftly_hip.h).  ``parts``: device
        tensor holding the contribution blocks (chunk offsets / facet strides in elements relative to its start),
        ``bands``: ``[F, facet_size, band length]``, ``masks``: float32 ``[F, facet_size]`` or None, ``touched``: uint8
        ``[band length]`` first-write flags (then ``bands`` may start uninitialised; :py:meth:`band_zero_untouched`)."""
        F = bands.shape[0]
        nch = len(chunk_offsets)
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_accumulate_facet_columns(
                self._handle, self._code(bands), cvp(parts.data_ptr()), int(part_row_stride), nch,
                self._i64(chunk_offsets), self._i64(chunk_facet_strides), cvp(table.data_ptr()), F,
                self._i64(facet_off0s), int(facet_size), cvp(masks.data_ptr()) if masks is not None else None,
                int(subgrid_off1), cvp(bands.data_ptr()), bands.stride(1), bands.stride(0), int(band[0]), int(band[1]),
                cvp(touched.data_ptr()) if touched is not None else None,
                cvp(workspace.data_ptr()) if workspace is not None else None,
                workspace.numel() * workspace.element_size() if workspace is not None else 0, self._stream(),
            )
        )
        return bands

    def band_zero_untouched(self, bands, touched):
        """Clear the band columns no :py:meth:`accumulate_facet_columns` call has written."""
        if not bands.is_contiguous():
            raise ValueError("band accumulators must be contiguous")
        rows = bands.numel() // bands.shape[-1]
        _lib.check(
            self._lib.swiftly_hip_band_zero_untouched(
                self._handle, self._code(bands), ctypes.c_void_p(bands.data_ptr()), rows, bands.shape[-1],
                bands.shape[-1], ctypes.c_void_p(touched.data_ptr()), self._stream(),
            )
        )

    def finish_facet_band(self, band_acc, band, facet_off, facet_size, mask=None, out=None):
        """``finish_facet`` (core.py:481-510) along the contiguous axis for a band accumulator ``[rows, band
        length]`` (``swiftly_hip_finish_facet_band``)."""
        torch = _torch()
        rows = band_acc.shape[0]
        if out is None:
            out = torch.empty((rows, int(facet_size)), dtype=band_acc.dtype, device=self._device)
        mvec = self._real_vec(mask, band_acc.dtype, int(facet_size)) if mask is not None else None
        cvp = ctypes.c_void_p
        _lib.check(
            self._lib.swiftly_hip_finish_facet_band(
                self._handle, self._code(band_acc), cvp(band_acc.data_ptr()), rows, band_acc.stride(0), int(band[0]),
                int(band[1]), cvp(out.data_ptr()), out.stride(0), int(facet_off), int(facet_size),
                cvp(mvec.data_ptr()) if mvec is not None else None, self._stream(),
            )
        )
        return out

    # ------------------------------------------------------------------ facet -> subgrid
    def prepare_facet(self, facet, facet_off, axis, out=None):
        """Window with 1/PSWF, zero-pad to ``yN_size``, shift by ``facet_off``
        and inverse-transform along ``axis`` (core.py:189-222)."""
        size = facet.shape[axis] if len(facet.shape) > 1 else facet.shape[0]
        return self._axis_call(
            "prepare_facet", facet, None, self.yN_size, axis, out, False, facet_off, size_arg=int(size)
        )

    def extract_from_facet(self, prep_facet, subgrid_off, axis, out=None):
        """Cut the ``xM_yN_size`` window of a prepared facet that contributes to
        the subgrid at ``subgrid_off`` (core.py:224-253).  Bit-exact copy."""
        return self._axis_call(
            "extract_from_facet", prep_facet, self.yN_size, self.xM_yN_size, axis, out, False, subgrid_off
        )

    def add_to_subgrid(self, facet_contrib, facet_off, axis, out=None):
        """Transform a contribution, weight with Fn and ADD it at its place in
        the padded subgrid (core.py:255-285)."""
        return self._axis_call(
            "add_to_subgrid", facet_contrib, self.xM_yN_size, self.xM_size, axis, out, True, facet_off
        )

    def add_to_subgrid_2d(self, facet_contrib, facet_off0, facet_off1, out=None):
        """Both axes of :py:meth:`add_to_subgrid` (core.py:752-778)."""
        if len(facet_contrib.shape) != 2:
            raise ValueError(f"Invalid number of dimensions in input array: {len(facet_contrib.shape)}")
        torch = _torch()
        dev, was_numpy = self._as_device(facet_contrib)
        m, xM = self.xM_yN_size, self.xM_size
        if tuple(dev.shape) != (m, m):
            raise ValueError(f"Input has shape {tuple(dev.shape)}, expected {(m, m)}!")
        if isinstance(out, torch.Tensor) or out is None:
            # ONE native call (swiftly_hip_add_to_subgrid_2d = Swiftly.add_to_subgrid_2d of the reference's shim)
            if out is None:
                res = torch.zeros((xM, xM), dtype=dev.dtype, device=self._device)
            else:
                if tuple(out.shape) != (xM, xM):
                    raise ValueError(f"Output array has shape {tuple(out.shape)}, expected {(xM, xM)}!")
                if out.device != self._device or out.dtype != dev.dtype:
                    raise ValueError("out= tensor must be a complex tensor of the input's dtype on the core's device")
                res = out
            if any(st < 0 for st in dev.stride()) or any(st < 0 for st in res.stride()):
                raise ValueError("negative strides are not supported")
            _lib.check(
                self._lib.swiftly_hip_add_to_subgrid_2d(
                    self._handle, self._code(dev), ctypes.c_void_p(dev.data_ptr()), dev.stride(0), dev.stride(1),
                    ctypes.c_void_p(res.data_ptr()), res.stride(0), res.stride(1), int(facet_off0), int(facet_off1),
                    self._stream(),
                )
            )
            if was_numpy and out is None:
                return res.cpu().numpy()
            return res
        tmp = self.add_to_subgrid(dev, facet_off0, axis=0)
        return self.add_to_subgrid(tmp, facet_off1, axis=1, out=out)  # numpy out=: through the accumulate copy-back

    def finish_subgrid(self, summed_contribs, subgrid_off, subgrid_size, out=None, masks=None):
        """Inverse-transform the summed contributions along every axis and cut
        out the subgrid (core.py:287-325).  ``subgrid_off`` is an int for 1-D
        input and a list for 2-D.  ``masks`` (extension, optional list of one
        real vector or None per axis) multiplies the result like
        api_helper.py:107-112 does after the fact."""
        dims = len(summed_contribs.shape)
        if not isinstance(subgrid_off, list):
            if dims != 1:
                raise ValueError("Subgrid offset must be given for every dimension!")
            subgrid_off = [subgrid_off]
        if len(subgrid_off) != dims:
            raise ValueError("Subgrid offset must be given for every dimension!")
        masks = list(masks) if masks is not None else [None] * dims
        if dims == 1:
            return self._axis_call(
                "finish_subgrid", summed_contribs, self.xM_size, subgrid_size, 0, out, False,
                subgrid_off[0], size_arg=int(subgrid_size), mask=masks[0],
            )
        if dims != 2:
            raise ValueError(f"Invalid shape {tuple(summed_contribs.shape)}!")
        # axis 1 first on all xM rows, then axis 0 on the surviving columns
        dev, was_numpy = self._as_device(summed_contribs)
        tmp = self._axis_call(
            "finish_subgrid", dev, self.xM_size, subgrid_size, 1, None, False,
            subgrid_off[1], size_arg=int(subgrid_size), mask=masks[1],
        )
        res = self._axis_call(
            "finish_subgrid", tmp, self.xM_size, subgrid_size, 0, out, False,
            subgrid_off[0], size_arg=int(subgrid_size), mask=masks[0],
        )
        if was_numpy and out is None:
            return res.cpu().numpy()
        return res

    # ------------------------------------------------------------------ subgrid -> facet
    def prepare_subgrid(self, subgrid, subgrid_off, out=None):
        """Pad to ``xM_size``, align with the global grid origin and transform
        along every axis (core.py:328-368)."""
        dims = len(subgrid.shape)
        if dims == 1 and not isinstance(subgrid_off, (tuple, list)):
            subgrid_off = (subgrid_off,)
        if len(subgrid_off) != dims:
            raise ValueError("Dimensionality mismatch between subgrid and offsets!")
        if dims == 1:
            return self._axis_call(
                "prepare_subgrid", subgrid, None, self.xM_size, 0, out, False,
                subgrid_off[0], size_arg=int(subgrid.shape[0]),
            )
        if dims != 2:
            raise ValueError(f"Invalid shape {tuple(subgrid.shape)}!")
        dev, was_numpy = self._as_device(subgrid)
        # axis 1 on the xA rows only, then axis 0 on all xM columns
        tmp = self._axis_call(
            "prepare_subgrid", dev, None, self.xM_size, 1, None, False,
            subgrid_off[1], size_arg=int(subgrid.shape[1]),
        )
        res = self._axis_call(
            "prepare_subgrid", tmp, None, self.xM_size, 0, out, False,
            subgrid_off[0], size_arg=int(subgrid.shape[0]),
        )
        if was_numpy and out is None:
            return res.cpu().numpy()
        return res

    def prepare_subgrid_inplace(self, padded, subgrid_off):
        """``Swiftly.prepare_subgrid_inplace[_2d]`` of the reference's native shim (core.py:837-855): ``padded`` is a
        device tensor ``[xM]``, ``[rows, xM]`` (one offset: last axis) or ``[xM, xM]`` with a list of two offsets that
        already holds the subgrid zero-padded to ``xM_size`` (``pad_mid``); it is transformed in place."""
        torch = _torch()
        if not isinstance(padded, torch.Tensor) or padded.device != self._device or not padded.is_complex():
            raise ValueError("prepare_subgrid_inplace needs a complex tensor on the core's device")
        xM = self.xM_size
        two = isinstance(subgrid_off, (list, tuple)) and len(subgrid_off) == 2
        if any(st < 0 for st in padded.stride()):
            raise ValueError("negative strides are not supported")
        if two:
            if tuple(padded.shape) != (xM, xM):
                raise ValueError(f"Invalid shape {tuple(padded.shape)}!")
            _lib.check(
                self._lib.swiftly_hip_prepare_subgrid_inplace_2d(
                    self._handle, self._code(padded), ctypes.c_void_p(padded.data_ptr()), padded.stride(0),
                    padded.stride(1), int(subgrid_off[0]), int(subgrid_off[1]), self._stream(),
                )
            )
            return padded
        off = subgrid_off[0] if isinstance(subgrid_off, (list, tuple)) else subgrid_off
        x2 = padded.unsqueeze(0) if padded.dim() == 1 else padded
        if x2.dim() != 2 or x2.shape[1] != xM:
            raise ValueError(f"Invalid shape {tuple(padded.shape)}!")
        _lib.check(
            self._lib.swiftly_hip_prepare_subgrid_inplace(
                self._handle, self._code(padded), ctypes.c_void_p(x2.data_ptr()), x2.shape[0], x2.stride(0), x2.stride(1),
                int(off), self._stream(),
            )
        )
        return padded

    def extract_from_subgrid(self, FSi, facet_off, axis, out=None):
        """Cut the window of a prepared subgrid that lands on the facet at
        ``facet_off``, weight with Fn and inverse-transform to contribution
        size (core.py:370-406)."""
        return self._axis_call(
            "extract_from_subgrid", FSi, self.xM_size, self.xM_yN_size, axis, out, False, facet_off
        )

    def add_to_facet(self, subgrid_contrib, subgrid_off, axis, out=None):
        """ADD a subgrid contribution at its place in the padded facet
        (core.py:408-449).  Exact additions."""
        return self._axis_call(
            "add_to_facet", subgrid_contrib, self.xM_yN_size, self.yN_size, axis, out, True, subgrid_off
        )

    def finish_facet(self, MiNjSi_sum, facet_off, facet_size, axis, out=None, mask=None):
        """Transform the accumulated contributions, cut the facet out and
        multiply with 1/PSWF (core.py:452-484).  ``mask`` (extension) folds the
        facet mask multiply of api_helper.py:175-176 / 195-196."""
        return self._axis_call(
            "finish_facet", MiNjSi_sum, self.yN_size, facet_size, axis, out, False,
            facet_off, size_arg=int(facet_size), mask=mask,
        )
