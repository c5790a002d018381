"""Per-call digests of K1 (prepare_facet_band) outputs for A/B runs of kernel variants: band columns only."""
import hashlib
import os
import sys

import numpy
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip  # noqa: E402

core = SwiftlyCoreHip(10.875, 65536, 1024, 32768)
rng = numpy.random.default_rng(5)


def band_cols(yN, band):
    start, length = band
    half = ((length + 1) // 2 + 15) // 16 * 16
    d = (numpy.arange(yN) - start) % yN
    return numpy.where(d < length, (d & 1) * half + (d >> 1), -1)


def dig(a):
    return hashlib.sha256(numpy.ascontiguousarray(a).tobytes()).hexdigest()[:12]


for rows, cols in ((5, 22528), (700, 22528), (300, 16384)):
    x = torch.from_numpy((rng.standard_normal((rows, cols)) + 1j * rng.standard_normal((rows, cols))).astype(numpy.complex64)).cuda()
    for off, bnd in ((0, (10736, 11472)), (64 * 352, (10736, 11472)), (-64 * 320, (32001, 2049)), (64 * 352, (0, 32768))):
        if cols == 16384 and off:
            continue
        out = core.prepare_facet_band(x, off, bnd, fold_other_axis_window=(rows != 700)).cpu().numpy()
        pc = band_cols(32768, bnd)
        keep = pc[pc >= 0]
        print("DIG", rows, cols, off, bnd, dig(out[:, keep]), "rows:", " ".join(dig(out[r, keep])[:6] for r in (0, 1, rows // 2, rows - 1)),
              "nan", int(numpy.isnan(out[:, keep]).sum()))
