"""Per-stage complex64 accuracy probe (GPU): for every primitive of the forward chain, on the SAME complex64 input,
relative RMSE vs the complex128 oracle of (a) the HIP kernel and (b) numpy's own float32 path (pocketfft in
complex64 with float32 windows) -- tells which stage, if any, is less accurate than a generic float32 FFT."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
from oracle import swiftly_oracle as orc  # noqa: E402  (checker)
from ska_sdp_exec_swiftly_amd import SwiftlyCoreHip  # noqa: E402


def rel(a, b):
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / np.mean(np.abs(b) ** 2)))


def f32core(P):
    c = orc.OracleCore(P["W"], P["N"], P["xM"], P["yN"])
    c.pswf = c.pswf.astype(np.float32)
    c.Fn = c.Fn.astype(np.float32)
    fw = c.facet_window
    c.facet_window = lambda n: fw(n).astype(np.float32)
    return c


def probe(P, nrows=64):
    print(f"== {P}")
    hip = SwiftlyCoreHip(P["W"], P["N"], P["xM"], P["yN"])
    ref = orc.OracleCore(P["W"], P["N"], P["xM"], P["yN"])
    r32 = f32core(P)
    rng = np.random.default_rng(5)
    yB, yN, xA, xM, m = P["yB"], P["yN"], P["xA"], P["xM"], ref.xM_yN_size
    fo0, fo1, so0, so1 = P["fo"], P["fo"] * 2, P["so"], -P["so"] * 3
    c64 = lambda a: a.astype(np.complex64)  # noqa: E731

    def stage(name, fn, x, *args):
        want = fn(ref, x.astype(complex), *args)
        got = fn(hip, x, *args)
        emu = c64(fn(r32, x, *args))
        print(f"  {name:<34} hip {rel(got, want):.3e}   numpy-f32 {rel(emu, want):.3e}   ratio {rel(got, want) / rel(emu, want):.2f}")
        return c64(want)

    rows = c64(rng.standard_normal((nrows, yB)) + 1j * rng.standard_normal((nrows, yB)))
    stage("prepare_facet axis1 (contiguous)", lambda c, x: c.prepare_facet(x, fo1, axis=1), rows)
    cols = np.ascontiguousarray(rows.T)
    bf = stage("prepare_facet axis0 (strided)", lambda c, x: c.prepare_facet(x, fo0, axis=0), cols)
    ext = c64(ref.extract_from_facet(bf.astype(complex), so0, axis=0))  # [m, nrows]
    sq = c64(rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)))
    a0 = stage("add_to_subgrid axis0", lambda c, x: c.add_to_subgrid(x, fo0, axis=0), sq)
    a01 = stage("add_to_subgrid axis1", lambda c, x: c.add_to_subgrid(x, fo1, axis=1), a0)
    big = c64(rng.standard_normal((xM, xM)) + 1j * rng.standard_normal((xM, xM)))
    stage("finish_subgrid 2d (random input)", lambda c, x: c.finish_subgrid(x, [so0, so1], xA), big)
    del ext, a01

    if yB > 2048:
        return
    # the chain on one dense facet: cumulative error after each stage
    facet = c64(rng.standard_normal((yB, yB)) + 1j * rng.standard_normal((yB, yB)))

    def chain(c, x, upto):
        t = c.prepare_facet(x, fo0, axis=0)
        if upto == 0:
            return t
        t = c.prepare_facet(c.extract_from_facet(t, so0, axis=0), fo1, axis=1)
        if upto == 1:
            return t
        t = c.extract_from_facet(t, so1, axis=1)
        t = c.add_to_subgrid(c.add_to_subgrid(t, fo0, axis=0), fo1, axis=1)
        if upto == 2:
            return t
        return c.finish_subgrid(t, [so0, so1], xA)

    for upto, name in enumerate(["BF_F", "NMBF_BF", "padded subgrid", "finished subgrid"]):
        want = chain(ref, facet.astype(complex), upto)
        got = chain(hip, facet, upto)
        t = facet
        emu = chain(r32, t, upto)
        print(f"  chain -> {name:<25} hip {rel(got, want):.3e}   numpy-f32 {rel(c64(emu), want):.3e}")


if __name__ == "__main__":
    probe(dict(W=11.0, N=8192, yB=1408, yN=2048, xA=1024, xM=2048, fo=4 * 16, so=4 * 4))
    probe(dict(W=10.875, N=65536, yB=22528, yN=32768, xA=928, xM=1024, fo=64 * 5, so=2 * 464), nrows=16)
