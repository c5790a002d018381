"""Experiment: can the latency/issue-bound full-facet row transform (K1) and the bandwidth-bound per-wave kernels
(K2..K5) share the chip on DISJOINT compute units?  (on the same CUs they cannot co-reside --
K1's two workgroups take 132 of 160 KB LDS and all VGPRs.)

  1. census: which (XCD, SE, CU) does a CU-masked stream reach, for a few mask shapes;
  2. K1 x 9 alone / waves x 25 alone on the full chip and on masked streams;
  3. both at once on complementary masks (independent data: K1 writes a second band buffer).

Usage: python tools/cu_mask_overlap.py [out.json]
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))

import torch  # noqa: E402

import bench  # noqa: E402
import ska_sdp_exec_swiftly_amd as sw  # noqa: E402
from ska_sdp_exec_swiftly_amd import _lib  # noqa: E402

lib = _lib.load()
NCU = 256
NW = NCU // 32


def make_stream(bits):
    words = (ctypes.c_uint32 * NW)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    _lib.check(lib.swiftly_hip_stream_create_cu_mask(ctypes.byref(st), words, NW))
    return torch.cuda.ExternalStream(st.value), st


def census(stream, nblocks=4096):
    out = torch.full((nblocks,), -1, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(stream):
        _lib.check(lib.swiftly_hip_cu_census(ctypes.c_void_p(out.data_ptr()), nblocks, ctypes.c_void_p(stream.cuda_stream)))
    stream.synchronize()
    vals = out.cpu().numpy()
    per_xcc = {}
    for v in set(vals.tolist()):
        per_xcc.setdefault(v >> 16, set()).add(v & 0xFFFF)
    return {int(k): len(v) for k, v in sorted(per_xcc.items())}


def main():
    res = {}
    torch.cuda.set_device(0)
    # -- 1. census
    shapes = {
        "all": list(range(NCU)),
        "first64": list(range(64)),
        "last64": list(range(192, 256)),
        "every4th": list(range(0, NCU, 4)),
        "low8_of_each32": [b for b in range(NCU) if b % 32 < 8],
    }
    res["census"] = {}
    for name, bits in shapes.items():
        st, raw = make_stream(bits)
        res["census"][name] = census(st)
        print("census", name, res["census"][name], flush=True)
    res["census"]["default_stream"] = census(torch.cuda.current_stream())
    print("census default", res["census"]["default_stream"], flush=True)

    # -- 2./3. the 64k-sparse workload
    wl = bench.WORKLOADS["64k-sparse"]
    p = wl["params"]
    cfg = sw.SwiftlyConfig(backend="hip", **p)
    facet_cfgs = sw.make_full_facet_cover(cfg)
    sg_cfgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
    yB = p["yB_size"]
    g = torch.Generator(device="cuda").manual_seed(1)
    one = torch.randn((yB, yB), dtype=torch.complex64, device="cuda", generator=g)
    facets = [one for _ in facet_cfgs]  # same data for every facet: timing only
    waves = {}
    for c in sg_cfgs:
        waves.setdefault(c.off1, []).append(c)
    waves = list(waves.values())

    def make():
        return sw.SwiftlyForward(cfg, list(zip(facet_cfgs, facets)), subgrid_configs=sg_cfgs, wave_axis=1)

    fwd_a, fwd_b = make(), make()
    fwd_a.prepare_all_facets()
    fwd_b.prepare_all_facets()
    for w in waves[:2]:
        fwd_a.get_wave(w)
    torch.cuda.synchronize()

    def run_k1(f):
        f.BF_Fs_persist = None
        f.prepare_all_facets()

    def run_waves(f):
        for w in waves:
            f.get_wave(w)

    def timed(fn_by_stream, reps=3):
        """fn_by_stream: [(stream or None, callable)]: all enqueued back to back, wall time until all are done"""
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st, fn in fn_by_stream:
                if st is None:
                    fn()
                else:
                    with torch.cuda.stream(st):
                        fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return round(best, 3)

    res["full_chip"] = dict(k1=timed([(None, lambda: run_k1(fwd_a))]), waves=timed([(None, lambda: run_waves(fwd_b))]))
    res["full_chip"]["sequential"] = timed([(None, lambda: (run_k1(fwd_a), run_waves(fwd_b)))])
    print("full chip", res["full_chip"], flush=True)
    # two unmasked streams: does plain concurrency help?
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res["two_plain_streams"] = timed([(s1, lambda: run_k1(fwd_a)), (s2, lambda: run_waves(fwd_b))])
    print("two plain streams", res["two_plain_streams"], flush=True)
    res["masked"] = {}
    for n_b in (32, 64, 96, 128):
        for layout in ("interleaved", "block"):
            if layout == "interleaved":
                per = NCU // n_b
                bits_b = [b for b in range(NCU) if b % per == 0] if NCU % n_b == 0 else [
                    b for b in range(NCU) if (b * n_b) // NCU != ((b - 1) * n_b) // NCU or b == 0
                ][:n_b]
            else:
                bits_b = list(range(NCU - n_b, NCU))
            bits_a = [b for b in range(NCU) if b not in set(bits_b)]
            sa, _ = make_stream(bits_a)
            sb, _ = make_stream(bits_b)
            ent = dict(
                census_a=census(sa), census_b=census(sb),
                k1_on_a=timed([(sa, lambda: run_k1(fwd_a))]),
                waves_on_b=timed([(sb, lambda: run_waves(fwd_b))]),
                both=timed([(sa, lambda: run_k1(fwd_a)), (sb, lambda: run_waves(fwd_b))]),
            )
            res["masked"][f"{NCU - n_b}+{n_b} {layout}"] = ent
            print(f"{NCU - n_b}+{n_b} {layout}", ent, flush=True)
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "cu_mask_overlap.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w", encoding="utf-8") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
