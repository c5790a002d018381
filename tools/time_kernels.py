"""Time the individual stages of the 64k workload on the GPU (HIP events)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch
import ska_sdp_exec_swiftly_amd as sw

P = dict(W=10.875, fov=1.0, N=65536, yB_size=22528, yN_size=32768, xA_size=928, xM_size=1024)


def timeit(fn, it=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    cfg = sw.SwiftlyConfig(backend="hip", **P)
    core = cfg.core
    yB, yN, m, xM, xA = P["yB_size"], P["yN_size"], core.xM_yN_size, P["xM_size"], P["xA_size"]
    facet = torch.randn((yB, yB), device="cuda", dtype=torch.complex64)
    out = {}
    bf = core.prepare_facet(facet, 22528, axis=0)
    ms = timeit(lambda: core.prepare_facet(facet, 22528, axis=0, out=bf))
    gb = 8 * (yB * yB + yN * yB) / 1e9
    out["K1_prepare_facet_axis0"] = dict(ms=ms, algo_GBs=gb / ms * 1e3, actual_GBs=8 * (yB * yB + 3 * yN * yB) / 1e9 / ms * 1e3)
    col = core.extract_column(bf, 928 * 3, 0)
    ms = timeit(lambda: core.extract_column(bf, 928 * 3, 0, out=col), it=20)
    gb = 8 * (m * yB + m * yN) / 1e9
    out["K2_extract_column"] = dict(ms=ms, algo_GBs=gb / ms * 1e3)
    S = 20
    contrib = torch.empty((S, m, m), device="cuda", dtype=torch.complex64)
    offs = [928 * i for i in range(S)]
    ms = timeit(lambda: core.launch("extract_from_facet", col, m, yN, 1, contrib, m, 1, nbatch=S, in_bs=0, out_bs=m * m, offs=offs), it=20)
    out["K3_extract_from_facet_x20"] = dict(ms=ms, algo_GBs=8 * 2 * S * m * m / 1e9 / ms * 1e3)
    from ska_sdp_exec_swiftly_amd.api import sum_and_finish_wave, SubgridConfig, make_full_facet_cover
    fcs = make_full_facet_cover(cfg)
    sgs = [SubgridConfig(928 * 3, 928 * i, xA) for i in range(S)]
    c9 = torch.randn((9, S, m, m), device="cuda", dtype=torch.complex64)
    ms = timeit(lambda: sum_and_finish_wave(core, c9, fcs, sgs), it=10)
    out["K45_sum_and_finish_wave20"] = dict(ms=ms, algo_GBs=8 * (9 * S * m * m + S * xA * xA) / 1e9 / ms * 1e3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
