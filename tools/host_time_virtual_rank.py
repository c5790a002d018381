"""Host (enqueue) time against total time of one virtual rank's forward pass: is a rank of an 8-GPU run bound by the
Python side?   python tools/host_time_virtual_rank.py [workload] [world] [rank]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import ska_sdp_exec_swiftly_amd as sw  # noqa: E402
from oracle import separable as sep  # noqa: E402  (data recipe only)
from ska_sdp_exec_swiftly_amd.distributed import DistributedForward  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "64k-sparse"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 1
wl = bench.WORKLOADS[name]
p = wl["params"]
cfg = sw.SwiftlyConfig(backend="hip", **p)
fcs = sw.make_full_facet_cover(cfg)
sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
one = bench.separable_facet(torch, sep.facet_vectors(1234, p["yB_size"]), fcs[0])
data = [one] * len(fcs)
axis = sw.api.preferred_wave_axis(cfg, torch.complex64, n_facets=len(fcs))
waves = {}
for c in sgs:
    waves.setdefault(c.off1 if axis == 1 else c.off0, []).append(c)
waves = list(waves.values())


WHOLE = os.environ.get("VR_WHOLE_WAVES") == "1"


def one_pass():
    dfw = DistributedForward(cfg, fcs, data, subgrid_configs=sgs, wave_axis=axis, dtype=torch.complex64, rank_world=(rank, world),
                             whole_waves=WHOLE)
    t0 = time.perf_counter()
    dfw.prepare_all_facets()
    for j in dfw.sharding.coop:  # cooperative facets: K1 on this rank's rows + (dummy) band-row exchange
        send, inc, outc = dfw.pack_coop(j)
        dfw.unpack_coop(j, torch.empty(sum(outc), dtype=torch.complex64, device="cuda"))
    t1 = time.perf_counter()
    tp = tu = 0.0
    for wave in waves:
        a = time.perf_counter()
        send, inc, outc = dfw.pack_wave(wave)
        b = time.perf_counter()
        recv = torch.empty(sum(outc), dtype=torch.complex64, device="cuda")
        dfw.unpack_wave(wave, recv)
        c = time.perf_counter()
        tp += b - a
        tu += c - b
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, tp * 1e3, tu * 1e3, (t2 - t0) * 1e3, (t3 - t0) * 1e3


one_pass()
torch.cuda.synchronize()
for _ in range(3):
    k1, tp, tu, host, total = one_pass()
    print(f"{name} world {world} rank {rank}: host enqueue {host:.2f} ms (K1 {k1:.2f}, pack {tp:.2f}, unpack {tu:.2f}) of {total:.2f} ms total, {len(waves)} waves")

if os.environ.get("VR_PROFILE") == "1":
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    one_pass()
    pr.disable()
    st = pstats.Stats(pr).strip_dirs()
    rows = sorted(((v[2], v[3], v[0], k) for k, v in st.stats.items()), reverse=True)  # tottime, cumtime, calls, (file, line, name)
    print("tottime_us cumtime_us calls function")
    for tt, ct, nc, k in rows[:45]:
        print(f"{tt * 1e6:9.0f} {ct * 1e6:9.0f} {nc:6d} {k[0]}:{k[1]}({k[2]})")
