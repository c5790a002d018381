"""Per-rank compute time and exchange volume of a forward (and backward) pass at world sizes 1..8, measured with
VIRTUAL ranks on ONE GPU: every rank's objects are built in one process (`rank_world=`), each rank's share of the
pass is timed on its own, and the all-to-all is NOT executed (a rank's receive buffer is a dummy of the right size).
What this gives: the compute-side critical path per rank (max over ranks) and the bytes each rank has to send per
pass -- the two inputs of a scaling estimate.  What it does not give: a scaling curve (RCCL over xGMI never runs here).

    python tools/virtual_rank_time.py [workload] [out.json]
"""
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
from oracle import separable as sep  # noqa: E402  (data recipe only)
from ska_sdp_exec_swiftly_amd.distributed import DistributedBackward, DistributedForward  # noqa: E402


WHOLE = os.environ.get("VR_WHOLE_WAVES") == "1"  # every wave's subgrids on one rank (DistributedForward(whole_waves=True))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "64k-sparse"
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    wl = bench.WORKLOADS[name]
    p = wl["params"]
    cfg = sw.SwiftlyConfig(backend="hip", **p)
    all_fcs = sw.make_full_facet_cover(cfg)
    sgs = bench.select_subgrids(sw.make_full_subgrid_cover(cfg), p["N"], p["xA_size"], wl["sparse_radius"])
    m = cfg.core.xM_yN_size
    xA = p["xA_size"]
    res = dict(workload=wl["name"], subgrid_ownership="whole waves" if WHOLE else "round-robin within a wave", worlds={})
    one = bench.separable_facet(torch, sep.facet_vectors(1234, p["yB_size"]), all_fcs[0])

    def single_path_ms():
        """the pass bench.py --gpus 1 times (SwiftlyForward, planned waves): what a scaling series divides by"""
        fcs = all_fcs if wl.get("max_facets_per_rank") is None else all_fcs[: wl["max_facets_per_rank"]]
        axis = sw.api.preferred_wave_axis(cfg, torch.complex64, n_facets=len(fcs))
        key = (lambda c: c.off1) if axis == 1 else (lambda c: c.off0)
        wv = {}
        for c in sgs:
            wv.setdefault(key(c), []).append(c)

        def one_pass():
            fwd = sw.SwiftlyForward(cfg, [(f, one) for f in fcs], lru_forward=1, subgrid_configs=sgs, wave_axis=axis)
            fwd.prepare_all_facets()
            for w in wv.values():
                fwd.get_wave(w)

        one_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            one_pass()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 * 1e3

    res["single_path_forward_ms"] = round(single_path_ms(), 2)
    print(f"{name}: single path (SwiftlyForward) {res['single_path_forward_ms']} ms", flush=True)
    for world in (1, 2, 4, 8):
        cap = wl.get("max_facets_per_rank")
        n_active = len(all_fcs) if cap is None else min(len(all_fcs), cap * world)
        fcs = all_fcs[:n_active]
        axis = sw.api.preferred_wave_axis(cfg, torch.complex64, n_facets=len(fcs))
        key = (lambda c: c.off1) if axis == 1 else (lambda c: c.off0)
        waves = {}
        for c in sgs:
            waves.setdefault(key(c), []).append(c)
        waves = list(waves.values())
        max_waves = int(os.environ.get("VR_MAX_WAVES", "0"))  # big covers: time a prefix of the waves and scale
        scale = 1.0
        if max_waves and len(waves) > max_waves:
            scale = len(waves) / max_waves
            waves = waves[:max_waves]
        data = [one] * len(fcs)  # timing only: every facet holds the same numbers
        # every rank when facets are worked on cooperatively (the wave ranges differ by one wave), else the extremes
        ranks = list(range(world)) if len(fcs) % world else sorted({0, 1, world - 1} & set(range(world)))
        ent = dict(facets=len(fcs), wave_axis=axis, waves_timed=len(waves), wave_scale=scale, ranks={})
        for rank in ranks:
            sent = [0]

            def forward_pass():
                dfw = DistributedForward(cfg, fcs, data, subgrid_configs=sgs, wave_axis=axis, dtype=torch.complex64,
                                         rank_world=(rank, world), whole_waves=WHOLE)
                dfw.prepare_all_facets()
                sent[0] = 0
                for j in dfw.sharding.coop:  # cooperative facets: K1 on this rank's rows + (dummy) band-row exchange
                    send, inc, outc = dfw.pack_coop(j)
                    sent[0] += 8 * (sum(inc) - inc[rank])
                    dfw.unpack_coop(j, torch.empty(sum(outc), dtype=torch.complex64, device="cuda"))
                if WHOLE and dfw.sharding.wave_groups:
                    # bench.py's grouped loop: the waves group by group (distinct owners), one (dummy) exchange per group,
                    # the finish of group g behind the packing of group g + 1
                    by_key = {dfw.wave_key(w): w for w in waves}
                    pend = None
                    for group in dfw.sharding.wave_groups:
                        gw = [by_key[k] for k in group if k in by_key]
                        if not gw:
                            continue
                        send, inc, outc = dfw.pack_group(gw)
                        sent[0] += 8 * (sum(inc) - inc[rank])
                        recv = send if world == 1 else torch.empty(sum(outc), dtype=torch.complex64, device="cuda")
                        if pend is not None:
                            dfw.unpack_group(*pend)
                        pend = (gw, recv)
                    if pend is not None:
                        dfw.unpack_group(*pend)
                    return
                for wave in waves:
                    send, inc, outc = dfw.pack_wave(wave)
                    sent[0] += 8 * (sum(inc) - inc[rank])
                    recv = send if world == 1 else torch.empty(sum(outc), dtype=torch.complex64, device="cuda")
                    dfw.unpack_wave(wave, recv)

            def backward_pass(subs):
                dbw = DistributedBackward(cfg, fcs, wave_axis=1 if cfg.core.supports_backward_band(torch.complex64) else 0,
                                          subgrid_configs=sgs, dtype=torch.complex64, rank_world=(rank, world),
                                          whole_waves=WHOLE)
                for wave, mine, got in subs:
                    send, inc, outc = dbw.pack_wave(wave, [got[k] for k in range(len(mine))] if got is not None else [])
                    recv = send if world == 1 else torch.empty(sum(outc), dtype=torch.complex64, device="cuda")
                    if world > 1:
                        recv.zero_()
                    dbw.unpack_wave(wave, recv)
                out = dbw.finish()
                for j in dbw.sharding.coop:  # finishing exchange of the cooperative facets (dummy receive buffer)
                    send, inc, outc = dbw.pack_coop_finish(j)
                    dbw.unpack_coop_finish(j, torch.zeros(sum(outc), dtype=torch.complex64, device="cuda"))
                return out

            def timed(fn, reps=2):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3

            t_f = timed(forward_pass) * scale
            # subgrid data for the backward leg: random subgrids of the right shape for the ones this rank holds
            dfw0 = DistributedForward(cfg, fcs, data, subgrid_configs=sgs, wave_axis=axis, dtype=torch.complex64,
                                      rank_world=(rank, world), whole_waves=WHOLE)
            subs = []
            bkey = (lambda c: c.off1) if cfg.core.supports_backward_band(torch.complex64) else (lambda c: c.off0)
            bw = {}
            for c in sgs:
                bw.setdefault(bkey(c), []).append(c)
            bwl = list(bw.values())[: len(waves)]
            for wave in bwl:
                mine = dfw0.subgrids_of(wave)
                got = torch.randn((len(mine), xA, xA), dtype=torch.complex64, device="cuda") if mine else None
                subs.append((wave, mine, got))
            del dfw0
            t_b = timed(lambda: backward_pass(subs)) * scale
            ent["ranks"][str(rank)] = dict(forward_ms=round(t_f, 2), backward_ms=round(t_b, 2),
                                           forward_sent_MB_per_pass=round(sent[0] * scale / 1e6, 1))
            print(f"{name} world {world} rank {rank}: forward {t_f:.2f} ms, backward {t_b:.2f} ms, sends {sent[0] * scale / 1e6:.0f} MB per forward pass",
                  flush=True)
            del subs
            torch.cuda.empty_cache()
        ent["critical_path_forward_ms"] = max(r["forward_ms"] for r in ent["ranks"].values())
        ent["critical_path_backward_ms"] = max(r["backward_ms"] for r in ent["ranks"].values())
        ent["speedup_vs_single_path_forward"] = round(res["single_path_forward_ms"] / ent["critical_path_forward_ms"], 2)
        res["worlds"][str(world)] = ent
    if out_path:
        with open(out_path, "w", encoding="utf-8") as fh:
            json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
