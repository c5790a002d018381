"""
world_size-2/3 gloo tests of the multi-GPU exchange logic on CPU: facet
sharding, all-to-all split sizes, arrival -> global facet reordering and
subgrid ownership.  The compute callables are the ORACLE here (test
infrastructure); the product wires the same exchange to the HIP kernels
(ska_sdp_exec_swiftly_amd.distributed.DistributedForward).
"""
import os
import socket

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import swiftly_oracle as orc
from ska_sdp_exec_swiftly_amd.distributed import (
    FacetSharding,
    backward_layout,
    exchange_blocks,
    exchange_contributions,
    forward_layout,
    start_exchange,
)

P = dict(W=11.0, N=512, yB_size=176, yN_size=256, xA_size=96, xM_size=128)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    core = orc.OracleCore(P["W"], P["N"], P["xM_size"], P["yN_size"])
    facet_items = orc.make_full_cover(P["N"], P["yB_size"])
    sg_items = [s for s in orc.make_full_cover(P["N"], P["xA_size"]) if s.off0 == 96]  # one wave of 6
    facets = []
    for j, f in enumerate(facet_items):
        r = numpy.random.default_rng(99 + j)
        d = r.standard_normal((P["yB_size"],) * 2) + 1j * r.standard_normal((P["yB_size"],) * 2)
        facets.append(d * f.mask0[:, None] * f.mask1[None, :])
    return core, facet_items, sg_items, facets


def _contribs(core, facet_items, facets, sg_items, which):
    out = numpy.empty((len(which), len(sg_items), core.xM_yN_size, core.xM_yN_size), dtype=complex)
    for a, j in enumerate(which):
        bf = core.prepare_facet(facets[j], facet_items[j].off0, 0)
        col = orc.extract_column(core, bf, sg_items[0].off0, facet_items[j].off1)
        for b, sg in enumerate(sg_items):
            out[a, b] = core.extract_from_facet(col, sg.off1, 1)
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        core, facet_items, sg_items, facets = _problem()
        sh = FacetSharding(len(facet_items), rank, world)
        local = torch.from_numpy(_contribs(core, facet_items, facets, sg_items, sh.local_facets))
        # blocking and pipelined (async) forms must agree
        allc = exchange_contributions(local, sh).numpy()
        h1 = start_exchange(local, sh)
        h2 = start_exchange(2 * local, sh)
        assert numpy.array_equal(h1.wait().numpy(), allc) and numpy.array_equal(h2.wait().numpy(), 2 * allc)
        mine = sh.subgrids_of(len(sg_items))
        res = [orc.sum_and_finish_subgrid(core, list(allc[:, k]), facet_items, sg_items[i]) for k, i in enumerate(mine)]
        q.put((rank, mine, res, allc.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_exchange_matches_serial(world):
    core, facet_items, sg_items, facets = _problem()
    want = orc.forward_all(core, facet_items, facets, sg_items)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        _, mine, res, shape = q.get(timeout=240)
        assert shape[0] == len(facet_items) and shape[1] == len(mine)
        for i, r in zip(mine, res):
            got[i] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got) == list(range(len(sg_items)))
    for i, w in enumerate(want):
        assert numpy.allclose(got[i], w, rtol=0, atol=1e-13 * numpy.abs(w).max())


def test_sharding_bookkeeping():
    sh = FacetSharding(9, 1, 4)
    assert sh.facets_of == [[0, 4, 8], [1, 5], [2, 6], [3, 7]]
    assert sh.local_facets == [1, 5]
    assert sh.arrival_order == [0, 4, 8, 1, 5, 2, 6, 3, 7]
    assert [sh.arrival_order[p] for p in sh.to_global] == list(range(9))
    # ranks 1..3 carry 2 facets, rank 0 carries 3: rank 0 takes no subgrids
    assert sh.subgrid_ranks == [1, 2, 3]
    assert sh.subgrids_of(10) == [0, 3, 6, 9]
    assert sh.subgrids_of(10, 3) == [2, 5, 8]
    assert sh.subgrids_of(10, 0) == []
    flat = FacetSharding(9, 1, 4, balance=False)
    assert flat.subgrids_of(10) == [1, 5, 9] and flat.subgrids_of(10, 3) == [3, 7]
    assert FacetSharding(8, 5, 8).subgrids_of(20) == [5, 13]  # even split: plain round-robin
    one = FacetSharding(3, 0, 1)
    t = torch.zeros((3, 2, 4, 4))
    assert exchange_contributions(t, one) is t


# --------------------------------------------------------------------------------------------------------------
# The copy-free layouts DistributedForward / DistributedBackward use (blocks written straight into the send
# buffer, receive buffer consumed in arrival order), forward + backward round trip, compute by the oracle.
def _roundtrip_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        core, facet_items, sg_items, facets = _problem()
        m = core.xM_yN_size
        S = len(sg_items)
        sh = FacetSharding(len(facet_items), rank, world)
        F_local = len(sh.local_facets)
        local = _contribs(core, facet_items, facets, sg_items, sh.local_facets)  # [F_local, S, m, m]
        # ---- forward: send buffer [dest][local facet][subgrid of dest][m, m]
        dests, in_counts, out_counts = forward_layout(sh, S, m * m)
        send = torch.empty(sum(in_counts), dtype=torch.complex128)
        pos = 0
        for d, cnt in zip(dests, in_counts):
            if cnt:
                send[pos : pos + cnt].view(F_local, len(d), m, m).copy_(torch.from_numpy(local[:, d]))
            pos += cnt
        recv = exchange_blocks(send, in_counts, out_counts).wait()
        mine = sh.subgrids_of(S)
        blocks = recv.view(len(facet_items), len(mine), m, m).numpy()  # facets in ARRIVAL order
        arrival_items = [facet_items[j] for j in sh.arrival_order]
        subgrids = [
            orc.sum_and_finish_subgrid(core, list(blocks[:, k]), arrival_items, sg_items[i]) for k, i in enumerate(mine)
        ]
        # ---- backward: contributions of my subgrids to ALL facets in owner-major order = the send buffer
        in_counts, out_counts = backward_layout(sh, S, m * m)
        parts = numpy.empty((len(facet_items), len(mine), m, m), dtype=complex)
        for k, i in enumerate(mine):
            split = orc.prepare_and_split_subgrid(core, subgrids[k], [sg_items[i].off0, sg_items[i].off1], arrival_items)
            for f, c in enumerate(split):
                parts[f, k] = c
        recv = exchange_blocks(torch.from_numpy(parts).reshape(-1), in_counts, out_counts).wait()
        cols = [None] * F_local
        pos = 0
        for r in range(world):
            idx = sh.subgrids_of(S, r)
            cnt = F_local * len(idx) * m * m
            if cnt:
                chunk = recv[pos : pos + cnt].view(F_local, len(idx), m, m).numpy()
                for a in range(F_local):
                    for b, i in enumerate(idx):
                        cols[a] = orc.accumulate_column(core, chunk[a, b], cols[a], sg_items[i].off1)
            pos += cnt
        out = []
        for a, j in enumerate(sh.local_facets):
            acc = orc.accumulate_facet(core, cols[a], None, facet_items[j], sg_items[0].off0)
            out.append(orc.finish_facet_2d(core, acc, facet_items[j]))
        q.put((rank, mine, subgrids, sh.local_facets, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_copy_free_layouts_forward_backward(world):
    core, facet_items, sg_items, facets = _problem()
    want_sg = orc.forward_all(core, facet_items, facets, sg_items)
    want_f = orc.backward_all(core, facet_items, sg_items, want_sg)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_roundtrip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got_sg, got_f = {}, {}
    for _ in range(world):
        _, mine, subgrids, lf, fac = q.get(timeout=240)
        got_sg.update(dict(zip(mine, subgrids)))
        got_f.update(dict(zip(lf, fac)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got_sg) == list(range(len(sg_items))) and sorted(got_f) == list(range(len(facet_items)))
    for i, w in enumerate(want_sg):
        assert numpy.allclose(got_sg[i], w, rtol=0, atol=1e-13 * numpy.abs(w).max())
    for j, w in enumerate(want_f):
        assert numpy.allclose(got_f[j], w, rtol=0, atol=1e-12 * numpy.abs(w).max())


def test_layout_counts():
    sh = FacetSharding(9, 2, 4, balance=False)  # local facets [2, 6]
    dests, inc, outc = forward_layout(sh, 10, 5)
    assert dests == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    assert inc == [2 * 3 * 5, 2 * 3 * 5, 2 * 2 * 5, 2 * 2 * 5]
    assert outc == [3 * 2 * 5, 2 * 2 * 5, 2 * 2 * 5, 2 * 2 * 5]  # facets of src x my 2 subgrids
    binc, boutc = backward_layout(sh, 10, 5)
    assert binc == outc and boutc == inc  # the backward exchange is the transpose of the forward one


# --------------------------------------------------------------------------------------------------------------
# Cooperative facets (r4): the facets that do not fill a round of ranks are produced, wave by wave, by the rank
# that owns the wave -- per-wave item counts and arrival orders through a real all_to_all (compute by the oracle).
def _coop_problem(n_facets):
    core, facet_items, _, facets = _problem()
    sgs = [s for s in orc.make_full_cover(P["N"], P["xA_size"]) if s.off0 in (0, 96)]
    waves = {}
    for s in sgs:
        waves.setdefault(s.off1, []).append(s)
    return core, facet_items[:n_facets], facets[:n_facets], waves


def _wave_blocks(core, facet_items, facets, wave, which):
    """[len(which), S, m, m] contributions of the facets `which` to the subgrids of one wave (same off1)"""
    m = core.xM_yN_size
    out = numpy.empty((len(which), len(wave), m, m), dtype=complex)
    for a, j in enumerate(which):
        bf = core.prepare_facet(facets[j], facet_items[j].off1, 1)
        col = core.prepare_facet(core.extract_from_facet(bf, wave[0].off1, 1), facet_items[j].off0, 0)
        for b, sg in enumerate(wave):
            out[a, b] = core.extract_from_facet(col, sg.off0, 0)
    return out


def _coop_worker(rank, world, port, q, n_facets):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        core, facet_items, facets, waves = _coop_problem(n_facets)
        m = core.xM_yN_size
        keys = sorted(waves)
        sh = FacetSharding(n_facets, rank, world, wave_keys=keys)
        assert sh.coop and all(len(sh.facets_of[r]) == n_facets // world for r in range(world))
        subgrids, cols = {}, {}
        for key in keys:
            wave = waves[key]
            S = len(wave)
            items = sh.items_of(rank, key)
            local = _wave_blocks(core, facet_items, facets, wave, items)
            dests, in_counts, out_counts = forward_layout(sh, S, m * m, key)
            send = torch.empty(sum(in_counts), dtype=torch.complex128)
            pos = 0
            for d, cnt in zip(dests, in_counts):
                if cnt:
                    send[pos : pos + cnt].view(len(items), len(d), m, m).copy_(torch.from_numpy(local[:, d]))
                pos += cnt
            recv = exchange_blocks(send, in_counts, out_counts).wait()
            mine = sh.subgrids_of(S)
            order = sh.arrival(key)
            assert sorted(order) == list(range(n_facets))
            blocks = recv.view(n_facets, len(mine), m, m).numpy()
            arrival_items = [facet_items[j] for j in order]
            fin = [orc.sum_and_finish_subgrid(core, list(blocks[:, k]), arrival_items, wave[i]) for k, i in enumerate(mine)]
            for k, i in enumerate(mine):
                subgrids[(key, i)] = fin[k]
            # backward: my subgrids' contributions to all facets in this wave's arrival order
            in_counts, out_counts = backward_layout(sh, S, m * m, key)
            parts = numpy.empty((n_facets, len(mine), m, m), dtype=complex)
            for k, i in enumerate(mine):
                for f, c in enumerate(orc.prepare_and_split_subgrid(core, fin[k], [wave[i].off0, wave[i].off1], arrival_items)):
                    parts[f, k] = c
            recv = exchange_blocks(torch.from_numpy(parts).reshape(-1), in_counts, out_counts).wait()
            pos = 0
            for r in range(world):
                idx = sh.subgrids_of(S, r)
                cnt = len(items) * len(idx) * m * m
                if cnt:
                    chunk = recv[pos : pos + cnt].view(len(items), len(idx), m, m).numpy()
                    for a, j in enumerate(items):
                        for b, i in enumerate(idx):
                            cols.setdefault(j, []).append((wave[i].off0, wave[i].off1, chunk[a, b]))
                pos += cnt
        q.put((rank, subgrids, cols))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n_facets", [(2, 9), (3, 7)])
def test_cooperative_layouts_forward_backward(world, n_facets):
    """per-wave item counts / arrival orders of the cooperative sharding under a real all_to_all: finished subgrids
    equal the serial oracle, and every (facet, subgrid) contribution of the backward exchange reaches exactly one rank"""
    core, facet_items, facets, waves = _coop_problem(n_facets)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_coop_worker, args=(r, world, port, q, n_facets)) for r in range(world)]
    for p in procs:
        p.start()
    got_sg, got_cols = {}, {}
    for _ in range(world):
        _, subgrids, cols = q.get(timeout=500)
        got_sg.update(subgrids)
        for j, lst in cols.items():
            got_cols.setdefault(j, []).extend(lst)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_sub = sum(len(w) for w in waves.values())
    assert len(got_sg) == n_sub
    for key, wave in waves.items():
        want = orc.forward_all(core, facet_items, facets, wave)
        for i, w in enumerate(want):
            assert numpy.allclose(got_sg[(key, i)], w, rtol=0, atol=1e-13 * numpy.abs(w).max())
    # backward: every facet received one contribution per subgrid, equal to the serial split of that subgrid
    for j in range(n_facets):
        assert len(got_cols[j]) == n_sub
        for off0, off1, c in got_cols[j][:3]:
            sg = got_sg[next((k, i) for k, w in waves.items() for i, s in enumerate(w) if (s.off0, s.off1) == (off0, off1))]
            want = orc.prepare_and_split_subgrid(core, sg, [off0, off1], [facet_items[j]])[0]
            assert numpy.allclose(c, want, rtol=0, atol=1e-13 * max(numpy.abs(want).max(), 1e-30))


def test_cooperative_sharding_bookkeeping():
    keys = [10, 20, 30, 40, 50]
    sh = FacetSharding(9, 3, 8, wave_keys=keys)
    assert sh.coop == [8] and sh.facets_of == [[r] for r in range(8)] and sh.subgrid_ranks == list(range(8))
    assert [len(k) for k in sh.keys_of] == [0, 1, 0, 1, 1, 0, 1, 1] and sorted(sh.key_owner) == keys
    assert sh.items_of(3, 20) == [3, 8] and sh.items_of(3, 30) == [3] and sh.items_of(3) == [3]
    assert sh.arrival(20) == [0, 1, 2, 3, 8, 4, 5, 6, 7]
    cuts = [sh.coop_rows(22528, r) for r in range(8)]
    assert cuts[0] == (0, 2816) and sum(n for _, n in cuts) == 22528 and all(r0 % 8 == 0 for r0, _ in cuts)
    assert [c[0] for c in cuts[1:]] == [sum(n for _, n in cuts[: i + 1]) for i in range(7)]
    small = FacetSharding(3, 0, 8, wave_keys=keys)  # fewer facets than ranks: all cooperative
    assert small.coop == [0, 1, 2] and small.local_facets == [] and small.arrival(40) == [0, 1, 2]
    assert FacetSharding(9, 0, 8).coop == [] and FacetSharding(8, 0, 8, wave_keys=keys).coop == []
    dests, inc, outc = forward_layout(sh, 10, 5, 20)
    assert inc == [2 * len(d) * 5 for d in dests]  # rank 3 owns wave 20: two items
    assert outc[3] == 2 * len(sh.subgrids_of(10)) * 5 and outc[2] == len(sh.subgrids_of(10)) * 5
    binc, boutc = backward_layout(sh, 10, 5, 20)
    assert binc == outc and boutc == inc
