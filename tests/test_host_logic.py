"""CPU tests of the host-side logic of the product package (no GPU, no
kernels): covers and masks, config classes, LRU cache, truth generators,
catalogue -- each against the reference-generated goldens or the oracle."""
import os

import numpy
import pytest

from oracle import swiftly_oracle as orc
from ska_sdp_exec_swiftly_amd import api, api_helper
from ska_sdp_exec_swiftly_amd.core_hip import calculate_pswf
from ska_sdp_exec_swiftly_amd.swift_configs import SWIFT_CONFIGS

SMALL = dict(N=512, yB_size=208, xA_size=100)


class Cfg:
    image_size = SMALL["N"]
    max_facet_size = SMALL["yB_size"]
    max_subgrid_size = SMALL["xA_size"]


def test_full_cover_matches_reference(golden_dir):
    g = numpy.load(os.path.join(golden_dir, "roundtrip2d.npz"))
    for maker, offs, m0, m1 in (
        (api.make_full_facet_cover, g["facet_offs"], g["facet_mask0"], g["facet_mask1"]),
        (api.make_full_subgrid_cover, g["sg_offs"], g["sg_mask0"], g["sg_mask1"]),
    ):
        cover = maker(Cfg)
        assert numpy.array_equal(numpy.array([[c.off0, c.off1] for c in cover]), offs)
        assert numpy.array_equal(numpy.array([c.mask0 for c in cover]), m0)
        assert numpy.array_equal(numpy.array([c.mask1 for c in cover]), m1)
    assert isinstance(api.make_full_facet_cover(Cfg)[0], api.FacetConfig)
    assert isinstance(api.make_full_subgrid_cover(Cfg)[0], api.SubgridConfig)


def test_cover_partitions_every_pixel():
    for N, chunk in ((512, 208), (512, 100), (1024, 416), (1024, 228), (96, 32)):
        cover = api.make_full_cover_config(N, chunk, api.FacetConfig)
        hits = numpy.zeros(N)
        for c in cover:
            if c.off1 != 0:
                continue
            idx = (numpy.arange(chunk) - chunk // 2 + c.off0) % N
            numpy.add.at(hits, idx, c.mask0)
        assert numpy.array_equal(hits, numpy.ones(N))


def test_config_mask_forms():
    arr = numpy.array([0.0, 1.0, 1.0, 0.0])
    c = api.SubgridConfig(3, 5, 4, arr, [[slice(1, 3)], 4])
    assert c.mask0 is arr
    assert numpy.array_equal(c.mask1, arr)
    assert api.FacetConfig(0, 0, 4).mask0 is None
    assert numpy.array_equal(api.make_mask_from_slice([slice(0, 1), slice(2, 4)], 5), [1, 0, 1, 1, 0])


def test_lru_cache_semantics():
    """interface of reference api.py:525-590"""
    lru = api.LRUCache(2)
    assert lru.get("a") is None
    assert lru.set("a", 1) == (None, None)
    assert lru.set("b", 2) == (None, None)
    assert lru.get("a") == 1  # refreshes a
    assert lru.set("c", 3) == ("b", 2)  # b is now the oldest
    assert lru.set("a", 10) == (None, None)  # update in place
    assert list(lru.pop_all()) == [("c", 3), ("a", 10)]
    assert list(lru.pop_all()) == []


def test_unknown_backend():
    with pytest.raises(ValueError, match="Unknown SwiFTly backend"):
        api.SwiftlyConfig(13.5625, 1.0, 1024, 416, 512, 228, 256, backend="numpy-ish")


def test_truth_generators_match_oracle():
    rng = numpy.random.default_rng(5)
    N = 256
    for dims in (1, 2):
        srcs = [(float(rng.standard_normal()), *rng.integers(-N // 2, N // 2, dims).tolist()) for _ in range(6)]
        for size, off in ((40, 0), (41, 17), (40, -300), (33, 250)):
            offs = [off, -off + 4][:dims]
            masks = [rng.integers(0, 2, size).astype(float) for _ in range(dims)]
            a = api_helper.make_facet_from_sources(srcs, N, size, offs, masks)
            b = orc.make_facet_from_sources(srcs, N, size, offs, masks)
            assert numpy.array_equal(a, b)
            a = api_helper.make_subgrid_from_sources(srcs, N, size, offs, masks)
            b = orc.make_subgrid_from_sources(srcs, N, size, offs, masks)
            assert numpy.allclose(a, b, rtol=0, atol=1e-15)


def test_check_helpers():
    cfg = api.FacetConfig(8, -4, 16, None, None)
    srcs = [(1.0, 9, -3), (0.5, 2, 2)]
    f = api_helper.make_facet(64, cfg, srcs)
    assert f.sum() == 1.5
    assert api_helper.check_facet(64, cfg, f, srcs) == 0
    assert api_helper.check_residual(numpy.ones((4, 4))) == 1
    sg_cfg = api.SubgridConfig(4, 6, 10)
    sg = api_helper.make_subgrid(64, sg_cfg, srcs)
    assert api_helper.check_subgrid(64, sg_cfg, sg, srcs) == 0


def test_pswf_matches_reference(golden_dir):
    g = numpy.load(os.path.join(golden_dir, "constants.npz"))
    assert numpy.array_equal(calculate_pswf(13.5625, 512), g["test_pswf"])
    assert numpy.array_equal(calculate_pswf(11.0, 2048), g["bench8k_pswf"])


def test_catalogue():
    assert len(SWIFT_CONFIGS) == 244
    first = next(iter(SWIFT_CONFIGS))
    assert first == "128k[1]-n32k-512"
    c = SWIFT_CONFIGS["64k[1]-n32k-1k"]
    assert (c["W"], c["N"], c["yB_size"], c["yN_size"], c["xA_size"], c["xM_size"]) == (10.875, 65536, 22528, 32768, 928, 1024)
    for c in SWIFT_CONFIGS.values():
        # the three divisibility rules of core.py:55-74 hold for every entry
        assert c["N"] % c["yN_size"] == 0 and c["N"] % c["xM_size"] == 0
        assert (c["xM_size"] * c["yN_size"]) % c["N"] == 0


def test_row_source_tables_are_add_to_facet_as_a_gather():
    """The gather-sum load of the backward band schedule (core_hip.build_row_sources): summing the contribution rows
    named by the tables must equal the oracle's add_to_facet scatter (core.py:441-478), for wrapped offsets, more than
    two overlapping subgrids (table groups) and contributions spread over several chunks."""
    from ska_sdp_exec_swiftly_amd.core_hip import build_row_sources

    W, N, xM, yN = 11.0, 1024, 256, 512
    ref = orc.OracleCore(W, N, xM, yN)
    m = ref.xM_yN_size
    step = ref.subgrid_off_step
    rng = numpy.random.default_rng(4)
    offs = [0, 192, 192 + step, 960, -64, 192, 192, 1024 + 384]  # duplicates: up to four sources per padded row
    chunks = [rng.standard_normal((3, m, 5)) + 1j * rng.standard_normal((3, m, 5)) for _ in range(3)]  # [block, m, cols]
    locs = [(b % 3, b // 3) for b in range(len(offs))]
    want = numpy.zeros((yN, 5), dtype=complex)
    for off, (c, blk) in zip(offs, locs):
        ref.add_to_facet(chunks[c][blk], off, axis=0, out=want)
    got = numpy.zeros((yN, 5), dtype=complex)
    flat = [ch.reshape(-1, 5) for ch in chunks]  # row index = block * m + k
    seen = []
    for members, tab in build_row_sources(N, yN, m, offs, locs):
        seen += members
        assert tab.shape == (2, yN) and tab.dtype == numpy.int32
        for lvl in range(2):
            rows = numpy.flatnonzero(tab[lvl] >= 0)
            enc = tab[lvl, rows]
            for r, e in zip(rows, enc):
                got[r] += flat[e >> 20][e & 0xFFFFF]
        assert (tab[1] >= 0).sum() <= (tab[0] >= 0).sum()
    assert sorted(seen) == list(range(len(offs)))
    assert len(build_row_sources(N, yN, m, offs, locs)) >= 2  # four sources per row cannot fit one table pair
    numpy.testing.assert_array_equal(got, want)


def test_band_range_is_the_smallest_cyclic_cover():
    """core_hip.band_range: contains the window of every offset (wrap-around included) and cannot be shortened at
    either end; the bench workload's band (25 columns of the 64k configuration)."""
    from ska_sdp_exec_swiftly_amd.core_hip import band_range

    N, yN, m = 65536, 32768, 512
    rng = numpy.random.default_rng(1)
    for trial in range(20):
        offs = [int(o) * 928 for o in rng.integers(-70, 71, size=rng.integers(1, 12))]
        start, length = band_range(N, yN, m, offs, align=1)
        inside = numpy.zeros(yN, dtype=bool)
        inside[(start + numpy.arange(length)) % yN] = True
        used = numpy.zeros(yN, dtype=bool)
        for off in offs:
            s = off * yN // N
            used[(yN // 2 - m // 2 + numpy.arange(m) + s) % yN] = True
        assert inside[used].all()
        assert used[start] and used[(start + length - 1) % yN], trial  # tight at both ends
        # no other cyclic range that covers `used` is shorter: the complement of the band is the largest unused run
        ring = numpy.concatenate([used, used])
        longest = max(len(run) for run in "".join("u" if u else "." for u in ring).split("u"))
        assert length == yN - min(longest, yN - used.sum())
        # default alignment: the band starts up to 31 columns early so that every window origin is a multiple of 32
        # columns from the start (both parity runs of a 64-column tile then begin on a 128-byte line)
        a_start, a_length = band_range(N, yN, m, offs)
        shift = (start - a_start) % yN
        assert shift < 32 and a_length == min(yN, length + shift) or (a_start, a_length) == (0, yN)
        if (a_start, a_length) != (0, yN):
            for off in offs:
                s = off * yN // N
                assert ((yN // 2 - m // 2 + s - a_start) + (-s % m)) % 32 == 0
    offs = [i * 928 for i in range(-12, 13)]
    assert band_range(N, yN, m, offs, align=1)[1] == 24 * 464 + 512
    assert band_range(N, yN, m, [i * 928 for i in range(71)]) == (0, yN)


def test_mixed_factor_covers_every_catalogue_length():
    """the lengths that take the radix-Q pass (csrc/swiftly_mixed.h): every transform length of the 244 catalogue
    entries is a power of two or Q * 2^k with Q in {3, 5, 7, 9}"""
    from ska_sdp_exec_swiftly_amd.core_hip import mixed_factor
    from ska_sdp_exec_swiftly_amd.swift_configs import SWIFT_CONFIGS

    assert mixed_factor(6144) == (3, 11) and mixed_factor(57344) == (7, 13) and mixed_factor(36864) == (9, 12)
    assert mixed_factor(160) == (5, 5) and mixed_factor(224) == (7, 5)
    assert mixed_factor(4096) is None and mixed_factor(11264) is None and mixed_factor(15 * 64) is None
    assert mixed_factor(24) == (3, 3) and mixed_factor(12) is None  # sub-transforms of at least 8 points
    kinds = set()
    for c in SWIFT_CONFIGS.values():
        for n in (c["yN_size"], c["xM_size"], c["xM_size"] * c["yN_size"] // c["N"]):
            f = mixed_factor(n)
            assert f is not None or n & (n - 1) == 0, n
            kinds.add(1 if f is None else f[0])
    assert kinds == {1, 3, 5, 7, 9}


def test_band_range_and_row_sources_for_a_padded_facet_of_3_times_2_to_the_k():
    """The host tables of the band schedules carry no power-of-two assumption (yN = 3 * 256, catalogue entry
    1536[1]-n768-512: the backward band accumulators are pruned bands for every yN): band_range covers every window and
    is tight; the gather-sum tables reproduce the oracle's add_to_facet."""
    from ska_sdp_exec_swiftly_amd.core_hip import band_range, build_row_sources

    W, N, xM, yN = 11.0, 1536, 512, 768
    ref = orc.OracleCore(W, N, xM, yN)
    m = ref.xM_yN_size
    assert m == 256 and ref.subgrid_off_step == 2
    offs = [0, 448, 2 * 448, -448, 1536 + 448]
    band_offs = [0, 448, -448]  # three windows: 704 of the 768 columns
    start, length = band_range(N, yN, m, band_offs, align=1)
    used = numpy.zeros(yN, dtype=bool)
    for off in band_offs:
        used[(yN // 2 - m // 2 + numpy.arange(m) + off * yN // N) % yN] = True
    inside = numpy.zeros(yN, dtype=bool)
    inside[(start + numpy.arange(length)) % yN] = True
    assert inside[used].all() and used[start] and used[(start + length - 1) % yN] and length < yN
    rng = numpy.random.default_rng(9)
    blocks = rng.standard_normal((len(offs), m, 3)) + 1j * rng.standard_normal((len(offs), m, 3))
    want = numpy.zeros((yN, 3), dtype=complex)
    for b, off in enumerate(offs):
        ref.add_to_facet(blocks[b], off, axis=0, out=want)
    got = numpy.zeros((yN, 3), dtype=complex)
    flat = blocks.reshape(-1, 3)
    for members, tab in build_row_sources(N, yN, m, offs, [(0, b) for b in range(len(offs))]):
        assert tab.shape == (2, yN)
        for lvl in range(2):
            rows = numpy.flatnonzero(tab[lvl] >= 0)
            got[rows] += flat[tab[lvl, rows] & 0xFFFFF]
    numpy.testing.assert_array_equal(got, want)


def test_planned_wave_prediction_follows_the_walk_direction():
    """SwiftlyForward._predict_next_wave (host logic of the planned-wave prefetch): the successor IN PLAN ORDER (first
    appearance of the wave keys in subgrid_configs) of the wave being served, the predecessor once the caller walks
    the plan backwards, the same direction after a repeated key, nothing at either end, for an unplanned key, without
    a plan or with the prefetch switched off."""
    fwd = object.__new__(api.SwiftlyForward)  # the predictor only reads the plan bookkeeping
    fwd._plan = [api.SubgridConfig(0, k, 8) for k in (30, 10, 20, 10, 40)]  # plan order of the keys: 30, 10, 20, 40
    fwd._planned_keys = {10, 20, 30, 40}
    assert fwd._predict_next_wave(30) == 10
    assert fwd._predict_next_wave(10) == 20
    assert fwd._predict_next_wave(10) == 20        # the same wave again (a partial request): direction kept
    assert fwd._predict_next_wave(40) is None      # end of the plan
    assert fwd._predict_next_wave(20) == 10        # 40 -> 20: the caller turned round
    assert fwd._predict_next_wave(20) == 10        # repeated key while walking backwards
    assert fwd._predict_next_wave(10) == 30
    assert fwd._predict_next_wave(30) is None      # start of the plan, walking backwards
    assert fwd._predict_next_wave(10) == 20        # forwards again
    assert fwd._predict_next_wave(25) is None      # not a planned wave
    off = object.__new__(api.SwiftlyForward)
    off._plan = None
    assert off._predict_next_wave(10) is None
    old = api._PREFETCH
    api._PREFETCH = False
    try:
        assert fwd._predict_next_wave(10) is None
    finally:
        api._PREFETCH = old


def test_mispredicted_prefetch_is_dropped_and_switches_the_prefetch_off():
    """SwiftlyForward._take_prefetched: a prefetched wave that is not the one asked for (and the one asked for is not
    cached) is a misprediction -- the buffer is dropped (parked until its K2 has finished); after two of them IN A ROW the
    predictor stops predicting, and a caller that follows the plan again for a few requests gets it back (r6)."""
    fwd = object.__new__(api.SwiftlyForward)
    fwd._plan = [api.SubgridConfig(0, k, 8) for k in (10, 20, 30, 40, 50, 60, 70)]
    fwd._planned_keys = {10, 20, 30, 40, 50, 60, 70}
    fwd.lru = api.LRUCache(1)
    fwd.__dict__["_prefetched"] = {20: ("Q20", None, None)}
    fwd._take_prefetched(40)                       # asked for 40, 20 was prefetched
    assert not fwd.__dict__["_prefetched"] and fwd.__dict__["_prefetch_missed"] == 1
    assert fwd.__dict__["_prefetch_parked"] == [("Q20", None, None)]
    assert fwd._predict_next_wave(10) == 20        # one miss: still predicting
    fwd.__dict__["_prefetched"] = {20: ("Q20", None, None), 30: ("Q30", None, None)}
    fwd.lru.set(("b", 30), ("Q30", None))
    fwd._take_prefetched(30)                       # a hit: the other prefetched wave stays, the miss count starts again
    assert list(fwd.__dict__["_prefetched"]) == [20] and fwd.__dict__["_prefetch_missed"] == 0
    fwd._take_prefetched(40)
    assert fwd.__dict__["_prefetch_missed"] == 1 and not fwd.__dict__.get("_prefetch_off")
    fwd.__dict__["_prefetched"] = {20: ("Q20", None, None)}
    fwd._take_prefetched(50)
    assert fwd.__dict__.get("_prefetch_off") and fwd._predict_next_wave(10) is None
    # four requests in plan order switch it on again
    assert [fwd._predict_next_wave(k) for k in (20, 30, 40)] == [None, None, None]
    assert fwd._predict_next_wave(50) == 60 and not fwd.__dict__["_prefetch_off"]


def test_announced_wave_order_and_the_next_pass_of_a_reused_object():
    """set_wave_order: the predictor follows the order the caller announces (the multi-GPU pass packs its waves group
    by group); a jump from the last wave of the order back to the first is the next pass of the same walk, not a turn."""
    fwd = object.__new__(api.SwiftlyForward)
    fwd._plan = [api.SubgridConfig(0, k, 8) for k in (10, 20, 30, 40)]
    fwd._planned_keys = {10, 20, 30, 40}
    fwd.set_wave_order([30, 10, 99, 40, 20, 10])   # 99 is not planned, the second 10 is a repeat
    assert fwd._wave_order == [30, 10, 40, 20]
    assert fwd._predict_next_waves(30, 2) == [10, 40]
    assert fwd._predict_next_waves(10, 2) == [40, 20]
    assert fwd._predict_next_waves(20, 2) == []
    assert fwd._predict_next_waves(30, 2) == [10, 40]   # wrapped round: still walking forwards
    assert fwd._predict_next_waves(10, 1) == [40]


def test_planned_wave_prediction_depth():
    """SwiftlyForward._predict_next_waves: up to `depth` waves ahead in the walk direction, nearest first, cut at the
    ends of the plan (the r5 free-running K2 chain, SWIFTLY_PREFETCH_DEPTH)."""
    fwd = object.__new__(api.SwiftlyForward)
    fwd._plan = [api.SubgridConfig(0, k, 8) for k in (10, 20, 30, 40, 50)]
    fwd._planned_keys = {10, 20, 30, 40, 50}
    assert fwd._predict_next_waves(10, 2) == [20, 30]
    assert fwd._predict_next_waves(30, 3) == [40, 50]
    assert fwd._predict_next_waves(50, 2) == []
    assert fwd._predict_next_waves(40, 2) == [30, 20]   # turned round
    assert fwd._predict_next_waves(20, 2) == [10]
    assert fwd._predict_next_waves(15, 2) == []          # not a planned wave


def test_backward_wave_entry_points_check_the_wave_key():
    """(r4 advice) SwiftlyBackward.accumulate_wave / accumulate_chunks fold a wave under ONE key -- off1 in the band
    schedule, off0 in the reference's: subgrids that do not share the key of the schedule the object resolved to must
    raise instead of being placed at the first subgrid's offset.  (Checked before anything touches the device.)"""
    import types

    bwd = object.__new__(api.SwiftlyBackward)
    bwd._auto_axis = False
    bwd.wave_axis = 1
    bwd.core = types.SimpleNamespace(xM_yN_size=4, yN_size=16)
    bwd.facets_config_list = [api.FacetConfig(0, 0, 8)]
    same_off0 = [api.SubgridConfig(0, 0, 8), api.SubgridConfig(0, 8, 8)]  # the reference's accumulate_column grouping
    with pytest.raises(ValueError, match="share off1"):
        bwd.accumulate_wave(same_off0, numpy.zeros((1, 2, 4, 4), dtype=numpy.complex64))
    with pytest.raises(ValueError, match="share off1"):
        bwd.accumulate_chunks(0, [(same_off0, numpy.zeros((1, 2, 4, 4), dtype=numpy.complex64))])
    bwd.wave_axis = 0
    same_off1 = [api.SubgridConfig(0, 0, 8), api.SubgridConfig(8, 0, 8)]
    with pytest.raises(ValueError, match="share off0"):
        bwd.accumulate_wave(same_off1, numpy.zeros((1, 2, 4, 4), dtype=numpy.complex64))


def test_trace_timeline_gap_accounting(tmp_path, capsys):
    """tools/trace_timeline.py gaps: busy / idle / overlap accounting of a dispatch timeline split into passes at the K1
    launches (the tool behind profiles/r5_timeline_64k_sparse.txt), on a hand-made timeline."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "trace_timeline.py")
    spec = importlib.util.spec_from_file_location("trace_timeline", path)
    tl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tl)
    assert tl.short("_ZN3swf20row_pass_band_kernelINS_7RGeoPreILi14ELi5ELb1ELb1EEELb1ELi1ELb1ELi22ELi1EEEvNS_11RowPassArgsE") == "K1_row_pass_band"
    assert tl.short("_ZN3swf15col_pass_kernelINS_4CGeoILi7ELi5ELb1ELi64ELi4EEELi0ELb1ELb0EfEEvNS_11ColPassArgsE") == "col_pass<128,c64,mode0>"
    rows = []
    t = 0
    for _ in range(2):  # two passes: 2 K1 launches back to back, then 10 waves of two overlapping column passes + a gap
        for _ in range(2):
            rows.append((t, t + 1000_000, 1, "K1_row_pass_band"))
            t += 1000_000
        for _ in range(10):
            rows.append((t, t + 100_000, 1, "col_pass<128,c64,mode0>"))
            rows.append((t + 50_000, t + 150_000, 2, "col_pass<256,c64,mode1>"))
            t += 150_000 + 30_000  # 30 us of idle behind every wave
    import csv as csvmod

    path_csv = tmp_path / "tl.csv"
    with open(path_csv, "w", newline="") as fh:  # (the kernel names contain commas: quoted as `dump` writes them)
        csvmod.writer(fh).writerows(rows)
    tl.gaps(str(path_csv))
    out = capsys.readouterr().out
    assert out.count("pass ") == 2
    first = out.splitlines()[0]
    # 2 ms of K1 + 10 x 150 us of waves, 9 gaps of 30 us inside the pass (the 10th follows its last dispatch), 50 us of
    # every wave with two kernels running
    assert "wall 3.770 ms" in first and "busy 3.500 ms" in first and "idle 0.270 ms" in first
    assert ">=2 kernels running 0.500 ms" in first
