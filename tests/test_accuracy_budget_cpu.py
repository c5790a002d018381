"""
The complex64 error budget of the forward dataflow, pinned on the CPU (tests/accuracy_model.py: the HIP pipeline's axis
order and stored intermediates replayed with numpy / scipy on the probe configuration N = 8192, one dense facet, one
subgrid, against the complex128 oracle chain).  These numbers are what DESIGN.md section 2 and the tolerances in
bench.py rest on; r3's storage-floor figure (1.0e-5) came from a probe that trusted numpy's complex64 ifft to compute
in double and was 4x too high.
"""
import accuracy_model as am


def test_storage_floor_and_the_stages_that_set_the_float32_error():
    rows = am.budget()
    for name, v in rows.items():
        print(f"{v:.3e}  {name}")
    floor = rows["storage floor: float64 arithmetic, complex64 intermediates (no scratch)"]
    floor_scratch = rows["storage floor incl. the complex64 four-step scratch of K2"]
    all32 = rows["float32 arithmetic everywhere (r3 kernels)"]
    assert 1.5e-6 < floor < 3e-6            # 2.3e-6
    assert floor < floor_scratch < 4.5e-6   # 3.3e-6: the scratch is one more complex64 rounding at the amplified level
    assert 1.0e-5 < all32 < 1.6e-5          # 1.3e-5: what the float32 kernels measure on this configuration (1.28e-5)
    # the float32 ARITHMETIC of two stages sets the end-to-end error: K2 (yN-point transform along the strided axis) and
    # K3 (the m-point transform behind it) work on data amplified by BOTH facet windows; every other stage is harmless
    assert rows["float32 arithmetic in k2 only"] > 2.5 * floor_scratch
    assert rows["float32 arithmetic in k3 only"] > 2.5 * floor_scratch
    for st in ("k1", "sf", "k5"):
        assert rows[f"float32 arithmetic in {st} only"] < 1.1 * floor_scratch
    # float64 arithmetic in K2 alone is not enough, in K2 and K3 it is
    assert rows["float64 arithmetic in k2 only (r4 kernels)"] > 2 * floor_scratch
    both = am.chain_error(dict(k1=32, k2=64, k3=64, sf=32, k5=32))
    print(f"{both:.3e}  float64 arithmetic in k2 and k3 (column_precision = 64)")
    assert both < 1.15 * floor_scratch      # 3.5e-6


def test_axis1_first_order_removes_the_double_window_from_the_column_passes():
    """(r6; r5 review item 3) The transforms of the two axes commute (api_helper.py:81-99, 200-210).  With the contiguous
    axis finished per wave BEFORE the strided-axis transforms (K1 output -> window gather -> m-point transform x Fn, stored in
    complex64; tests/accuracy_model.py: forward_chain_reordered) K2 and K3 work on data that carries ONE facet window: the
    all-float32 error of the probe configuration falls from 1.3e-5 to 2.1e-6, the storage floor from 3.3e-6 to 3.7e-7, and
    no single stage's float32 arithmetic costs more than 1.5e-6.  This is the dataflow of SwiftlyConfig(axis1_first=True)
    (HIP kernels on the N = 65536 workload: 2.04e-6 against 1.03e-5)."""
    rows = am.reordered_budget()
    for name, v in rows.items():
        print(f"{v:.3e}  {name}")
    all32 = rows["reordered: float32 arithmetic everywhere"]
    floor = rows["reordered: storage floor (float64 arithmetic, complex64 intermediates + scratch)"]
    assert 1.5e-6 < all32 < 3e-6          # 2.07e-6
    assert 2e-7 < floor < 6e-7            # 3.7e-7
    for st in ("k1", "k2", "k3", "sf", "k5"):
        assert rows[f"reordered: float32 arithmetic in {st} only"] < 2e-6


def test_half_spectra_from_the_two_workgroup_k1_would_cost_part_of_the_gain():
    """(r6, measured and removed) The two-workgroup K1 holds one output parity per workgroup, so what IT can store per window
    are the two decimation-in-time HALF spectra; K2 / K3 would run on those and the radix-2 join + Fn follow.  The halves are
    aliased (u folded onto u + m/2), so the column passes -- and even the complex64 STORES -- round at the level of the window
    leakage that Fn has not yet suppressed: 7.4e-6 all-float32 on the probe configuration (default order 1.3e-5, finished
    rows 2.1e-6), 1.8e-6 with float64 arithmetic everywhere (3.7e-7 for finished rows).  HIP kernels of that form on the
    N = 65536 workload: 5.8e-6 at 41.0 ms.  What ships instead finishes the window in the epilogue of a WHOLE-ROW K1 (one
    workgroup owns both parities): 2.04e-6 at 39.0 ms."""
    all32 = am.halves_error()
    stores_only = am.halves_error(bits=dict(k1=64, k2=64, k3=64, sf=64, k5=64))
    print(f"{all32:.3e}  halves: float32 arithmetic everywhere")
    print(f"{stores_only:.3e}  halves: float64 arithmetic, complex64 intermediates")
    assert 5e-6 < all32 < 9.5e-6
    assert 1.2e-6 < stores_only < 2.6e-6
