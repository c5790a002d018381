"""Per-kernel roofline table of the forward pass from three rocprofv3 runs of the SAME build and command
(`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-backward`):

    python tools/pmc_kernels.py <kernel_trace.db> <pmc_fetch.db> <pmc_write.db> [out.json] [workload]

  kernel_trace.db : rocprofv3 --kernel-trace                      -> average launch durations (un-profiled clocks)
  pmc_fetch.db    : rocprofv3 --kernel-trace --pmc FETCH_SIZE     -> bytes read from the memory side of L2 per launch
  pmc_write.db    : rocprofv3 --kernel-trace --pmc WRITE_SIZE     -> bytes written per launch

(FETCH_SIZE / WRITE_SIZE need separate passes on gfx950; FETCH_SIZE tallies 64 B per 128 B request and is doubled;
MI355X_MICROARCH.md, section HBM.)  Writes profiles/r6_pmc_kernels.json, which bench.py reads for `roofline.traffic`,
`roofline.sustained` and the `kernels` table of the JSON line.  A sixth argument `backward` selects the stage table of
the subgrid -> facet direction (trace of tools/run_backward.py; key "<workload>:backward", `backward.kernels`)."""
import collections
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel-name pattern -> stage id of DESIGN.md section 4 (first match wins)
STAGES = [
    (r"row_pass_band_kernel.*Lb1ELi1E", "K1"),            # HAS_WIN, ST = 1 (band store)
    (r"col_pass_kernel.*CGeoILi[5-9]E.*EELi0ELb", "K2a"),   # four-step pass A (mapped load, raw store)
    (r"col_pass_kernel.*CGeoILi[5-9]E.*EELi1ELb", "K2b"),   # four-step pass B
    (r"col_pass_kernel.*CGeoILi9E.*EELi2ELb", "K3"),        # transform_contributions (512-point single pass)
    (r"sum_finish_facets_kernel", "K4b5a"),
    (r"col_pass_kernel.*CGeoILi10E.*EELi2ELb", "K5b"),      # axis-0 finish (1024-point single pass)
]


# the subgrid -> facet direction, band schedule: the mirror stages, same byte model
STAGES_BACKWARD = [
    (r"row_pass_band_kernel.*Lb0ELi2E", "B8"),             # finish_facet along the contiguous axis (mapped store)
    (r"col_pass_kernel.*CGeoILi[5-9]E.*EELi0ELb", "B5a"),   # gather-sum four-step pass A (add_to_facet axis 0 in the load)
    (r"col_pass_kernel.*CGeoILi[5-9]E.*EELi1ELb", "B5b"),   # four-step pass B (crop x Fb x mask0, column scatter-add)
    (r"col_pass_kernel.*CGeoILi9E.*EELi2ELb", "B4"),        # in-place m-point column pass of extract_from_subgrid axis 0
    (r"split_prepare_facets_kernel", "B2"),                 # prepare_subgrid axis 1 + extract_from_subgrid axis 1 per facet
    (r"col_pass_kernel.*CGeoILi10E.*EELi2ELb", "B1"),       # prepare_subgrid axis 0 (1024-point single pass)
]
_ACTIVE = STAGES


def stage_of(name):
    for pat, st in _ACTIVE:
        if re.search(pat, name):
            return st
    return None


def per_kernel(path, counter=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda pre: next(t for t in tabs if t.startswith(pre))  # noqa: E731
    disp, sym = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    rows = cur.execute(
        f"select d.event_id, s.kernel_name, d.start, d.end from '{disp}' d join '{sym}' s on d.kernel_id = s.id"
    ).fetchall()
    vals = collections.defaultdict(float)
    if counter:
        pev, pinfo = tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
        for ev, val in cur.execute(
            f"select e.event_id, e.value from '{pev}' e join '{pinfo}' i on e.pmc_id = i.id where i.name = ?", (counter,)
        ):
            vals[ev] += val
    agg = {}
    for ev, name, st, en in rows:
        stg = stage_of(name)
        if stg is None:
            continue
        a = agg.setdefault(stg, dict(n=0, ns=0, kib=0.0, name=name[:120]))
        a["n"] += 1
        a["ns"] += en - st
        a["kib"] += vals.get(ev, 0.0)
    return agg


def main():
    global _ACTIVE  # pylint: disable=global-statement
    kt, pf, pw = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "r6_pmc_kernels.json")
    workload = sys.argv[5] if len(sys.argv) > 5 else "64k-sparse"
    direction = sys.argv[6] if len(sys.argv) > 6 else "forward"
    _ACTIVE = STAGES_BACKWARD if direction == "backward" else STAGES
    t, f, w = per_kernel(kt), per_kernel(pf, "FETCH_SIZE"), per_kernel(pw, "WRITE_SIZE")
    table = {}
    for stg in [s for _, s in _ACTIVE]:
        if stg not in t or stg not in f or stg not in w:
            continue
        fetch = 2.0 * 1024.0 * f[stg]["kib"] / f[stg]["n"]
        write = 1024.0 * w[stg]["kib"] / w[stg]["n"]
        table[stg] = dict(
            kernel=t[stg]["name"],
            launches_in_trace=t[stg]["n"],
            avg_us=round(t[stg]["ns"] / t[stg]["n"] / 1e3, 2),
            avg_us_under_pmc=[round(f[stg]["ns"] / f[stg]["n"] / 1e3, 2), round(w[stg]["ns"] / w[stg]["n"] / 1e3, 2)],
            fetch_bytes_per_launch=int(fetch),
            write_bytes_per_launch=int(write),
            counter_bytes_per_launch=int(fetch + write),
        )
    try:
        with open(out, encoding="utf-8") as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        rec = {}
    # which build this was collected on (r4): the source hash compiled into the library, the hash of the .so file and
    # the git commit stamped by the Makefile; bench.py reports "stale" when the running library differs
    sys.path.insert(0, os.path.join(ROOT, "ska-sdp-distributed-fourier-transform_amd"))
    from ska_sdp_exec_swiftly_amd import _lib  # noqa: E402  pylint: disable=import-outside-toplevel

    how = ("rocprofv3 --kernel-trace (durations) and --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on "
           "`%s` of this build (%s); per-launch averages over all launches of the kernel in the run; FETCH_SIZE doubled "
           "per MI355X_MICROARCH.md (64 B tallied per 128 B request on gfx950)")
    if direction == "backward":
        key = workload + ":backward"
        note = how % ("tools/run_backward.py", "band schedule of SwiftlyBackward on random subgrids of the workload's plan, "
                      "no forward pass in the process, one kernel at a time")
    else:
        key = workload
        note = how % ("SWIFTLY_PREFETCH=0 SWIFTLY_K2_CHUNK=0 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify "
                      "--no-backward", "every kernel alone on the chip; the production pass overlaps K2 of the next wave with "
                      "K3-K5 and the two K2 passes of different chunks")
    rec[key] = dict(build=_lib.build_info(), kernels=table, note=note)
    with open(out, "w", encoding="utf-8") as fh:
        json.dump(rec, fh, indent=1)
    for stg, e in table.items():
        gbs = e["counter_bytes_per_launch"] / (e["avg_us"] * 1e-6) / 1e9
        print(f"{stg:6s} {e['avg_us']:9.1f} us  fetch {e['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {e['write_bytes_per_launch'] / 1e6:9.1f} MB"
              f"  sustained {gbs:7.0f} GB/s  x{e['launches_in_trace']}")


if __name__ == "__main__":
    main()
