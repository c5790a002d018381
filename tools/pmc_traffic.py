"""HBM traffic per launch of the roofline kernel (K1 = row_pass_band_kernel) from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE must be collected separately on gfx950):

    python tools/pmc_traffic.py <fetch_results.db> <write_results.db> [out.json]

FETCH_SIZE tallies 64 B per 128 B request on gfx950 (MI355X_MICROARCH.md, section HBM): it is doubled.  Writes
profiles/r2_pmc_traffic.json, which bench.py reads for `roofline.traffic`."""
import json
import os
import sqlite3
import sys

KERNEL = "row_pass_band_kernel"


def per_launch(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    tab = lambda pre: next(t for t in tabs if t.startswith(pre))  # noqa: E731
    disp, sym, pev, pinfo = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    evs = [
        (ev, en - st)
        for ev, name, st, en in cur.execute(
            f"select d.event_id, s.kernel_name, d.start, d.end from '{disp}' d join '{sym}' s on d.kernel_id = s.id"
        )
        if KERNEL in name
    ]
    total = 0.0
    for ev, _ in evs:
        total += sum(
            v for (v,) in cur.execute(
                f"select e.value from '{pev}' e join '{pinfo}' i on e.pmc_id = i.id where e.event_id = ? and i.name = ?",
                (ev, counter),
            )
        )
    return total / len(evs), sum(d for _, d in evs) / len(evs) / 1e3, len(evs)


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r2_pmc_traffic.json")
    f_kib, f_us, n = per_launch(fetch_db, "FETCH_SIZE")
    w_kib, w_us, _ = per_launch(write_db, "WRITE_SIZE")
    total = int(2 * f_kib * 1024 + w_kib * 1024)
    rec = {
        "64k-sparse": {
            "kernel": KERNEL,
            "bytes_per_launch": total,
            "fetch_size_kib": f_kib,
            "write_size_kib": w_kib,
            "launches": n,
            "avg_us_under_pmc": [f_us, w_us],
            "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/run_fwd_kernels.py "
                    "(K1 of one 22528^2 facet of the 64k-sparse workload, same build); FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (64 B tallied per 128 B request on gfx950)",
        }
    }
    with open(out, "w", encoding="utf-8") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
