"""Per-kernel PMC counter summary from a rocprofv3 rocpd database (ROCm 7.2 default output):
    python tools/pmc_stats.py <results.db> [more.db ...]
For every kernel: launches, average duration and, per counter, the average per launch of the sum over all
counter instances (XCCs / SEs).  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them (on gfx950 FETCH_SIZE
tallies 64 B per 128 B request: double it before comparing with byte counts, MI355X_MICROARCH.md)."""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^_ZN3swf", "", name)
    m = re.search(r"(\d+)(col_pass_kernel|row_pass_band_kernel|row_pass_split_kernel|row_pass_kernel|sum_finish_facets_kernel|"
                  r"sum_finish_rows_kernel|fft_rows_kernel|modcopy_kernel)(.*?)EEv", name)
    if m:
        return m.group(2) + "<" + re.sub(r"NS_\d?[A-Z]?Geo", "Geo", m.group(3))[:60] + ">"
    return name[:90]


def main(paths):
    for path in paths:
        db = sqlite3.connect(path)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        tab = lambda pre: next(t for t in tabs if t.startswith(pre))  # noqa: E731
        disp, sym, pev, pinfo = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
        kern = {}
        for ev, name, st, en in cur.execute(
            f"select d.event_id, s.kernel_name, d.start, d.end from '{disp}' d join '{sym}' s on d.kernel_id = s.id"
        ):
            kern[ev] = (short(name), en - st)
        vals = collections.defaultdict(lambda: collections.defaultdict(float))
        for ev, cname, val in cur.execute(f"select e.event_id, i.name, e.value from '{pev}' e join '{pinfo}' i on e.pmc_id = i.id"):
            vals[ev][cname] += val
        agg = collections.OrderedDict()
        for ev, (name, dur) in kern.items():
            a = agg.setdefault(name, dict(n=0, dur=0, c=collections.defaultdict(float)))
            a["n"] += 1
            a["dur"] += dur
            for k, v in vals.get(ev, {}).items():
                a["c"][k] += v
        print(f"== {path}")
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
            print(f"{name}: launches {a['n']}, avg {a['dur'] / a['n'] / 1e3:.1f} us")
            for k in sorted(a["c"]):
                print(f"    {k:<28} {a['c'][k] / a['n']:18.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])
