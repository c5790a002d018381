"""Per-kernel statistics from a rocprofv3 rocpd database (the default output
format of rocprofv3 in ROCm 7.2):  python tools/rocpd_stats.py <results.db>
Equivalent to `rocprofv3 --kernel-trace --stats` kernel_stats, computed offline
so that the profiling command on the GPU box stays short."""
import collections
import re
import sqlite3
import sys


def demangle_short(name):
    m = re.search(r"fft_rows_kernelINS_3GeoI([fd])Li(\d+)ELi(\d+)ELi(\d+)ELb([01])", name)
    if m:
        ty = "float" if m.group(1) == "f" else "double"
        return f"swf::fft_rows_kernel<Geo<{ty}, N=2^{m.group(2)}, P=2^{m.group(3)}, NT={m.group(4)}, split={m.group(5)}>>"
    m = re.search(r"modcopy_kernelI([fd])Lb([01])", name)
    if m:
        return f"modcopy_kernel<{'float' if m.group(1) == 'f' else 'double'}, scatter={m.group(2)}>"
    m = re.search(r"(k\d_\w+?_kernel)", name)
    return name[:100]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = cur.execute(
        f"select s.kernel_name, d.start, d.end, s.arch_vgpr_count, s.sgpr_count, d.group_segment_size, "
        f"d.private_segment_size from '{disp}' d join '{sym}' s on d.kernel_id = s.id"
    ).fetchall()
    agg = collections.OrderedDict()
    for name, st, en, vg, sg, lds, scr in rows:
        key = demangle_short(name)
        a = agg.setdefault(key, dict(n=0, tot=0, mn=1 << 62, mx=0, vgpr=vg, sgpr=sg, lds=lds, scratch=scr))
        dur = en - st
        a["n"] += 1
        a["tot"] += dur
        a["mn"] = min(a["mn"], dur)
        a["mx"] = max(a["mx"], dur)
    total = sum(a["tot"] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, total kernel time {total / 1e6:.3f} ms")
    print(f"{'total_ms':>10} {'%':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'vgpr':>5} {'lds':>7} {'scratch':>7}  kernel")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        print(
            f"{a['tot'] / 1e6:10.3f} {100 * a['tot'] / total:6.2f} {a['n']:6d} {a['tot'] / a['n'] / 1e3:10.1f} "
            f"{a['mn'] / 1e3:10.1f} {a['mx'] / 1e3:10.1f} {a['vgpr']:5d} {a['lds']:7d} {a['scratch']:7d}  {key}"
        )


def timeline(path, pattern, count):
    """print `count` consecutive dispatches starting at the LAST-but-`count`... simple view: the dispatches around the
    last occurrence of a kernel whose name contains `pattern` (start offset, duration, gap to the previous end)"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = cur.execute(
        f"select s.kernel_name, d.start, d.end from '{disp}' d join '{sym}' s on d.kernel_id = s.id order by d.start"
    ).fetchall()
    hits = [i for i, r in enumerate(rows) if pattern in r[0]]
    if not hits:
        print("no dispatch matches", pattern)
        return
    i0 = max(0, hits[len(hits) // 2] - count // 2)
    t0 = rows[i0][1]
    prev_end = None
    for name, st, en in rows[i0 : i0 + count]:
        gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(st - t0) / 1e3:10.1f} us  dur {(en - st) / 1e3:8.1f} us  gap {gap:8.1f} us  {demangle_short(name)[:90]}")
        prev_end = en


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        timeline(sys.argv[1], sys.argv[2], int(sys.argv[3]))
    else:
        main(sys.argv[1])
