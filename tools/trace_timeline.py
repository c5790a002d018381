"""Dispatch timeline of a rocprofv3 kernel trace (rocpd sqlite database): where the GPU is idle.

    python tools/trace_timeline.py dump  <results.db> <out.csv>     # on the GPU box: one line per dispatch
    python tools/trace_timeline.py gaps  <out.csv> [K1-pattern]      # anywhere: busy / idle accounting per pass

`dump` writes  start_ns,end_ns,queue,short_name  sorted by start.  `gaps` splits the trace into passes at the first
dispatch of a run of K1 launches (row_pass_band_kernel by default), and prints for every pass: wall time, the union of
the busy intervals, the idle time between dispatches (by size class), the time during which two or more kernels were
running, and per kernel the sum of its durations and its EXCLUSIVE time (alone on the chip)."""
import collections
import csv
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"row_pass_band_kernel", name)
    if m:
        return "K1_row_pass_band"
    if "row_pass_whole_kernel" in name:  # the K1 of the axis-1-first order
        return "K1_row_pass_whole"
    m = re.search(r"col_pass_kernelINS_4CGeoILi(\d+)ELi(\d+)ELb[01]ELi(\d+)ELi\d+EEELi(\d)", name)
    if m:
        return f"col_pass<{1 << int(m.group(1))},c{m.group(3)},mode{m.group(4)}>"
    m = re.search(r"(sum_finish_facets_kernel|split_prepare_facets_kernel)ILi(\d+)ELi(\d+)", name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    m = re.search(r"N3swf\d+(\w+?_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"at6native\d+(\w+?)I", name)
    if m:
        return "torch:" + m.group(1)[:40]
    return name[:60]


def dump(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info('{disp}')")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute(
        f"select d.start, d.end, d.{qcol}, s.kernel_name from '{disp}' d join '{sym}' s on d.kernel_id = s.id order by d.start"
    ).fetchall()
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        for st, en, q, name in rows:
            w.writerow([st, en, q, short(name)])
    print(f"{len(rows)} dispatches -> {out_path}")


def gaps(csv_path, k1_pattern="K1_row_pass"):
    rows = []
    with open(csv_path) as f:
        for st, en, q, name in csv.reader(f):
            rows.append((int(st), int(en), q, name))
    rows.sort()
    # passes: a pass starts at a K1 dispatch whose predecessor (in time) is not a K1 dispatch
    def is_k1(n):
        return k1_pattern in n or n == "row_pass_whole_kernel"

    starts = [i for i, r in enumerate(rows) if is_k1(r[3]) and (i == 0 or not is_k1(rows[i - 1][3]))]
    starts.append(len(rows))
    for p in range(len(starts) - 1):
        seg = rows[starts[p] : starts[p + 1]]
        if len(seg) < 20:
            continue
        # drop trailing dispatches that belong to whatever follows the pass (verification etc.): keep up to the last
        # sum_finish / col_pass dispatch
        last = max((i for i, r in enumerate(seg) if "col_pass" in r[3] or "sum_finish" in r[3]), default=len(seg) - 1)
        seg = seg[: last + 1]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        ev = []
        for st, en, q, name in seg:
            ev.append((st, 1, name))
            ev.append((en, -1, name))
        ev.sort(key=lambda e: (e[0], e[1]))
        depth, prev, busy, multi = 0, t0, 0, 0
        idle_classes = collections.Counter()
        idle_total = 0
        running = collections.Counter()
        excl = collections.Counter()
        for t, d, name in ev:
            dt = t - prev
            if depth == 0 and dt > 0:
                idle_total += dt
                cls = "<2us" if dt < 2000 else "<5us" if dt < 5000 else "<10us" if dt < 10000 else "<20us" if dt < 20000 else ">=20us"
                idle_classes[cls] += dt
                idle_classes["n" + cls] += 1
            elif depth >= 1:
                busy += dt
                if depth >= 2:
                    multi += dt
                else:
                    only = next(k for k, v in running.items() if v > 0)
                    excl[only] += dt
            prev = t
            depth += d
            running[name] += d
        tot = collections.Counter()
        cnt = collections.Counter()
        for st, en, q, name in seg:
            tot[name] += en - st
            cnt[name] += 1
        print(f"pass {p}: {len(seg)} dispatches, wall {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {idle_total / 1e6:.3f} ms, >=2 kernels running {multi / 1e6:.3f} ms")
        print("   idle by gap size:", {k: (round(v / 1e6, 3) if not k.startswith("n") else v) for k, v in sorted(idle_classes.items())})
        for name, v in sorted(tot.items(), key=lambda kv: -kv[1]):
            print(f"   {name:44s} n {cnt[name]:4d}  sum {v / 1e6:8.3f} ms  avg {v / cnt[name] / 1e3:8.1f} us  exclusive {excl[name] / 1e6:8.3f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3])
    else:
        gaps(sys.argv[2], *(sys.argv[3:4]))
