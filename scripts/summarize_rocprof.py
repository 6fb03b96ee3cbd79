#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (gpurun_out/prof_*/...) into the small text
summaries kept under profiles/.  Usage:
  summarize_rocprof.py stats <db>                 -> per-kernel table (like --stats)
  summarize_rocprof.py pmc <db> <kernel-substr>   -> per-counter averages for matching dispatches
"""
import sqlite3
import sys


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats  (durations in us, as the rocpd top_kernels view reports them)")
    print("%-64s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows[:25]:
        short = name.split("(")[0].replace("void ", "")
        print("%-64s %8d %14.0f %12.1f %6.2f%%" % (short[:64], calls, tot * 1.0, avg * 1.0, pct))


def pmc(db, sub):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, lds_block_size, grid_size, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
         "avg(vgpr_count) from counters_collection where kernel_name like ? "
         "group by kernel_name, lds_block_size, grid_size, counter_name order by kernel_name, lds_block_size, grid_size, counter_name")
    print("# rocprofv3 --pmc, dispatches of kernels matching %r (grouped by kernel, LDS bytes, grid)" % sub)
    last = None
    for r in cur.execute(q, ("%" + sub + "%",)):
        name = r[0].split("(")[0].replace("void ", "")
        head = (name, r[1], r[2])
        if head != last:
            print("kernel=%s lds=%d grid=%d vgpr=%d" % (name[:60], r[1], r[2], r[9]))
            last = head
        print("  counter=%s dispatches=%d avg=%.3f min=%.3f max=%.3f avg_dispatch_ns=%.0f" % (r[3], r[4], r[5], r[6], r[7], r[8]))


def dispatches(db, sub):
    """per (kernel, LDS bytes, grid) group of the kernel trace: launches and average duration -- the timed
    launches of bench.py are the largest group (the warm-up, recall and counter passes use other shapes)"""
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        print("# (no kernels view in this rocpd database: %s)" % ", ".join(n for n in names if "kernel" in n.lower()))
        return
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
    lds = "lds_size" if "lds_size" in cols else ("lds_block_size" if "lds_block_size" in cols else "0")
    grid = "grid_x" if "grid_x" in cols else ("grid_size" if "grid_size" in cols else "0")
    dur = "duration" if "duration" in cols else "(end - start)"
    q = ("select name, %s, %s, count(*), avg(%s), min(%s), max(%s) from %s where name like ? group by name, %s, %s "
         "order by count(*) desc" % (lds, grid, dur, dur, dur, view, lds, grid))
    print("# kernel-trace dispatches of kernels matching %r, grouped by (kernel, LDS bytes, grid); durations in ns" % sub)
    for r in cur.execute(q, ("%" + sub + "%",)):
        print("kernel=%s lds=%s grid=%s launches=%d avg_ns=%.0f min_ns=%.0f max_ns=%.0f" % (
            r[0].split("(")[0].replace("void ", "")[:60], r[1], r[2], r[3], r[4], r[5], r[6]))


def timed_rows(rows, batch):
    """the dispatch groups with the timed step's launch shape (bench.py: timed_launch_rows): one workgroup of 64
    threads per query, or the general kernel's grid-stride form (<= 2048 workgroups); never the two-wave kernel,
    never the 1024-query chunks of a host batch.  Most dispatches first."""
    want = {64 * batch, 64 * min(batch, 2048)}
    hit = [r for r in rows if int(r[2]) in want and "duo" not in r[0]]
    return sorted(hit, key=lambda r: -r[3])


def traffic(fetch_db, write_db, sub, commit, args, batch):
    """JSON entry for profiles/traffic.json: HBM bytes per launch of the k_search dispatch group whose grid is the
    TIMED step's (batch queries per launch) -- not the most frequent group: with --batch 4096 the 1024-query chunks of
    the host-buffer leg outnumber the timed launches.  FETCH_SIZE is KiB and reports half of a wide coalesced read on
    gfx950: x2, see MI355X_MICROARCH.md."""
    import json

    def top(db, counter):
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, lds_block_size, grid_size, count(*), avg(value) from counters_collection "
             "where kernel_name like ? and counter_name = ? group by kernel_name, lds_block_size, grid_size")
        rows = timed_rows(list(cur.execute(q, ("%" + sub + "%", counter))), batch)
        return rows[0] if rows else None
    f, w = top(fetch_db, "FETCH_SIZE"), top(write_db, "WRITE_SIZE")
    out = dict(commit=commit, bench_args=args, batch=batch, kernel=f[0].split("(")[0].replace("void ", "") if f else None,
               lds_bytes=f[1] if f else None, grid=f[2] if f else None, launches_sampled=f[3] if f else 0,
               fetch_size_kib_per_launch=f[4] if f else None, write_size_kib_per_launch=w[4] if w else None,
               fetch_correction=2.0)
    if f and w:
        out["k_search_hbm_bytes_per_launch"] = int(f[4] * 1024 * 2.0 + w[4] * 1024)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "dispatches":
        dispatches(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6], int(sys.argv[7]))
    else:
        pmc(sys.argv[2], sys.argv[3])
