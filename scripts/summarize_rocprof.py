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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
