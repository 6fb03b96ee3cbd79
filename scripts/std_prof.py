"""Phase times of k_insert_std_heap (a library built with -DHNSW_STD_PROF prints one line per insert): averages in microseconds
(100 MHz ticks / 100).  GPU box.  usage: python scripts/std_prof.py"""
import re
import subprocess
import sys

out = subprocess.run([sys.executable, "scripts/tie_mode_cost.py"], capture_output=True, text=True).stdout
rows = [list(map(int, re.findall(r"\d+", ln)[1:])) for ln in out.splitlines() if ln.startswith("STDPROF")]
print("inserts profiled:", len(rows))
if rows:
    names = ["search", "select", "connect", "econn", "shr_select", "shr_update", "n_dist"]
    for i, nm in enumerate(names):
        col = [r[i] for r in rows]
        print("%-11s mean %9.1f %s" % (nm, sum(col) / len(col) / (1.0 if nm == "n_dist" else 100.0), "" if nm == "n_dist" else "us"))
print(out.splitlines()[-1][:300])
