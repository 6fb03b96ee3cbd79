#!/bin/bash
# HNSW.SEARCH throughput by vector dimension (fast-built 200 k-node graphs, M=16, ef=200, k=10, three 1024-query launches
# in flight): which kernel serves the dim, QPS, fraction of 8 TB/s in the reference's algorithmic bytes.
# dims % 32 != 0 take the reference's scalar summation order (metrics.rs:79-84).   usage: bash scripts/dim_sweep.sh [nodes]
N=${1:-200000}
for D in 32 64 96 100 128 256 384 512 768 1024 1536; do
  python bench.py --nodes $N --dim $D --graph fast --steps 60 --warmup 5 --no-extras --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('dim %5d  %9.0f QPS  %.3f ms/step  frac %.3f  n_dist %.0f  recall %.3f  kernel_ms %.3f' % ($D, d['value'], d['ms_per_step'], r['frac'], r['n_dist_per_query'], d.get('recall_at_10') or -1, r['kernel_ms']))"
done
