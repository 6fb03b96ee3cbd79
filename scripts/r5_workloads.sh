#!/bin/bash
# sanity of the bench's workload presets after the round-5 refactor: c1 (one query per step) and c5 (the index BUILT on the GPU in
# the reference's order, then benched)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5w
python bench.py --workload c1 --steps 20 --warmup 3 --no-traffic 2>gpurun_out/r5w/c1.err | tail -1 > gpurun_out/r5w/c1.json
python bench.py --workload c5 --steps 20 --warmup 5 --no-extras --no-traffic 2>gpurun_out/r5w/c5.err | tail -1 > gpurun_out/r5w/c5.json
for f in c1 c5; do python -c "
import json; d=json.load(open('gpurun_out/r5w/$f.json')); print('$f', d['config']['workload'][:70], d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('build_seconds'), d['cpu_baseline']['value'] if d['cpu_baseline'] else None)"; done
tail -2 gpurun_out/r5w/c5.err
