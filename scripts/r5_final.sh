#!/bin/bash
# Round-5 final evidence run on the GPU box: smoke, the whole -m gpu suite, the driver's bench command, then the rocprofv3
# evidence of C2 / C3 / C4 at this tree (scripts/r5_profiles.sh).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r5final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5final/smoke.log 2>&1; tail -1 gpurun_out/r5final/smoke.log
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r5final/gpu_suite.log 2>&1; tail -10 gpurun_out/r5final/gpu_suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5final/bench.json 2> gpurun_out/r5final/bench.err; tail -2 gpurun_out/r5final/bench.err
WORKLOADS="c2 c3 c4" bash scripts/r5_profiles.sh > gpurun_out/r5final/profiles.log 2>&1
for c in c2 c3 c4; do cat gpurun_out/profiles_r5_$c/traffic_entry.json | tr -d '\n' | cut -c1-400; echo; done
