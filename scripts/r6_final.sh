#!/bin/bash
# Round-6 final evidence run on the GPU box: smoke, the whole -m gpu suite, the driver's bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6final/smoke.log 2>&1; tail -1 gpurun_out/r6final/smoke.log
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r6final/gpu_suite.log 2>&1; tail -10 gpurun_out/r6final/gpu_suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r6final/bench.json 2> gpurun_out/r6final/bench.err; tail -2 gpurun_out/r6final/bench.err
