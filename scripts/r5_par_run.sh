cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -x -q > gpurun_out/r5f/tests.log 2>&1
tail -5 gpurun_out/r5f/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err
tail -4 gpurun_out/r5f/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r5f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['setup_seconds'])
print(d['gpu_exact_build_at_1m']); print(d['gpu_exact_build']['inserts_per_s'], d['single_add']['gpu_ms'], d['single_delete']['gpu_ms'])"
