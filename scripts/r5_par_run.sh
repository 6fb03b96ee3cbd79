cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_robustness.py -m gpu -x -q > gpurun_out/r5b/tests.log 2>&1
tail -5 gpurun_out/r5b/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
tail -3 gpurun_out/r5b/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r5b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('hbm_measured_gather'))
print(d['gpu_exact_build']); print(d['single_add']); print(d['single_delete']); print(d['cpu_baseline']['tie_census'])
print(d['roofline']['lone_launch_1024']); print(d['device_call'])"
