cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ties.py -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 --no-clustered > gpurun_out/bench_r6b.json 2> gpurun_out/bench_r6b.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r6b.json') if l.startswith('{')][0])
print(d["tie_census_engine"]); print(d["gpu_exact_build"]["ties"], d["gpu_exact_build"]["inserts_per_s"]); print(d["gpu_exact_build_at_1m"]["ties"], d["gpu_exact_build_at_1m"]["inserts_per_s"])
c=d["cpu_baseline"]["tie_census"]; print({k:v for k,v in c.items() if k!="note"})
PY
