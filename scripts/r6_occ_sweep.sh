cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ties.py -x -q 2>&1 | tail -15
python scripts/occ_at_scale.py 12288 2>&1 | grep "inserts at"
python bench.py --steps 20 --warmup 5 --no-clustered > gpurun_out/bench_r6c.json 2> gpurun_out/bench_r6c.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r6c.json') if l.startswith('{')][0])
print(d["value"], d["roofline"]["frac"])
print({k:v for k,v in d["tie_census_engine"].items() if k!="note"}); print({k:v for k,v in d["gpu_exact_build"]["ties"].items() if k!="note"}, d["gpu_exact_build"]["inserts_per_s"]); print(d["gpu_exact_build_at_1m"]["ties"], d["gpu_exact_build_at_1m"]["inserts_per_s"])
c=d["cpu_baseline"]["tie_census"]; print({k:v for k,v in c.items() if k!="note"})
PY
