import json,sys
for line in open(sys.argv[1]):
    line=line.strip()
    if line.startswith('{"metric"'):
        d=json.loads(line)
        b=d.get("bf16_storage_mode") or {}
        print(sys.argv[1], "f32 qps %.0f frac %.3f kernel_ms %.3f | bf16 qps %s ms %s frac %s recall %s"%(d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], b.get("value"), b.get("ms_per_step"), b.get("frac"), b.get("recall_at_10")))
