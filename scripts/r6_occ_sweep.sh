cd /root/repo
TUNINGS=";occ_stage_ahead=0;occ_stage_ahead=20;plan_split_x10=12;plan_split_x10=20" python scripts/occ_at_scale.py 12288 2>&1 | grep -v "amdgpu.ids\|groups \|rounds ended"
python scripts/occ_probe.py 50000 128 16 200 64 1 2>&1 | grep -v "amdgpu.ids\|^commit kernel\|^recomputed"
timeout 900 python -m pytest tests/test_gpu_parity.py -k "parallel_validated or windowed or exact" -x -q 2>&1 | tail -5
