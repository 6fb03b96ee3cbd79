cd /root/repo
TUNINGS=";occ_slack_base=3;" python scripts/occ_at_scale.py 12288 2>&1 | grep -v "amdgpu.ids\|rounds ended"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ties.py -k "parallel_validated or windowed or exact or ties or census or unknown" -x -q 2>&1 | tail -5
