cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ties.py -k "tie_mode" -x -q 2>&1 | tail -15
