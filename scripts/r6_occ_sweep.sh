cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_ties.py -x -q 2>&1 | tail -15
