cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "c3_search or parallel_validated or exact_insert_builds or delete" --durations=5 > gpurun_out/r5k/tests.log 2>&1
tail -12 gpurun_out/r5k/tests.log
