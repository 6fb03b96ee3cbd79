cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5suite
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_robustness.py tests/test_gpu_group.py tests/test_bench_gpu.py -m gpu -x -q > gpurun_out/r5suite/gpu_rest.log 2>&1; tail -5 gpurun_out/r5suite/gpu_rest.log
