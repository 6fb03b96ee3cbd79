cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5fz
timeout 1300 python scripts/fuzz_campaign.py 1200 920000 2>&1 | grep -v amdgpu.ids > gpurun_out/r5fz/fuzz_campaign.txt
tail -4 gpurun_out/r5fz/fuzz_campaign.txt
timeout 700 python scripts/fuzz_build.py 600 7000 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r5fz/fuzz_build.txt; cat gpurun_out/r5fz/fuzz_build.txt
