cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6fz
echo "# commit $(cat .commit_for_profiles)  python scripts/fuzz_campaign.py 800 960000" > gpurun_out/r6fz/fuzz_campaign.txt
timeout 1100 python scripts/fuzz_campaign.py 800 960000 2>&1 | grep -v amdgpu.ids | grep -v "^ok\|^  " >> gpurun_out/r6fz/fuzz_campaign.txt
tail -4 gpurun_out/r6fz/fuzz_campaign.txt
echo "# python scripts/fuzz_build.py 300 8100" >> gpurun_out/r6fz/fuzz_campaign.txt
timeout 500 python scripts/fuzz_build.py 300 8100 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r6fz/fuzz_campaign.txt; tail -6 gpurun_out/r6fz/fuzz_campaign.txt
echo "# python scripts/del_campaign.py 14 1" >> gpurun_out/r6fz/fuzz_campaign.txt
timeout 400 python scripts/del_campaign.py 14 1 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r6fz/fuzz_campaign.txt; tail -3 gpurun_out/r6fz/fuzz_campaign.txt
