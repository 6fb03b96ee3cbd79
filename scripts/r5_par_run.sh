cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_occ -o p -- python scripts/occ_probe.py 300000 128 16 200 32 0 > gpurun_out/r5n/probe.log 2>&1
grep "^N=\|^group\|^a dry\|^parallel" gpurun_out/r5n/probe.log
python scripts/summarize_rocprof.py stats /tmp/prof_occ/p_results.db > gpurun_out/r5n/occ_300k_kernel_stats.txt
head -12 gpurun_out/r5n/occ_300k_kernel_stats.txt | cut -c1-150
