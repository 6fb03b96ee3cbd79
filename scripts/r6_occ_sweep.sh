cd /tmp && export TMPDIR=/tmp
cd /root/repo
rm -rf /tmp/prof_occ
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_occ -o p -- python scripts/occ_probe.py 12000 768 32 400 64 0 2>&1 | grep -v "amdgpu.ids\|simple_timer\|generateRocpd" | tail -6
python scripts/summarize_rocprof.py stats /tmp/prof_occ/p_results.db | head -14
