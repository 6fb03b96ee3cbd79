#!/bin/bash
# kernel-level view of what tuning tie_mode costs (scripts/tie_mode_cost.py under rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out/tm
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tm -o tm -- python scripts/tie_mode_cost.py > gpurun_out/tm/run.log 2>&1 < /dev/null
tail -1 gpurun_out/tm/run.log | cut -c1-400
F=$(find /tmp/tm -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" gpurun_out/tm/kernel_stats.csv; head -16 "$F" | cut -c1-260; else find /tmp/tm -type f | head; fi
