#!/bin/bash
# SQ instruction mix of the search kernels (specialised vs general) on one graph: two rocprofv3 --pmc passes.
# Usage: scripts/pmc_compare.sh <graph.npz|fast> <nodes> "<variants>"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
G=${1:-fast}; N=${2:-1000000}; V=${3:-"B=1024;B=1024,lean=0"}
TMP=/tmp/pmc_cmp; rm -rf $TMP; mkdir -p $TMP $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $R
OUT=$R/gpurun_out/pmc_compare.txt
echo "# python scripts/search_sweep.py --graph $G --nodes $N --reps 10 --variants \"$V\"" > $OUT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $TMP/a -o p -- python scripts/search_sweep.py --graph $G --nodes $N --reps 10 --variants "$V" > $TMP/a.log 2>&1
grep -E "ms/launch" $TMP/a.log >> $OUT
python scripts/summarize_rocprof.py pmc $TMP/a/p_results.db k_search >> $OUT
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH -d $TMP/b -o p -- python scripts/search_sweep.py --graph $G --nodes $N --reps 10 --variants "$V" > $TMP/b.log 2>&1
python scripts/summarize_rocprof.py pmc $TMP/b/p_results.db k_search >> $OUT
cat $OUT | cut -c1-170
