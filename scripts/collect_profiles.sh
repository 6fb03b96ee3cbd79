#!/bin/bash
# Collect the rocprofv3 evidence for one bench.py workload on the GPU box and leave only small text summaries
# under gpurun_out/profiles_$TAG/ (the rocpd databases are too large to copy back).
# Usage: scripts/collect_profiles.sh TAG BATCH [bench args...]   e.g.  r5_c2 1024   |   r5_c3 4096 --workload c3 --graph fast
# BATCH = queries per timed launch: the counters are read from the dispatch group with that grid, never "the most frequent"
set -u
TAG=${1:-r5}; shift || true
BATCH=${1:-1024}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
TMP=/tmp/prof_$TAG
mkdir -p $OUT $TMP
cd /tmp && export TMPDIR=/tmp
cd $R
ARGS="--steps 40 --warmup 4 --no-cpu-baseline --no-extras --batch $BATCH $*"
PARGS="--steps 40 --warmup 4 --only-timed --batch $BATCH $*"   # the counter passes stop after the timed region
COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null || echo unknown)
echo "# commit $COMMIT" > $OUT/kernel_stats.txt
echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" >> $OUT/kernel_stats.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $TMP/stats -o p -- python bench.py $ARGS > $TMP/stats.log 2>&1
grep -E '^\{' $TMP/stats.log >> $OUT/kernel_stats.txt
python scripts/summarize_rocprof.py stats $TMP/stats/p_results.db >> $OUT/kernel_stats.txt
python scripts/summarize_rocprof.py dispatches $TMP/stats/p_results.db k_search >> $OUT/kernel_stats.txt
rm -rf $TMP/stats
echo "# commit $COMMIT" > $OUT/pmc_hbm.txt
echo "# rocprofv3 --pmc <counter> -- python bench.py $PARGS   (one counter per pass; kernels are serialised under --pmc)" >> $OUT/pmc_hbm.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C -d $TMP/$C -o p -- python bench.py $PARGS > $TMP/$C.log 2>&1
  python scripts/summarize_rocprof.py pmc $TMP/$C/p_results.db k_search >> $OUT/pmc_hbm.txt
done
python scripts/summarize_rocprof.py traffic $TMP/FETCH_SIZE/p_results.db $TMP/WRITE_SIZE/p_results.db k_search "$COMMIT" "$PARGS" $BATCH > $OUT/traffic_entry.json
rm -rf $TMP/FETCH_SIZE $TMP/WRITE_SIZE
echo "# calibration: scripts/calib_fetch.py streams 2 x 2e6 x 512 B = 2.048e9 B through the engine's row access pattern" >> $OUT/pmc_hbm.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $TMP/calib -o p -- python scripts/calib_fetch.py > $TMP/calib.log 2>&1
python scripts/summarize_rocprof.py pmc $TMP/calib/p_results.db k_metric >> $OUT/pmc_hbm.txt
rm -rf $TMP/calib
echo "# commit $COMMIT" > $OUT/pmc_sq.txt
echo "# rocprofv3 --pmc (8 SQ counters per pass) -- python bench.py $PARGS" >> $OUT/pmc_sq.txt
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $TMP/sq1 -o p -- python bench.py $PARGS > $TMP/sq1.log 2>&1
python scripts/summarize_rocprof.py pmc $TMP/sq1/p_results.db k_search >> $OUT/pmc_sq.txt; rm -rf $TMP/sq1
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH -d $TMP/sq2 -o p -- python bench.py $PARGS > $TMP/sq2.log 2>&1
python scripts/summarize_rocprof.py pmc $TMP/sq2/p_results.db k_search >> $OUT/pmc_sq.txt; rm -rf $TMP/sq2
head -30 $OUT/kernel_stats.txt | cut -c1-220
cat $OUT/pmc_hbm.txt | cut -c1-160
cat $OUT/traffic_entry.json
