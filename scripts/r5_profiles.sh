#!/bin/bash
# Round-5 evidence run on the GPU box: the gather microbenchmark, then kernel stats / HBM counters / SQ counters of
# C2, C3 and C4 at this tree (scripts/collect_profiles.sh).  Output under gpurun_out/ (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r5
cd $R
( echo "# commit $(cat .commit_for_profiles 2>/dev/null)"; scripts/microbench/gather_bw ) > gpurun_out/r5/gather_bw.txt 2>&1
for W in ${WORKLOADS:-c2 c3 c4}; do
  case $W in
    c2) bash scripts/collect_profiles.sh r5_c2 1024 > gpurun_out/r5/collect_c2.log 2>&1 ;;
    c3) bash scripts/collect_profiles.sh r5_c3 4096 --nodes 1000000 --dim 768 --m 32 --ef 400 --k 100 --graph fast --streams 1 > gpurun_out/r5/collect_c3.log 2>&1 ;;
    c4) bash scripts/collect_profiles.sh r5_c4 1024 --nodes 10000000 --dim 128 --m 16 --ef 200 --k 10 --graph fast > gpurun_out/r5/collect_c4.log 2>&1 ;;
  esac
done
tail -5 gpurun_out/r5/gather_bw.txt
