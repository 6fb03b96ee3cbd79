#!/bin/bash
# Round-6 evidence run on the GPU box (outputs under gpurun_out/, copied to profiles/ by hand):
#   C2 headline and C3's launch shape on the GPU-built 300 k reference-order graph: kernel stats, HBM and SQ counters
#   (scripts/collect_profiles.sh); the reference-order 1 M build checked against the oracle's fixture; kernel trace of the
#   windowed build's rounds (300 k build, and 12 k inserts at the 1 M end state)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd $R
for W in ${WORKLOADS:-c2 c3ref build trace}; do
  case $W in
    c2) bash scripts/collect_profiles.sh r6_c2 1024 > gpurun_out/r6/collect_c2.log 2>&1 ;;
    c3ref) bash scripts/collect_profiles.sh r6_c3ref 4096 --workload c3ref --streams 1 > gpurun_out/r6/collect_c3ref.log 2>&1 ;;
    build) python scripts/exact_build_check.py > gpurun_out/r6/c5_exact_build_1m.json 2> gpurun_out/r6/c5_exact_build_1m.err; tail -c 400 gpurun_out/r6/c5_exact_build_1m.json ;;
    trace)
      O=gpurun_out/r6/c5_occ_round_kernel_stats.txt
      echo "# commit $(cat .commit_for_profiles 2>/dev/null)" > $O
      echo "# rocprofv3 --kernel-trace --stats -- python scripts/occ_probe.py 300000 128 16 200 64 0   (the windowed reference-order build of 300 k x 128, M 16, ef 200)" >> $O
      rm -rf /tmp/prof_occ; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_occ -o p -- python scripts/occ_probe.py 300000 128 16 200 64 0 2>&1 | grep "^N=\|^group commit\|^a dry run\|^parallel group" >> $O
      python scripts/summarize_rocprof.py stats /tmp/prof_occ/p_results.db | head -12 >> $O
      echo "# rocprofv3 --kernel-trace --stats -- python scripts/occ_at_scale.py 12288   (12 032 reference-order inserts into the imported 1 M graph: the end state of BASELINE config 5)" >> $O
      rm -rf /tmp/prof_occ; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_occ -o p -- python scripts/occ_at_scale.py 12288 2>&1 | grep "inserts at\|   groups\|rounds ended" >> $O
      python scripts/summarize_rocprof.py stats /tmp/prof_occ/p_results.db | head -9 >> $O
      cat $O ;;
  esac
done
