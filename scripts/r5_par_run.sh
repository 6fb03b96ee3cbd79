cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
for T in "occ_ahead_x10=20" "occ_ahead_x10=18,commit_par_min_x10=60"; do
  TUNING=$T timeout 900 python scripts/exact_build_check.py > gpurun_out/r5c/build2_$T.json 2> gpurun_out/r5c/build2_$T.err
  echo "$T: $(python -c "
import json,sys; d=json.load(open('gpurun_out/r5c/build2_$T.json')); p=d.get('parallel_commit',{})
print(d['build_seconds'], d['rounds'], d['commits_per_round'], d['stale_plans'], p.get('groups_per_round'), p.get('nodes_per_group'), p.get('dry_runs_per_commit'), p.get('us_per_iteration_workgroup0'), d.get('identical_to_oracle_serial_build'))")"
done
