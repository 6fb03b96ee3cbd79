cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5par
timeout 900 python scripts/exact_build_check.py > gpurun_out/r5par/build_1m_par.json 2> gpurun_out/r5par/build_1m_par.err
cat gpurun_out/r5par/build_1m_par.json
TUNING=occ_window=64,occ_ahead_x10=25 timeout 900 python scripts/exact_build_check.py > gpurun_out/r5par/build_1m_par_w64.json 2> gpurun_out/r5par/build_1m_par_w64.err
cat gpurun_out/r5par/build_1m_par_w64.json
