"""Run tests/test_gpu_fuzz.py's differential op sequences over many random configurations (GPU box).
usage: python scripts/fuzz_campaign.py [seconds] [first_seed]"""
import sys
import time
import traceback

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as oracle_mod
from redis_hnsw_amd import index as eng
from tests.test_gpu_fuzz import test_random_op_sequences_match_the_oracle as run

oracle_mod.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
done = bad = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    kind = str(rng.choice(["uniform", "clustered", "line", "lattice", "dupes"]))
    dim = int(rng.choice([4, 12, 32, 33, 64, 96, 100, 128, 128, 128, 256, 768]))   # 128: the specialised search / plan routines; 768: the other register variant
    m = int(rng.choice([2, 3, 4, 5, 8, 12, 16, 24, 31, 32, 33, 40, 48, 64, 65, 100, 128]))   # above 64: the serial kernels
    ef = int(rng.choice([m, m + 1, 16, 40, 100, 200, 300]))
    ef = max(ef, 2)
    case = (kind, dim, m, ef, int(rng.choice([40, 80, 120])), seed)
    tun = []
    if rng.random() < 0.6:                                  # non-default engine paths, results must not change
        for key, vals in (("occ_window", [2, 5, 16, 64]), ("occ_log_cap", [40, 300, 3072]), ("select_shortcut", [0, 1]),
                          ("visited_bounded", [0, 1]), ("lean", [0, 1]), ("lds_hash_bits", [8, 10]),
                          ("occ_min_batch", [2, 64]), ("occ_ahead_x10", [10, 30]), ("waves_per_cu", [1, 4, 8]),
                          ("tag_table", [0, 1]), ("pipe_chunk", [64, 256, 1024]), ("grid_stride", [0, 1]),
                          ("force_restride", [16, 48]), ("plan_lean", [0, 1]), ("single_window", [0, 1]),
                          ("duo", [0, 1]), ("plan_duo", [0, 1]), ("commit_team", [0, 1]), ("commit_par", [0, 2, 2]), ("plan_split", [0, 1]), ("occ_chain", [1, 4, 8]),
                          ("occ_stage_ahead", [0, 8, 32]), ("occ_depth_x10", [0, 30]), ("occ_front_max", [3, 16]), ("par_max_resident", [3, 0]),
                          ("tie_census", [0, 1])):
            if rng.random() < 0.4:
                tun.append((key, int(rng.choice(vals))))
    case = case + (tuple(tun),)
    try:
        run(eng, oracle_mod, *case, stress=True)
    except Exception as e:                                  # report and go on: one line per failing case
        bad += 1
        print("FAIL", case, type(e).__name__, str(e).split("\n")[0][:200], flush=True)
        traceback.print_exc(limit=2)
    done += 1
    seed += 1
print("cases %d, failures %d, %.0f s" % (done, bad, time.time() - t0))
