import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n, dim, m, ef, k = [int(x) for x in z["params"]]
    L = int(z["max_layer"]) + 1
    g = dict(levels=z["levels"].astype(np.uint32), enterpoint=int(z["enterpoint"]), max_layer=int(z["max_layer"]),
             row_ptr=[z["row_ptr_%d" % l] for l in range(L)], col=[z["col_%d" % l] for l in range(L)])
    return dict(n=n, dim=dim, m=m, ef=ef, k=k, V=z["vectors"], Q=z["queries"], levels=z["levels"], graph=g,
                ids=z["ids"], sims_bits=z["sims_bits"], n_out=z["n_out"],
                search_counters=z["search_counters"], insert_counters=z["insert_counters"],
                n_first=int(z["n_first"]) if "n_first" in z else n,
                deleted=z["deleted"] if "deleted" in z else np.zeros(0, dtype=np.int64))
