"""What tuning tie_mode costs: the 20 k x 128 reference-order build (SURVEY's model shape) in the total order and with the flagged
inserts redone in std's heap order, and a 4 096-query batch on the result.  GPU box; prints one JSON object."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from redis_hnsw_amd import index as eng                                    # noqa: E402
from tests.golden_util import load_transcribed                              # noqa: E402
from tests.util import graphs_equal, make_data                              # noqa: E402

c = load_transcribed("transcribed_20k_dim128")
out = {}
for mode in (0, 1, 0, 1):
    gi = eng.Index("cost%d" % mode, c["dim"], c["m"], c["ef"])
    gi.set_tuning("tie_mode", mode)
    t0 = time.perf_counter()
    gi.add_batch(c["V"], levels=c["levels"], mode="exact")
    dt = time.perf_counter() - t0
    lib = eng._capi.load()
    lib.hnsw_debug_tie_redone.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    red = C.c_uint64(0)
    lib.hnsw_debug_tie_redone(gi._h, C.byref(red))
    rec = dict(build_seconds=round(dt, 2), inserts_per_s=round(c["n"] / dt, 1), redone=int(red.value))
    if mode == 1:
        rec["equals_rust_golden"] = bool(graphs_equal(c["graph"], gi.export_graph())[0])
    Q = make_data(4096, c["dim"], seed=5)
    gi.search_batch(Q[:64], 10)
    gi.reset_counters()
    t0 = time.perf_counter()
    gi.search_batch(Q, 10)
    rec["search_4096_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    if mode == 1:
        rec["queries_flagged"] = gi.tie_counters()["queries_with_tie"]
    out["tie_mode_%d" % mode] = rec                                         # (the second pass of each mode: warm)
    gi.close()
print(json.dumps(out))
