"""Dump a reference-order graph fixture (data/c2_ref_graph_*.npz) as raw files for tests/experiments/occ_model.c
(FIX=<dir>): the model then measures the yield of the windowed / grouped commits ON that graph instead of building a
small one first.  TEST INFRASTRUCTURE.   python tests/experiments/dump_fixture.py data/c2_ref_graph_1m.npz /tmp/c2fix"""
import os
import sys

import numpy as np

z = np.load(sys.argv[1])
out = sys.argv[2]
os.makedirs(out, exist_ok=True)
n, dim = int(z["nodes"]), int(z["dim"])
L = int(z["max_layer"]) + 1
np.random.default_rng(1).random((n, dim), dtype=np.float32).tofile(os.path.join(out, "vec.f32"))   # bench.py's base vectors
z["levels"].astype(np.uint32).tofile(os.path.join(out, "levels.u32"))
for l in range(L):
    rp = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(z["deg%d" % l].astype(np.uint64), out=rp[1:])
    rp.tofile(os.path.join(out, "rp%d.u64" % l))
    z["col%d" % l].astype(np.uint32).tofile(os.path.join(out, "col%d.u32" % l))
open(os.path.join(out, "meta.txt"), "w").write("%d %d %d\n" % (n, L, int(z["enterpoint"])))
print("dumped", n, "nodes,", L, "layers to", out)
