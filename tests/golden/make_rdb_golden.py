#!/usr/bin/env python3
"""Generates tests/golden/rdb/*.npz: the reference's two persisted value types (hnswindex, hnswnodet;
src/types.rs:243-284, 410-428) for the graph of a committed golden case -- i.e. what the module's RDB save
callbacks would stream for an index the ORACLE built (adds, deletes, more adds).  The fixture pins the field
order and the name/level/neighbour conventions of redis_hnsw_amd/rdb.py.
    python tests/golden/make_rdb_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from redis_hnsw_amd import rdb  # noqa: E402  (pure Python: no GPU, no library)
from tests.golden_util import load_golden  # noqa: E402

CASES = ["u360_dim32_m5_ef16_del", "line100_dim4_m5_ef16"]


def node_key(index, i):
    return "%s.%s.node%d" % (rdb.PREFIX, index, i)        # src/lib.rs:342-343


def build(case):
    c = load_golden(case)
    g = dict(c["graph"])
    g["vectors"] = c["V"]
    n = c["n"]
    dead = np.zeros(n, dtype=bool)
    dead[c["deleted"]] = True
    names = [node_key("gold", i) for i in range(n)]
    ir, nodes = rdb.graph_to_redis("%s.gold" % rdb.PREFIX, c["dim"], c["m"], c["ef"], g, names, dead)
    keys = sorted(nodes, key=lambda s: int(s.rsplit("node", 1)[1]))
    blobs = [rdb.save_node(nodes[k]) for k in keys]
    out = dict(case=np.array(case), index_value=np.frombuffer(rdb.save_index(ir), dtype=np.uint8),
               node_keys=np.array(keys), node_values=np.frombuffer(b"".join(blobs), dtype=np.uint8),
               node_value_len=np.array([len(b) for b in blobs], dtype=np.int64))
    path = os.path.join(HERE, "rdb", case + ".npz")
    np.savez_compressed(path, **out)
    print(case, "live nodes", len(keys), "index value", out["index_value"].size, "B, node values", out["node_values"].size, "B ->", os.path.getsize(path), "B")


if __name__ == "__main__":
    for case in CASES:
        build(case)
