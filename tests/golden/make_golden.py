#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference (Rust) cannot be built or run in this image, so these vectors are
produced by the CPU oracle (oracle/hnsw_oracle.c), which is itself pinned to the
reference's known-answer tests (tests/test_oracle_kat.py).  They freeze the
expected behaviour of the whole path -- insert order -> adjacency rows, search ->
ids / similarity bits / work counters -- so that a change to either the oracle or
the engine that alters results is caught even where the two would still agree
with each other.  Re-run only when the reference semantics are re-read:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle  # noqa: E402

CASES = {
    # name: (n, dim, m, ef, k, nq, data_seed, level_seed)
    "line100_dim4_m5_ef16": None,                       # core_tests.rs:21-53 (levels fixed by seed 3)
    "u600_dim32_m5_ef16": (600, 32, 5, 16, 5, 32, 1, 7),
    "u500_dim128_m16_ef64": (500, 128, 16, 64, 10, 32, 1, 7),
    "u400_dim12_m6_ef24": (400, 12, 6, 24, 8, 32, 5, 9),   # scalar metric order
    "u360_dim32_m5_ef16_del": (360, 32, 5, 16, 5, 32, 11, 13),   # 300 adds, 100 deletes, 60 more adds
}
DELETES = {"u360_dim32_m5_ef16_del": (300, 100, 17)}   # name: (adds before, deletes, order seed)


def build(name, spec):
    if spec is None:
        n, dim, m, ef, k = 100, 4, 5, 16, 5
        V = np.repeat(np.arange(n, dtype=np.float32)[:, None], dim, axis=1)
        Q = np.full((1, dim), 10.0, np.float32)
        lv = oracle.draw_levels(n, m, 3)
    else:
        n, dim, m, ef, k, nq, ds, ls = spec
        V = np.random.default_rng(ds).random((n, dim), dtype=np.float32)
        Q = np.random.default_rng(ds + 100).random((nq, dim), dtype=np.float32)
        lv = oracle.draw_levels(n, m, ls)
    o = oracle.OracleIndex(dim, m, ef)
    n_first = n if name not in DELETES else DELETES[name][0]
    o.add_batch(V[:n_first], lv[:n_first])
    deleted = np.zeros(0, dtype=np.int64)
    if name in DELETES:       # HNSW.NODE.DEL (enterpoint first), then the remaining HNSW.NODE.ADDs
        _, n_del, seed = DELETES[name]
        order = np.random.default_rng(seed).permutation(n_first)[:n_del]
        ep0 = o.enterpoint
        deleted = np.concatenate([[ep0], order[order != ep0]]).astype(np.int64)
        for i in deleted:
            o.delete(int(i))
        o.add_batch(V[n_first:], lv[n_first:])
    g = o.export()
    ids, sims, n_out, ct = o.search_batch(Q, k)
    ic = o.insert_counters()
    out = dict(params=np.array([n, dim, m, ef, k], dtype=np.int64), vectors=V, queries=Q, levels=lv.astype(np.int32),
               n_first=np.int64(n_first), deleted=deleted,
               enterpoint=np.int64(g["enterpoint"]), max_layer=np.int64(g["max_layer"]),
               ids=ids, sims_bits=sims.view(np.uint32), n_out=n_out,
               search_counters=np.array([ct.n_dist, ct.n_ids, ct.n_expand], dtype=np.int64),
               insert_counters=np.array([ic.n_dist, ic.n_ids, ic.n_expand], dtype=np.int64))
    for l, (rp, cl) in enumerate(zip(g["row_ptr"], g["col"])):
        out["row_ptr_%d" % l] = rp
        out["col_%d" % l] = cl
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "nodes", n, "layers", g["max_layer"] + 1, "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    for name, spec in CASES.items():
        build(name, spec)
