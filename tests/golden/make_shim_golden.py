"""Golden answers for tests/cpp/shim_sequence.c: the same HNSW.NEW / NODE.ADD / NODE.DEL / SEARCH sequence played
on the CPU oracle (TEST INFRASTRUCTURE; levels drawn from seed 12345 like the engine's generator, core.rs:601-605),
one line per query: `n name:simbits ...`.  The C program compares what the engine answers through the shim's
call sequence with this file.

Second file, shim_reload.txt: the C program then RELOADS its index from a reference-shaped keyspace (a promoted node
saved with its pre-promotion rows only, layer sets top-only) and plays three more multi-level adds and a delete on the
reloaded index; the oracle plays the same commands WITHOUT a restart (the reference's reload is an identity on the
graph, src/lib.rs:252-315), and this file holds every adjacency row it ends with (`R name layer n nbr...`, stored
order) and 20 more answers.

    python tests/golden/make_shim_golden.py        # rewrites tests/golden/shim_sequence.txt and shim_reload.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

DIM, M, EFC, N = 16, 4, 24, 400


class Lcg:
    def __init__(self, s):
        self.s = s

    def frand(self):
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return np.float32(self.s >> 8) / np.float32(16777216.0)


def main():
    oracle.build()
    o = oracle.OracleIndex(DIM, M, EFC, seed=12345)
    rng = Lcg(7)
    names = []
    for i in range(N):
        v = np.array([rng.frand() for _ in range(DIM)], dtype=np.float32)
        assert o.add(v, -1) == i
        names.append("hnsw.idx.n%d" % i)
        if i % 9 == 8:
            o.delete(i - 5)
    # the last command before the restart: a node above the current top layer (core.rs:587-593)
    old_top = o.max_layer
    v = np.array([rng.frand() for _ in range(DIM)], dtype=np.float32)
    assert o.add(v, old_top + 2) == N
    names.append("hnsw.idx.n%d" % N)
    assert o.max_layer == old_top + 2 and o.enterpoint == N
    lines = []
    for _ in range(60):
        q = np.array([rng.frand() for _ in range(DIM)], dtype=np.float32)
        ids, sims = o.search(q, 5)
        lines.append("%d %s" % (len(ids), " ".join("%s:%08x" % (names[int(i)].split(".")[-1], int(s.view(np.uint32)))
                                                    for i, s in zip(ids, sims))))
    out = os.path.join(ROOT, "tests", "golden", "shim_sequence.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote %s (%d queries, %d live nodes)" % (out, len(lines), o.live_count))
    # ---- after the restart: three multi-level adds (the second one a promotion) and a delete
    for a, lv in enumerate((1, old_top + 3, 2)):
        v = np.array([rng.frand() for _ in range(DIM)], dtype=np.float32)
        assert o.add(v, lv) == N + 1 + a
        names.append("hnsw.idx.n%d" % (N + 1 + a))
    o.delete(17)
    assert o.max_layer == old_top + 3 and o.enterpoint == N + 2
    short = [nm.split(".")[-1] for nm in names]
    rows = []
    for i in range(len(names)):
        if not o.is_live(i):
            continue
        for l in range(o.level(i) + 1):
            nb = [short[int(j)] for j in o.neighbors(i, l)]
            rows.append("R %s %d %d%s" % (short[i], l, len(nb), "".join(" " + x for x in nb)))
    lines2 = ["ROWS %d" % len(rows)] + rows
    for _ in range(20):
        q = np.array([rng.frand() for _ in range(DIM)], dtype=np.float32)
        ids, sims = o.search(q, 5)
        lines2.append("%d %s" % (len(ids), " ".join("%s:%08x" % (short[int(i)], int(s.view(np.uint32)))
                                                     for i, s in zip(ids, sims))))
    out2 = os.path.join(ROOT, "tests", "golden", "shim_reload.txt")
    open(out2, "w").write("\n".join(lines2) + "\n")
    print("wrote %s (%d rows, 20 queries, %d live nodes, max_layer %d)" % (out2, len(rows), o.live_count, o.max_layer))


if __name__ == "__main__":
    main()
