"""Golden answers for tests/cpp/shim_sequence.c: the same HNSW.NEW / NODE.ADD / NODE.DEL / SEARCH sequence played
on the CPU oracle (TEST INFRASTRUCTURE; levels drawn from seed 12345 like the engine's generator, core.rs:601-605),
one line per query: `n name:simbits ...`.  The C program compares what the engine answers through the shim's
call sequence with this file.

    python tests/golden/make_shim_golden.py        # rewrites tests/golden/shim_sequence.txt
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
    lines = []
    for _ in range(60):
        q = np.array([rng.frand() for _ in range(DIM)], dtype=np.float32)
        ids, sims = o.search(q, 5)
        lines.append("%d %s" % (len(ids), " ".join("%s:%08x" % (names[int(i)].split(".")[-1], int(s.view(np.uint32)))
                                                    for i, s in zip(ids, sims))))
    out = os.path.join(ROOT, "tests", "golden", "shim_sequence.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote %s (%d queries, %d live nodes)" % (out, len(lines), o.live_count))


if __name__ == "__main__":
    main()
