import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    """goldens written by the C oracle itself (make_golden.py): they pin stability"""
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                  if not n.startswith(("transcribed_", "tiecase_")))


def transcribed_cases(kind="exact"):
    """goldens written by the Python transcription of core.rs (tests/transcription/): they pin FIDELITY -- nothing the
    oracle or the engine computed went into them.
      kind "exact": the files the oracle and the engine must reproduce bit for bit -- runs in the reference's own tie
                    order (std's BinaryHeap, restated) in which no decision met a tie, and runs in the (sim, id) total
                    order the oracle and the engine use;
      kind "tied":  runs in the reference's tie order in which decisions DID meet ties: what the Rust binary builds
                    where the total order chooses differently (the divergence is measured, not asserted away)."""
    import json
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "transcribed_*.npz"))):
        st = json.loads(bytes(np.load(p)["stats"]).decode())
        decided_by_tie = st["ties_total"]["accept_657"] + st["ties_total"]["select_733"] > 0
        tied = st.get("ties", "fifo") != "total" and decided_by_tie
        if (kind == "tied") == tied:
            out.append(os.path.splitext(os.path.basename(p))[0])
    return out


def load_transcribed(name):
    """-> the case with its inputs regenerated from SURVEY 8d's seeds (vectors default_rng(1), queries default_rng(2),
    levels floor(-ln U / ln M) from default_rng(7), node 0 at level 0), as make_transcribed_golden.py drew them"""
    import json
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n, dim, m, ef, k, nq = [int(x) for x in z["params"]]
    V = np.random.default_rng(1).random((n, dim), dtype=np.float32)
    Q = np.random.default_rng(2).random((nq, dim), dtype=np.float32)
    u = np.maximum(np.random.default_rng(7).random(n), np.finfo(np.float64).tiny)
    lv = np.floor(-np.log(u) * (1.0 / np.log(float(m)))).astype(np.int64)
    lv[0] = 0
    lv = np.minimum(lv, 31).astype(np.int32)
    L = int(z["max_layer"]) + 1
    g = dict(levels=lv.astype(np.uint32), enterpoint=int(z["enterpoint"]), max_layer=int(z["max_layer"]),
             row_ptr=[z["row_ptr_%d" % l] for l in range(L)], col=[z["col_%d" % l] for l in range(L)])
    return dict(n=n, dim=dim, m=m, ef=ef, k=k, V=V, Q=Q, levels=lv, graph=g, ids=z["ids"], sims_bits=z["sims_bits"],
                n_out=z["n_out"], search_counters=z["search_counters"], insert_counters=z["insert_counters"],
                stats=json.loads(bytes(z["stats"]).decode()))


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


def load_tiecase(name="tiecase_rust_lattice"):
    """tests/transcription/make_rust_tie_golden.py: a graph built AND searched by the transcription in its "rust" tie
    mode (std's BinaryHeap restated) on distinct points of the lattice {0,1,2}^8 -- equal similarities everywhere"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n, dim, m, ef, k, nq = [int(x) for x in z["params"]]

    def lattice(cnt, seed):
        codes = np.random.default_rng(seed).choice(3 ** dim, size=cnt, replace=False)
        return np.stack([(codes // 3 ** j) % 3 for j in range(dim)], axis=1).astype(np.float32)
    L = int(z["max_layer"]) + 1
    g = dict(levels=z["levels"].astype(np.uint32), enterpoint=int(z["enterpoint"]), max_layer=int(z["max_layer"]),
             row_ptr=[z["row_ptr_%d" % l] for l in range(L)], col=[z["col_%d" % l] for l in range(L)], vectors=lattice(n, 1))
    return dict(n=n, dim=dim, m=m, ef=ef, k=k, Q=lattice(nq, 2), graph=g, ids=z["ids"], sim_bits=z["sim_bits"], n_out=z["n_out"],
                accept_ties=int(z["accept_ties"]))


def load_tiecase_del(name="tiecase_rust_lattice_del"):
    """tests/transcription/make_rust_tie_del_golden.py: the transcription in its "rust" tie mode builds n0 nodes on lattice
    data, deletes `victims` in that order, adds n1 more; the graph after all that"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n0, n1, dim, m, ef = [int(x) for x in z["params"]]
    L = int(z["max_layer"]) + 1
    g = dict(levels=z["levels"].astype(np.uint32), enterpoint=int(z["enterpoint"]), max_layer=int(z["max_layer"]),
             row_ptr=[z["row_ptr_%d" % l] for l in range(L)], col=[z["col_%d" % l] for l in range(L)])
    return dict(n0=n0, n1=n1, dim=dim, m=m, ef=ef, V=z["V"], levels=z["levels"], victims=z["victims"], graph=g)
