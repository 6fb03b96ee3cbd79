"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def make_data(n, dim, seed=1):
    return np.random.default_rng(seed).random((n, dim), dtype=np.float32)


def build_oracle(oracle_mod, V, m, ef, level_seed=7):
    lv = oracle_mod.draw_levels(V.shape[0], m, level_seed)
    idx = oracle_mod.OracleIndex(V.shape[1], m, ef)
    idx.add_batch(V, lv)
    return idx, lv


def brute_force_topk(V, Q, k):
    """exact kNN by squared L2 in float64 -> ids [B][k]"""
    V64, Q64 = V.astype(np.float64), Q.astype(np.float64)
    out = np.empty((Q.shape[0], k), dtype=np.int64)
    vn = (V64 * V64).sum(1)
    for i in range(0, Q.shape[0], 256):
        q = Q64[i:i + 256]
        d = vn[None, :] - 2.0 * q @ V64.T + (q * q).sum(1)[:, None]
        out[i:i + 256] = np.argsort(d, axis=1, kind="stable")[:, :k]
    return out


def recall_at_k(ids, gt):
    k = gt.shape[1]
    hit = 0
    for a, b in zip(ids, gt):
        hit += len(set(int(x) for x in a[:k]) & set(int(x) for x in b))
    return hit / (gt.shape[0] * k)


def graphs_equal(ga, gb):
    """same levels, enterpoint, and per-layer rows in the same stored order"""
    if ga["enterpoint"] != gb["enterpoint"] or ga["max_layer"] != gb["max_layer"]:
        return False, "enterpoint/max_layer %s/%s vs %s/%s" % (ga["enterpoint"], ga["max_layer"], gb["enterpoint"], gb["max_layer"])
    if not np.array_equal(ga["levels"], gb["levels"]):
        return False, "levels differ"
    for l, (ra, rb, ca, cb) in enumerate(zip(ga["row_ptr"], gb["row_ptr"], ga["col"], gb["col"])):
        if not np.array_equal(ra, rb):
            bad = int(np.nonzero(np.diff(ra.astype(np.int64)) != np.diff(rb.astype(np.int64)))[0][0])
            return False, "layer %d: degree of node %d differs" % (l, bad)
        if not np.array_equal(ca, cb):
            bad = int(np.nonzero(ca != cb)[0][0])
            node = int(np.searchsorted(ra, bad, side="right") - 1)
            return False, "layer %d: row of node %d differs" % (l, node)
    return True, ""
