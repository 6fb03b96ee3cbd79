#!/usr/bin/env python3
"""FETCH_SIZE calibration for the engine's own access pattern (8 lanes x 16 B per
128-B block of a 512-B row): streams a known byte count through k_metric_pairs_reg.
Run under `rocprofv3 --pmc FETCH_SIZE`.  Known bytes = 2 * n * 512."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from redis_hnsw_amd import index as eng
n = 2_000_000
rng = np.random.default_rng(0)
a = rng.random((n, 128), dtype=np.float32)
b = rng.random((n, 128), dtype=np.float32)
out = eng.metric_pairs(a, b)
print("pairs", n, "known_read_bytes", 2 * n * 512, "checksum", float(out[:1000].sum()))
