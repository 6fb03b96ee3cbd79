"""redis_hnsw_amd -- MI355X-native engine for the redis_hnsw hot path.

The product is the C-ABI library (include/hnsw_mi355x.h, csrc/); this package
is the Python host-side mirror of the reference's Index interface
(src/hnsw/core.rs: Index::new / add_node / search_knn) used by the tests and
bench.py.  Every operation is a HIP kernel launch; nothing computes on the CPU.
"""
from .index import HNSWError, Index, SearchResult  # noqa: F401
