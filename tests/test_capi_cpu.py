"""CPU-side checks of the boundary: the library builds for gfx950, loads, and
exports every symbol include/hnsw_mi355x.h declares.  No compute calls."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from redis_hnsw_amd import _capi, build
    build.build_library()
    return _capi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hnsw_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(capi):
    lib = capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "libhnsw_mi355x.so does not export %s" % name
    # and the Python binding covers exactly the header
    assert sorted(capi.SIGNATURES) == declared


def test_no_gpu_is_a_loud_error_not_a_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from redis_hnsw_amd import HNSWError, Index
    with pytest.raises(HNSWError) as e:
        Index("foo", 4, 5, 16)
    assert e.value.status == capi.ERR_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "redis_hnsw_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".cpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "hnsw_oracle", "oracle/"):
                    assert needle not in src, "%s references the oracle (%s)" % (f, needle)


def test_cpp_host_mirror_builds(capi):
    from redis_hnsw_amd import build
    exe = build.build_host_test()
    assert os.path.exists(exe)


def test_error_string_is_the_debug_rendering():
    """core.rs:42-46: clients see format!("{:?}", HNSWError::String(msg))"""
    from redis_hnsw_amd import HNSWError
    assert HNSWError("data dimension: 3 does not match Index").error_string() == \
        'String("data dimension: 3 does not match Index")'
    assert HNSWError('Node: "a\\b" already exists').error_string() == 'String("Node: \\"a\\\\b\\" already exists")'
