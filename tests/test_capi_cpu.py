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


def test_ctypes_structs_have_the_header_s_layout(capi, tmp_path):
    """the Python binding's Structures against the C compiler's view of include/hnsw_mi355x.h: sizes and the offset
    of every field (a silent mismatch would corrupt hnsw_get_info / replica / pipeline calls)"""
    import ctypes as C
    import subprocess
    structs = {"hnsw_info": capi.Info, "hnsw_counters": capi.Counters, "hnsw_replica": capi.Replica, "hnsw_pipeline": capi.Pipeline}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hnsw_mi355x.h"', 'int main(void) {']
    for cname, st in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(st, fname).offset, (cname, fname)


def test_insert_units_synchronise_their_wave_through_hnsw_wave_sync():
    """The insert / delete kernels hand data between the lanes of a wave through HBM as well as through LDS; their
    "__syncthreads()" must therefore wait for the wave's own stores.  Every translation unit that holds
    such kernels includes hnsw_wave_sync.hpp before anything else, the header spells the wait out, and no unit brings
    a weaker definition of its own."""
    import os, re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "redis_hnsw_amd", "csrc")
    hdr = open(os.path.join(csrc, "hnsw_wave_sync.hpp")).read()
    assert re.search(r's_waitcnt vmcnt\(0\) lgkmcnt\(0\)"\s*:::\s*"memory"', hdr)
    assert "#define __syncthreads() ::hnsw::wave_sync_full()" in hdr
    units = ["hnsw_tu_insert.hip", "hnsw_tu_occ.hip", "hnsw_tu_planlean.hip", "hnsw_tu_planduo.hip", "hnsw_tu_occteam.hip"]
    for u in units:
        src = open(os.path.join(csrc, u)).read()
        incs = re.findall(r'^#include\s+[<"]([^>"]+)[>"]', src, flags=re.M)
        assert incs and incs[0] == "hnsw_wave_sync.hpp", (u, incs[:2])
        assert "#define __syncthreads" not in src, u
    # and the kernels of those units live in headers that no other unit instantiates with the plain barrier
    for f in os.listdir(csrc):
        if f.endswith(".hip") and f not in units:
            src = open(os.path.join(csrc, f)).read()
            for k in ("k_occ_commit<", "k_occ_del_commit<", "k_insert_commit_exact<", "k_delete_exact<", "k_occ_shrinks<", "k_insert_plan<"):
                assert k not in src, (f, k)
