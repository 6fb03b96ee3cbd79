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


def test_insert_units_wait_for_their_stores_at_every_wave_barrier(tmp_path):
    """The insert / delete kernels hand data between the lanes of a wave through HBM as well as through LDS (rows, journal,
    read log); their synchronisation points are hnsw::wave_sync() -- s_waitcnt vmcnt(0) lgkmcnt(0) + wave barrier (DESIGN
    4.2e, hnsw_wave_sync.hpp).  Checked where it counts, in the ISA of an insert unit (hipcc -S of hnsw_tu_occ.hip, the
    kernels of the windowed insert and the speculative delete): EVERY wave barrier of the unit is preceded by the full
    wait, or is a declared LDS-only ordering point (hnsw::lds_order(), marked in the ISA); a one-wave unit holds no
    s_barrier at all.  And the units say which synchronisation the shared code of hnsw_device.hpp gets, by name: nothing
    redefines __syncthreads(), nothing depends on include order."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "redis_hnsw_amd", "csrc")
    hdr = open(os.path.join(csrc, "hnsw_wave_sync.hpp")).read()
    assert re.search(r's_waitcnt vmcnt\(0\) lgkmcnt\(0\)"\s*:::\s*"memory"', hdr)
    full = ["hnsw_tu_insert.hip", "hnsw_tu_occ.hip", "hnsw_tu_occteam.hip", "hnsw_tu_occpar.hip", "hnsw_tu_planlean.hip", "hnsw_tu_planduo.hip"]
    block = ["hnsw_engine.hip", "hnsw_tu_search.hip", "hnsw_tu_lean.hip", "hnsw_tu_duo.hip"]
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read()
        assert "#define __syncthreads" not in src, f
        if f in full:
            assert re.search(r"^#define HNSW_SYNC_WAVE_FULL\b", src, flags=re.M) and "HNSW_SYNC_BLOCK" not in src, f
        elif f in block:
            assert re.search(r"^#define HNSW_SYNC_BLOCK\b", src, flags=re.M) and "#define HNSW_SYNC_WAVE_FULL" not in src, f
        # the insert / delete kernels are instantiated in the units that wait for their stores, nowhere else
        if f.endswith(".hip") and f not in full:
            for k in ("k_occ_commit<", "k_occ_commit_par<", "k_occ_del_commit<", "k_insert_commit_exact<", "k_delete_exact<", "k_occ_shrinks<",
                      "k_insert_plan<"):
                assert k not in src, (f, k)
    # the shared insert code names its synchronisation
    for f in ("hnsw_insert.hpp", "hnsw_occ.hpp", "hnsw_occ_par.hpp", "hnsw_plan_lean.hpp"):
        src = open(os.path.join(csrc, f)).read()
        assert "__syncthreads()" not in src and "wave_sync" in src, f
        assert "__builtin_amdgcn_wave_barrier" not in src, f
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = tmp_path / "occ.s"
    from redis_hnsw_amd import build as b
    subprocess.check_call([hipcc] + [x for x in b.FLAGS if x != "-fPIC"] + ["-DHNSW_VARIANT=1", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                                                                               "-o", str(asm), os.path.join(csrc, "hnsw_tu_occ.hip")], cwd=csrc)
    lines = asm.read_text().split("\n")
    n_full = n_lds = 0
    for i, line in enumerate(lines):
        assert not re.match(r"\s*s_barrier\b", line), "a one-wave insert unit has no workgroup barrier (line %d)" % i
        if "; wave barrier" not in line:
            continue
        kind, j = None, i - 1
        while j >= 0 and kind is None:                  # the scheduler may put ALU instructions (spill reloads ...) in between,
            t = lines[j].strip()                        # never a memory instruction: the first one met ends the search
            if "s_waitcnt vmcnt(0) lgkmcnt(0)" in t:
                kind = "full"
            elif "; hnsw lds_order" in t:
                kind = "lds"
            elif re.match(r"(global_|flat_|buffer_|scratch_|s_load|s_buffer_load|s_endpgm|s_setpc|s_swappc)", t) or i - j > 200:
                break
            j -= 1
        assert kind is not None, "wave barrier without the full wait or an lds_order mark:\n" + "\n".join(lines[max(0, i - 10):i + 1])
        n_full += kind == "full"
        n_lds += kind == "lds"
    assert n_full > 500 and n_lds > 0, (n_full, n_lds)
