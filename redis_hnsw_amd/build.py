"""Build libhnsw_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One translation unit per kernel family and metric variant (csrc/hnsw_host.hpp lists them), compiled in
parallel into redis_hnsw_amd/build_obj/ and linked into one shared library; each object is rebuilt only when
its own sources change."""
import concurrent.futures
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
OBJ_DIR = os.path.join(_HERE, "build_obj")
LIB_PATH = os.path.join(LIB_DIR, "libhnsw_mi355x.so")
_COMMON = ["hnsw_host.hpp", "hnsw_device.hpp", "hnsw_insert.hpp", "hnsw_occ.hpp", "hnsw_wave_sync.hpp",
           os.path.join("..", "..", "include", "hnsw_mi355x.h")]
# (source, variants, headers besides _COMMON)
UNITS = [
    ("hnsw_engine.hip", [None], ["hnsw_kernels.hpp", "hnsw_search_lean.hpp", "hnsw_search_duo.hpp", "hnsw_plan_lean.hpp", "hnsw_insert_host.inc",
                                 "hnsw_pipeline.inc", "hnsw_transfer.inc", "hnsw_snapshot.inc"]),
    ("hnsw_tu_lean.hip", [0, 1, 2, 3, 4, 5], ["hnsw_search_lean.hpp"]),
    ("hnsw_tu_duo.hip", [0, 1], ["hnsw_search_lean.hpp", "hnsw_search_duo.hpp"]),
    ("hnsw_tu_search.hip", [0, 1, 2, 3, 4, 5], ["hnsw_kernels.hpp"]),
    ("hnsw_tu_insert.hip", [0, 1, 2, 3], []),
    ("hnsw_tu_occ.hip", [0, 1, 2, 3], []),
    ("hnsw_tu_occteam.hip", [0, 1, 2, 3], []),
    ("hnsw_tu_occpar.hip", [0, 1, 2, 3], ["hnsw_occ_par.hpp"]),
    ("hnsw_tu_planlean.hip", [0, 1], ["hnsw_plan_lean.hpp", "hnsw_search_lean.hpp", "hnsw_search_duo.hpp"]),
    ("hnsw_tu_planduo.hip", [0, 1], ["hnsw_plan_lean.hpp", "hnsw_search_lean.hpp", "hnsw_search_duo.hpp"]),
    ("hnsw_tu_std.hip", [None], ["hnsw_std_heap.hpp"]),   # the reference binary's tie order on one lane (tuning tie_mode)
    ("hnsw_group.hip", [None], []),                   # one process, several GPUs: host code above the C ABI
]
SOURCES = [u[0] for u in UNITS]
DEPS = sorted(set(SOURCES + _COMMON + [d for u in UNITS for d in u[2]]))
# -ffp-contract=off: the metric must round exactly where the reference's does
# (explicit fma only, metrics.rs:57); never -ffast-math.
# -fno-slp-vectorize: the SLP vectoriser pairs the scalar adds of the metric's reduction tree into v_pk_add_f32 and
# leaves their DPP operands behind as separate v_mov_dpp (6 instructions per exchange step instead of 4
# v_add_f32_dpp); the packed arithmetic of the distance loop is written out by hand and does not need it.
FLAGS = ["--offload-arch=gfx950", "-Os", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-undefined-func-template"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libhnsw_mi355x.so)")


def _jobs(obj_dir, extra_flags=()):
    """[(object path, command, dependency paths)] for every (source, variant)"""
    out = []
    for src, variants, deps in UNITS:
        for v in variants:
            stem = os.path.splitext(src)[0] + ("" if v is None else "_v%d" % v)
            obj = os.path.join(obj_dir, stem + ".o")
            cmd = [hipcc()] + FLAGS + list(extra_flags) + ([] if v is None else ["-DHNSW_VARIANT=%d" % v]) + \
                  ["-c", os.path.join(CSRC, src), "-o", obj]
            out.append((obj, cmd, [os.path.join(CSRC, d) for d in [src] + _COMMON + deps]))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    if os.path.getmtime(os.path.abspath(__file__)) > t:  # the flags live here
        return True
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    jobs = _jobs(OBJ_DIR)
    if _stale(LIB_PATH, [d for _, _, deps in jobs for d in deps]):
        return True
    return False


def _compile_and_link(lib_path, obj_dir, force=False, extra_flags=(), workers=None):
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    jobs = _jobs(obj_dir, extra_flags)
    todo = [(obj, cmd) for obj, cmd, deps in jobs if force or _stale(obj, deps)]
    workers = workers or max(1, min(len(todo), os.cpu_count() or 1, 8))
    if todo:
        def run(job):
            obj, cmd = job
            p = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
            if p.returncode != 0:
                raise RuntimeError("%s\n%s%s" % (" ".join(cmd), p.stdout[-4000:], p.stderr[-8000:]))
            return obj
        # the heaviest units first (OCC, insert), so the pool drains evenly
        todo.sort(key=lambda j: ("occ" not in j[0], "insert" not in j[0]))   # (occ, occteam first)
        with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(run, todo))
    objs = [obj for obj, _, _ in jobs]
    if todo or force or not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs):
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs, cwd=CSRC)
    return lib_path


def build_library(force=False, extra_flags=()):
    if not force and not extra_flags and not needs_build():
        return LIB_PATH
    return _compile_and_link(LIB_PATH, OBJ_DIR, force=force or bool(extra_flags), extra_flags=extra_flags)


PROF_LIB_PATH = os.path.join(LIB_DIR, "libhnsw_mi355x_prof.so")


def build_profiling_library():
    """Development aid: the same library with per-phase cycle counters compiled in
    (-DHNSW_PHASE_TIMERS); use it with HNSW_MI355X_LIB=<path> scripts/phase_profile.py."""
    return _compile_and_link(PROF_LIB_PATH, OBJ_DIR + "_prof", extra_flags=["-DHNSW_PHASE_TIMERS"])


DEBUG_LIB_PATH = os.path.join(LIB_DIR, "libhnsw_mi355x_dbg.so")


def build_debug_library():
    """Development aid: the library with -DHNSW_OCC_DEBUG -- the delete commit recomputes EVERY re-selection beside the
    validation's verdict on its speculative record and counts the misses (scripts/del_repro.py reads them)."""
    return _compile_and_link(DEBUG_LIB_PATH, OBJ_DIR + "_dbg", extra_flags=["-DHNSW_OCC_DEBUG"])


if __name__ == "__main__":
    print(build_library(force=True))


HOST_TEST_SRC = os.path.join(_HERE, "..", "tests", "cpp", "hnsw_test.cpp")
HOST_TEST_BIN = os.path.join(LIB_DIR, "hnsw_test")


def build_host_test(force=False):
    """g++ build of tests/cpp/hnsw_test.cpp (the reference's core_tests.rs over the
    C++ host mirror), linked against the in-tree library."""
    src = os.path.abspath(HOST_TEST_SRC)
    hdr = os.path.join(_HERE, "host", "hnsw_index.hpp")
    if (not force and os.path.exists(HOST_TEST_BIN)
            and os.path.getmtime(HOST_TEST_BIN) > max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB_PATH))):
        return HOST_TEST_BIN
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-o", HOST_TEST_BIN, src, "-L" + LIB_DIR, "-lhnsw_mi355x",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)
    return HOST_TEST_BIN


SHIM_TEST_SRC = os.path.join(_HERE, "..", "tests", "cpp", "shim_sequence.c")
SHIM_TEST_BIN = os.path.join(LIB_DIR, "shim_sequence")


def build_shim_test(force=False):
    """gcc build of tests/cpp/shim_sequence.c: the Rust shim's call sequence in plain C over
    the C ABI, with a stand-in keyspace for the hnswnodet write-through."""
    src = os.path.abspath(SHIM_TEST_SRC)
    hdr = os.path.join(_HERE, "..", "include", "hnsw_mi355x.h")
    if (not force and os.path.exists(SHIM_TEST_BIN)
            and os.path.getmtime(SHIM_TEST_BIN) > max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB_PATH))):
        return SHIM_TEST_BIN
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-o", SHIM_TEST_BIN, src, "-L" + LIB_DIR, "-lhnsw_mi355x",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lm"]
    subprocess.check_call(cmd)
    return SHIM_TEST_BIN
