"""Build libhnsw_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhnsw_mi355x.so")
SOURCES = ["hnsw_engine.hip"]
DEPS = ["hnsw_engine.hip", "hnsw_device.hpp", "hnsw_kernels.hpp", "hnsw_insert.hpp", "hnsw_insert_host.inc", "hnsw_search_lean.hpp", "hnsw_occ.hpp",
        os.path.join("..", "..", "include", "hnsw_mi355x.h")]
# -ffp-contract=off: the metric must round exactly where the reference's does
# (explicit fma only, metrics.rs:57); never -ffast-math.
FLAGS = ["--offload-arch=gfx950", "-Os", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libhnsw_mi355x.so)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    if os.path.getmtime(os.path.abspath(__file__)) > t:  # the flags live here
        return True
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_library(force=False, extra_flags=()):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


PROF_LIB_PATH = os.path.join(LIB_DIR, "libhnsw_mi355x_prof.so")


def build_profiling_library():
    """Development aid: the same library with per-phase cycle counters compiled in
    (-DHNSW_PHASE_TIMERS); use it with HNSW_MI355X_LIB=<path> scripts/phase_profile.py."""
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + ["-DHNSW_PHASE_TIMERS", "-o", PROF_LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    subprocess.check_call(cmd, cwd=CSRC)
    return PROF_LIB_PATH


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
