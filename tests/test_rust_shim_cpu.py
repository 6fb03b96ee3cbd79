"""The Rust shim is committed as files (integration/rust/) but cannot be compiled in this image (no cargo):
what can be checked without a compiler is checked here -- every `extern "C"` declaration of ffi.rs against
include/hnsw_mi355x.h (name, arity, argument and return types), the #[repr(C)] hnsw_info against the header's
struct, the status constants, and that the lib.rs patch is well formed.  The shim's call sequence itself runs in C
under -m gpu (tests/cpp/shim_sequence.c, against the oracle's golden answers)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "integration", "rust")

C2RUST = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "int": "c_int", "float": "f32",
          "uint8_t": "u8", "void": "c_void", "char": "c_char", "hnsw_status": "c_int"}


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _c_type_to_rust(ctype):
    """'const uint64_t *const *' -> '*const *const u64'"""
    t = ctype.strip()
    stars = []
    while t.endswith("*") or t.endswith("*const"):
        if t.endswith("*const"):
            t = t[: -len("*const")].strip()
            stars.append("const_ptr_level")     # a const pointer: constness of what it points to is decided below
        else:
            t = t[:-1].strip()
            stars.append("ptr")
    const_base = t.startswith("const ")
    base = t.replace("const ", "").strip()
    rust = C2RUST.get(base, base)
    # innermost pointer takes the base's constness; outer levels are '*const' iff that level was declared '*const'
    out = rust
    for i, lvl in enumerate(stars[::-1]):
        if i == 0:
            out = ("*const " if const_base else "*mut ") + out
        else:
            inner_is_const_ptr = stars[::-1][i - 1] == "const_ptr_level"
            out = ("*const " if inner_is_const_ptr else "*mut ") + out
    return out


def _header_functions():
    text = _strip_comments(open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read())
    fns = {}
    for m in re.finditer(r"\b(hnsw_status|void|const char \*|uint32_t|hnsw_index \*)\s*(hnsw_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            mm = re.match(r"(.*?)([A-Za-z_][A-Za-z_0-9]*)$", a)          # type, then the parameter name
            params.append(_c_type_to_rust(mm.group(1)))
        fns[name] = (params, {"hnsw_status": "c_int", "void": None, "const char *": "*const c_char", "uint32_t": "u32",
                                "hnsw_index *": "*mut hnsw_index"}[ret])
    return fns


def _rust_functions():
    text = re.sub(r"//.*", "", open(os.path.join(RUST, "src", "hnsw", "ffi.rs")).read())
    fns = {}
    for m in re.finditer(r"pub fn (hnsw_[a-z_0-9]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", text, flags=re.S):
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in m.group(2).split(",") if a.strip()]
        fns[m.group(1)] = (params, m.group(3).strip() if m.group(3) else None)
    return fns


def test_every_rust_extern_matches_the_header():
    c, r = _header_functions(), _rust_functions()
    assert len(r) >= 15
    for name, (rparams, rret) in r.items():
        assert name in c, "ffi.rs declares %s, which include/hnsw_mi355x.h does not" % name
        cparams, cret = c[name]
        assert rparams == cparams, "%s: ffi.rs %r vs header %r" % (name, rparams, cparams)
        assert rret == cret, "%s: return type %r vs %r" % (name, rret, cret)
    # everything the module's call sites need is there
    for need in ("hnsw_create", "hnsw_destroy", "hnsw_last_error", "hnsw_add", "hnsw_delete", "hnsw_search",
                 "hnsw_search_batch", "hnsw_import", "hnsw_get_info", "hnsw_get_levels", "hnsw_get_vector", "hnsw_get_neighbors"):
        assert need in r


def test_repr_c_info_struct_and_status_codes_match_the_header():
    htext = _strip_comments(open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read())
    body = re.search(r"typedef struct \{([^}]*)\} hnsw_info;", htext, flags=re.S).group(1)
    cfields = []
    for decl in [d.strip() for d in body.split(";") if d.strip()]:
        ctype, names = decl.split(None, 1)
        cfields += [(n.strip(), C2RUST[ctype]) for n in names.split(",")]
    rtext = open(os.path.join(RUST, "src", "hnsw", "ffi.rs")).read()
    rbody = re.search(r"pub struct hnsw_info \{([^}]*)\}", rtext, flags=re.S).group(1)
    rfields = [(m.group(1), m.group(2)) for m in re.finditer(r"pub ([a-z_0-9]+): ([a-z0-9]+),", rbody)]
    assert rfields == cfields
    enum = re.search(r"typedef enum \{([^}]*)\} hnsw_status;", htext, flags=re.S).group(1)
    for m in re.finditer(r"(HNSW_[A-Z_]+)\s*=\s*(\d+)", enum):
        assert re.search(r"pub const %s: c_int = %s;" % (m.group(1), m.group(2)), rtext), m.group(1)


def test_the_shim_uses_only_declared_entry_points_and_the_patch_is_well_formed():
    r = _rust_functions()
    gpu = open(os.path.join(RUST, "src", "hnsw", "gpu_index.rs")).read()
    used = set(re.findall(r"ffi::(hnsw_[a-z_0-9]+)\s*\(", gpu))
    assert used and used <= set(r), used - set(r)
    # the reference's three error strings, verbatim (core.rs:390, 408, 421, 479)
    for msg in ('"data dimension: {} does not match Index"', '"Node: {:?} already exists"', '"Node: {:?} does not exist"'):
        assert msg in gpu
    diff = open(os.path.join(RUST, "lib_rs.diff")).read().split("\n")
    assert diff[0] == "--- a/src/lib.rs" and diff[1] == "+++ b/src/lib.rs"
    hunks = [l for l in diff if l.startswith("@@")]
    assert len(hunks) >= 8
    starts = [int(re.match(r"@@ -(\d+)", h).group(1)) for h in hunks]
    assert starts == sorted(starts)                                   # applies top to bottom
    for fn in ("GpuIndex::new(", "GpuIndex::from_keys(", "index_redis_of(", "node_redis_of(", "index.sync_redis(ir)"):
        assert any(l.startswith("+") and fn in l for l in diff), fn
    for f in ("build.rs", "Cargo.toml.diff", "README.md"):
        assert os.path.exists(os.path.join(RUST, f))


def _rust_fn_body(text, signature_start):
    """the brace-balanced body of the fn whose text starts with `signature_start`"""
    a = text.index(signature_start)
    # the first '{' after the parameter list closes is the body
    depth_par, j = 0, a
    while True:
        c = text[j]
        if c == "(":
            depth_par += 1
        elif c == ")":
            depth_par -= 1
            if depth_par == 0:
                break
        j += 1
    i = text.index("{", j)
    depth, k = 0, i
    while True:
        if text[k] == "{":
            depth += 1
        elif text[k] == "}":
            depth -= 1
            if depth == 0:
                return text[i:k + 1]
        k += 1


def _no_comments(body):
    return re.sub(r"//.*", "", body)


def test_the_shim_keeps_the_reference_s_layer_sets_and_reload_semantics():
    """What a reference-written keyspace needs from the shim, checked on the Rust text (no cargo here; the same logic
    runs in C under -m gpu, tests/cpp/shim_sequence.c):
      * layers(): a node goes into the set of its TOP layer only (core.rs:596, src/types.rs:73-82) -- round 3's
        text pushed it into every set 0..=l;
      * from_keys(): levels come from IndexRedis.layers and the layer count from IndexRedis.max_layer
        (src/lib.rs:287-299), never from a node's row count (a promoted node is saved with fewer rows than
        level + 1, core.rs:523, 587-593);
      * update_index is O(1): sync_redis edits the stored value, it does not rebuild it."""
    gpu = open(os.path.join(RUST, "src", "hnsw", "gpu_index.rs")).read()
    layers = _no_comments(_rust_fn_body(gpu, "pub fn layers(&self)"))
    assert "out[self.levels[id] as usize].push(n.clone())" in layers
    assert "take(" not in layers and "0..=" not in layers and "for layer in" not in layers
    fk = _no_comments(_rust_fn_body(gpu, "pub fn from_keys("))
    sig = gpu[gpu.index("pub fn from_keys("):gpu.index("-> Result<Self, HNSWError>", gpu.index("pub fn from_keys("))]
    assert "layers: &[Vec<String>]" in sig and "max_layer: usize" in sig
    assert "nbrs.len()" not in fk and ".len().max(1)" not in fk          # round 3: levels[i] = nbrs.len().max(1) - 1
    assert re.search(r"for \(l, set\) in layers\.iter\(\)\.enumerate\(\)", fk) and "levels[i] = l as u32" in fk
    assert "is in no layer set" in fk
    assert re.search(r"let n_layers = \(max_layer \+ 1\)", fk)
    assert "ffi::hnsw_import(" in fk and "levels.as_ptr()" in fk
    # the per-command path is O(1): one level looked up, one name appended / swap-removed
    add = _no_comments(_rust_fn_body(gpu, "pub fn add_node(&mut self"))
    assert "ffi::hnsw_get_level(" in add and "hnsw_get_levels" not in add
    assert add.index("self.names.push(") < add.index("does not fit the buffer")   # bookkeeping before any late error
    sync = _no_comments(_rust_fn_body(gpu, "pub fn sync_redis(&mut self"))
    assert "swap_remove" in sync and "ir.nodes.push(" in sync and ".collect()" not in sync and "for (id, name)" not in sync
    assert "ir.layers[l].push(name)" in sync and "ir.layers.truncate(" in sync
    # the lib.rs patch hands the layer sets through and no longer re-serialises the index per command
    diff = open(os.path.join(RUST, "lib_rs.diff")).read()
    assert "&ir.layers, ir.max_layer" in diff
    upd = diff[diff.index("+fn update_index("):]
    upd = upd[:upd.index("@@")]
    assert "index.sync_redis(ir)" in upd and "+            key.set_value" not in upd
    assert diff.count("update_index(ctx, &index_name, &mut index)?;") == 2
    # readers share one engine handle: the FFI search is serialised inside GpuIndex (the header's one-caller rule)
    search = _no_comments(_rust_fn_body(gpu, "pub fn search_knn(&self"))
    assert "self.search_lock.lock()" in search and search.index("search_lock.lock()") < search.index("ffi::hnsw_search(")
