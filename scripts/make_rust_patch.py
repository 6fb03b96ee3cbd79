"""Regenerate integration/rust/lib_rs.diff: the call-site changes of the reference's src/lib.rs that put
GpuIndex (integration/rust/src/hnsw/gpu_index.rs) behind the HNSW.* commands.  Needs the reference tree
(/root/reference, this container only); the edits are applied to a scratch copy and `diff -U2` is what is kept."""
import os
import subprocess
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(REF, "src", "lib.rs")).read()


def sub(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, count)


def cut(start, end, new):
    """replace everything from `start` up to (not including) `end`"""
    global src
    a = src.index(start)
    b = src.index(end, a)
    src = src[:a] + new + src[b:]


sub("use hnsw::{Index, Node};", "use hnsw::gpu_index::{GpuIndex, NodeView};")
sub("type IndexT = Index<f32, f32>;",
    "type IndexT = GpuIndex; // the MI355X engine behind Index<f32,f32>'s methods (src/hnsw/gpu_index.rs)")
# HNSW.NEW
cut("            let index = Index::new(", "            // Add index to global hashmap",
    "            let index = GpuIndex::new(&index_name, data_dim, m, ef_construction)\n"
    "                .map_err(|e| e.error_string())?;\n"
    "            key.set_value::<IndexRedis>(&HNSW_INDEX_REDIS_TYPE, index_redis_of(&index))?;\n")
# HNSW.GET
cut("    ctx.log_debug(format!(\"Index: {:?}\", index).as_str());", "    Ok(index_redis.into())",
    "    let index_redis: IndexRedis = index_redis_of(&index);\n\n")
# HNSW.DEL
sub("    for (node_name, _) in index.nodes.iter() {", "    for node_name in index.node_names() {")
# make_index: one upload instead of rebuilding Node / NodeWeak objects key by key
cut("fn make_index(", "fn update_index(",
    "fn make_index(ctx: &Context, ir: &IndexRedis) -> Result<IndexT, RedisError> {\n"
    "    // read every hnswnodet key once, then ONE upload: the graph goes straight into HBM (hnsw_import)\n"
    "    let mut nodes = Vec::with_capacity(ir.node_count);\n"
    "    for node_name in &ir.nodes {\n"
    "        let key = ctx.open_key(&node_name);\n"
    "        let nr = key\n"
    "            .get_value::<NodeRedis>(&HNSW_NODE_REDIS_TYPE)?\n"
    "            .ok_or_else(|| format!(\"Node: {} does not exist\", node_name))?;\n"
    "        nodes.push((node_name.to_owned(), nr.data.clone(), nr.neighbors.clone()));\n"
    "    }\n"
    "    // levels come from the layer sets and max_layer, as in the loop this replaces (src/lib.rs:287-299)\n"
    "    GpuIndex::from_keys(&ir.name, ir.data_dim, ir.m, ir.ef_construction, nodes, &ir.layers, ir.max_layer,\n"
    "                        ir.enterpoint.clone())\n"
    "        .map_err(|e| RedisError::String(e.error_string()))\n"
    "}\n\n")
# update_index: edit the stored value in place (one name appended / swap-removed + the header fields) instead of
# re-serialising every name and layer set on every HNSW.NODE.ADD (src/lib.rs:322 `index.clone().into()`, O(N))
sub("fn update_index(ctx: &Context, index_name: &str, index: &IndexT) -> Result<(), RedisError> {",
    "fn update_index(ctx: &Context, index_name: &str, index: &mut IndexT) -> Result<(), RedisError> {")
sub("        Some(_) => {\n            ctx.log_debug(format!(\"update index: {}\", index_name).as_str());\n"
    "            key.set_value::<IndexRedis>(&HNSW_INDEX_REDIS_TYPE, index.clone().into())?;",
    "        Some(ir) => {\n            ctx.log_debug(format!(\"update index: {}\", index_name).as_str());\n"
    "            index.sync_redis(ir); // O(1) per command: GpuIndex::sync_redis")
sub("    update_index(ctx, &index_name, &index)?;", "    update_index(ctx, &index_name, &mut index)?;", 2)
# HNSW.NODE.ADD / HNSW.NODE.DEL: the update closure gets a view of the engine's node
sub("    let up = |name: String, node: Node<f32>| {\n        write_node(ctx, &name, (&node).into()).unwrap();\n    };",
    "    let up = |name: String, node: NodeView| {\n        write_node(ctx, &name, node_redis_of(&node)).unwrap();\n    };", 2)
sub("    let node = index.nodes.get(&node_name).unwrap();\n    write_node(ctx, &node_name, node.into())?;",
    "    let node = index.node(&node_name).unwrap();\n    write_node(ctx, &node_name, node_redis_of(&node))?;")
cut("    let node = index.nodes.get(&node_name).unwrap();\n    if Arc::strong_count", "    let up = |name: String, node: NodeView|",
    "    // (no per-node Arc to be busy: the engine orders a delete behind every search in flight)\n"
    "    if !index.contains(&node_name) {\n"
    "        return Err(format!(\"Node: {:?} does not exist\", &node_name).into());\n"
    "    }\n\n")
# HNSW.SEARCH reply
sub("                let sr: SearchResultRedis = r.into();",
    "                let sr = SearchResultRedis { sim: r.sim as f64, name: r.name.clone() };")
src += '''
// ---- the two conversions that used to walk Index / Node objects (src/types.rs:62-91, 292-309) ----
fn index_redis_of(index: &GpuIndex) -> IndexRedis {
    let info = index.info();
    IndexRedis {
        name: index.name.clone(),
        mfunc_kind: "Euclidean".to_owned(),
        data_dim: index.data_dim,
        m: index.m,
        m_max: index.m_max,
        m_max_0: index.m_max_0,
        ef_construction: index.ef_construction,
        level_mult: index.level_mult,
        node_count: info.node_count as usize,
        max_layer: info.max_layer as usize,
        layers: index.layers(),
        nodes: index.node_names().cloned().collect(),
        enterpoint: index.enterpoint(),
    }
}

fn node_redis_of(node: &NodeView) -> NodeRedis {
    NodeRedis {
        data: node.data(),
        neighbors: node.neighbors(),
    }
}
'''
with tempfile.TemporaryDirectory() as tmp:
    os.makedirs(os.path.join(tmp, "a", "src"))
    os.makedirs(os.path.join(tmp, "b", "src"))
    open(os.path.join(tmp, "a", "src", "lib.rs"), "w").write(open(os.path.join(REF, "src", "lib.rs")).read())
    open(os.path.join(tmp, "b", "src", "lib.rs"), "w").write(src)
    p = subprocess.run(["diff", "-U2", "a/src/lib.rs", "b/src/lib.rs"], cwd=tmp, capture_output=True, text=True)
    out = "\n".join(l for l in p.stdout.split("\n") if True)
    # drop the timestamps diff puts on the header lines
    lines = out.split("\n")
    lines[0] = "--- a/src/lib.rs"
    lines[1] = "+++ b/src/lib.rs"
    open(os.path.join(ROOT, "integration", "rust", "lib_rs.diff"), "w").write("\n".join(lines))
    # and prove it applies to a pristine copy
    chk = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "integration", "rust", "lib_rs.diff")],
                         cwd=os.path.join(tmp, "a"), capture_output=True, text=True)
    print(chk.stdout.strip() or chk.stderr.strip())
    assert chk.returncode == 0
print("wrote integration/rust/lib_rs.diff (%d lines)" % len(lines))
