//! extern "C" view of include/hnsw_mi355x.h -- the entry points the Redis module uses.
//! Each declaration must match the header argument for argument (tests/test_rust_shim_cpu.py checks
//! names, arity and scalar types against it).  Ids are dense u32 in insertion order; names never cross.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct hnsw_index {
    _private: [u8; 0],
}

/// one process, several GPUs: the primary index plus one replica per further device
#[repr(C)]
pub struct hnsw_group {
    _private: [u8; 0],
}

pub const HNSW_OK: c_int = 0;
pub const HNSW_ERR_DIM_MISMATCH: c_int = 1;
pub const HNSW_ERR_DUPLICATE: c_int = 2;
pub const HNSW_ERR_NOT_FOUND: c_int = 3;
pub const HNSW_ERR_DEVICE: c_int = 4;
pub const HNSW_ERR_INVALID: c_int = 5;
pub const HNSW_ERR_CAPACITY: c_int = 6;

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct hnsw_info {
    pub dim: u32,
    pub m: u32,
    pub m_max: u32,
    pub m_max0: u32,
    pub ef_construction: u32,
    pub node_count: u32,
    pub max_layer: u32,
    pub enterpoint: i64,
    pub stride0: u32,
    pub stride_upper: u32,
    pub max_degree0: u32,
    pub max_degree_upper: u32,
    pub hbm_bytes: u64,
    pub allocated_ids: u32,
}

extern "C" {
    // Index::new  core.rs:322-347
    pub fn hnsw_create(dim: u32, m: u32, ef_construction: u32, seed: u64, device: c_int, out: *mut *mut hnsw_index) -> c_int;
    pub fn hnsw_destroy(h: *mut hnsw_index);
    pub fn hnsw_last_error(h: *const hnsw_index) -> *const c_char;
    // Index::add_node -> insert  core.rs:383-412, 489-599
    pub fn hnsw_add(h: *mut hnsw_index, v: *const f32, dim: u32, level: i32, out_id: *mut u32, touched: *mut u32,
                    touched_cap: u32, n_touched: *mut u32) -> c_int;
    pub fn hnsw_add_batch(h: *mut hnsw_index, v: *const f32, n: u32, dim: u32, levels: *const i32, mode: u32) -> c_int;
    // Index::delete_node  core.rs:414-475
    pub fn hnsw_delete(h: *mut hnsw_index, id: u32, touched: *mut u32, touched_cap: u32, n_touched: *mut u32) -> c_int;
    // Index::search_knn  core.rs:477-486, 865-892
    pub fn hnsw_search(h: *mut hnsw_index, q: *const f32, dim: u32, k: u32, ids: *mut u32, sims: *mut f32,
                       n_out: *mut u32) -> c_int;
    pub fn hnsw_search_batch(h: *mut hnsw_index, q: *const f32, b: u32, dim: u32, k: u32, ids: *mut u32,
                             sims: *mut f32, n_out: *mut u32) -> c_int;
    pub fn hnsw_search_batch_device(h: *mut hnsw_index, dq: *const f32, b: u32, dim: u32, k: u32, d_ids: *mut u32,
                                    d_sims: *mut f32, d_n_out: *mut u32, stream: *mut c_void) -> c_int;
    // make_index  src/lib.rs:252-315
    pub fn hnsw_import(h: *mut hnsw_index, n: u32, vectors: *const f32, levels: *const u32, enterpoint: i64,
                       n_layers: u32, row_ptr: *const *const u64, col: *const *const u32) -> c_int;
    // IndexRedis / NodeRedis write-through  src/types.rs:62-91, 292-309
    pub fn hnsw_get_info(h: *mut hnsw_index, info: *mut hnsw_info) -> c_int;
    pub fn hnsw_get_levels(h: *mut hnsw_index, levels: *mut u32) -> c_int;
    pub fn hnsw_get_level(h: *mut hnsw_index, id: u32, level: *mut u32) -> c_int;
    pub fn hnsw_get_vector(h: *mut hnsw_index, id: u32, out: *mut f32) -> c_int;
    pub fn hnsw_get_neighbors(h: *mut hnsw_index, id: u32, layer: u32, out: *mut u32, cap: u32, n: *mut u32) -> c_int;
    // snapshot: what save_index / load_index may stream instead of one key per node  src/types.rs:176-284
    pub fn hnsw_serialize_size(h: *mut hnsw_index, bytes: *mut u64) -> c_int;
    pub fn hnsw_serialize(h: *mut hnsw_index, buf: *mut c_void, cap: u64, written: *mut u64) -> c_int;
    pub fn hnsw_deserialize(buf: *const c_void, bytes: u64, seed: u64, device: c_int, out: *mut *mut hnsw_index) -> c_int;
    // the tie census: decisions that compared equal similarities of two different nodes (core.rs:292-300 orders SimPair by
    // sim alone and leaves ties to BinaryHeap; the engine breaks them by id).  out4[2] / [3] after an add_node: zero = the
    // links are what this crate's own insert() would have made
    pub fn hnsw_get_tie_counters(h: *mut hnsw_index, out4: *mut u64) -> c_int;
    pub fn hnsw_reset_counters(h: *mut hnsw_index) -> c_int;
    pub fn hnsw_set_tuning(h: *mut hnsw_index, key: *const c_char, value: i64) -> c_int;
    // more than one GPU behind one Redis process: searches shard over the members, writes are replayed on each
    pub fn hnsw_group_create(primary: *mut hnsw_index, devices: *const c_int, n_devices: u32, seed: u64,
                             out: *mut *mut hnsw_group) -> c_int;
    pub fn hnsw_group_destroy(g: *mut hnsw_group);
    pub fn hnsw_group_last_error(g: *const hnsw_group) -> *const c_char;
    pub fn hnsw_group_size(g: *const hnsw_group) -> u32;
    pub fn hnsw_group_member(g: *mut hnsw_group, i: u32) -> *mut hnsw_index;
    pub fn hnsw_group_refresh(g: *mut hnsw_group) -> c_int;
    pub fn hnsw_group_search_batch(g: *mut hnsw_group, q: *const f32, b: u32, dim: u32, k: u32, ids: *mut u32,
                                   sims: *mut f32, n_out: *mut u32) -> c_int;
    pub fn hnsw_group_add(g: *mut hnsw_group, v: *const f32, dim: u32, level: i32, out_id: *mut u32, touched: *mut u32,
                          touched_cap: u32, n_touched: *mut u32) -> c_int;
    pub fn hnsw_group_delete(g: *mut hnsw_group, id: u32, touched: *mut u32, touched_cap: u32, n_touched: *mut u32) -> c_int;
    pub fn hnsw_group_add_batch(g: *mut hnsw_group, v: *const f32, n: u32, dim: u32, levels: *const i32, mode: u32) -> c_int;
}
