//! GpuIndex -- the methods src/lib.rs calls on Index<f32,f32> (src/hnsw/core.rs:322-486), executed by the
//! MI355X engine.  The struct keeps what the reference keeps in `Index.nodes` (core.rs:316): the node names;
//! the engine speaks dense ids.  Error strings are the reference's (core.rs:390, 408, 421, 479), rendered by
//! error_string() exactly as clients see them today (core.rs:42-46).
use super::ffi;
use crate::types::IndexRedis;
use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::Mutex;

#[derive(Debug)]
pub enum HNSWError {
    Str(&'static str),
    String(String),
}
impl HNSWError {
    pub fn error_string(&self) -> String {
        format!("{:?}", self) // core.rs:42-46
    }
}
impl From<String> for HNSWError {
    fn from(s: String) -> Self {
        HNSWError::String(s)
    }
}

pub struct SearchResult {
    pub sim: f32,     // -(squared L2), metrics.rs:75
    pub name: String, // last '.'-separated segment of the node key, core.rs:885-887
    pub id: u32,
}

/// what the `update_fn` closure of add_node / delete_node receives instead of a Node<f32>:
/// enough to rebuild the node's NodeRedis (src/types.rs:292-309) from the engine
pub struct NodeView<'a> {
    pub index: &'a GpuIndex,
    pub id: u32,
}
impl<'a> NodeView<'a> {
    pub fn data(&self) -> Vec<f32> {
        let mut v = vec![0f32; self.index.data_dim];
        unsafe { ffi::hnsw_get_vector(self.index.h, self.id, v.as_mut_ptr()) };
        v
    }
    /// neighbour NAMES per layer, in stored order (what NodeRedis.neighbors holds)
    pub fn neighbors(&self) -> Vec<Vec<String>> {
        let mut info = ffi::hnsw_info::default();
        unsafe { ffi::hnsw_get_info(self.index.h, &mut info) };
        let cap = info.stride0.max(info.stride_upper) as usize;
        let mut out = Vec::new();
        let mut buf = vec![0u32; cap];
        for layer in 0..=self.index.levels[self.id as usize] {
            let mut n = 0u32;
            unsafe { ffi::hnsw_get_neighbors(self.index.h, self.id, layer, buf.as_mut_ptr(), cap as u32, &mut n) };
            out.push(buf[..n as usize].iter().map(|&j| self.index.names[j as usize].clone().unwrap()).collect());
        }
        out
    }
}

/// room for the ids one insert / delete passes to update_fn (core.rs:580-584, :441-446); the engine reports the count
/// even when it is larger, and the call then fails instead of skipping write-throughs
const TOUCHED_CAP: usize = 65536;

pub struct GpuIndex {
    pub name: String,
    pub data_dim: usize,
    pub m: usize,
    pub m_max: usize,
    pub m_max_0: usize,
    pub ef_construction: usize,
    pub level_mult: f64,
    h: *mut ffi::hnsw_index,
    names: Vec<Option<String>>, // id -> "hnsw.{idx}.{node}"; None = deleted (ids are never reused)
    ids: HashMap<String, u32>,  // name -> id
    levels: Vec<u32>,           // id -> top layer (the layer sets of IndexRedis.layers)
    // where each live id sits in the persisted hnswindex value (IndexRedis.nodes / IndexRedis.layers[level]),
    // so that HNSW.NODE.ADD / HNSW.NODE.DEL edit that value in O(1) instead of rebuilding it (sync_redis)
    pos_nodes: Vec<u32>,
    pos_layer: Vec<u32>,
    pending: Vec<Change>,       // what the commands since the last sync_redis did
    gone: HashMap<String, u32>, // names deleted since the last sync_redis (they have left `ids` but still sit in the stored value)
    // search_knn takes &self (core.rs:477) and the module lets readers in concurrently (try_read, src/lib.rs:474),
    // but one engine handle serves one caller at a time (staging buffers, stream, error string): readers queue here
    search_lock: Mutex<()>,
}
const NOT_STORED: u32 = u32::MAX;
enum Change {
    Added(u32),
    Removed(u32),
}
unsafe impl Send for GpuIndex {}
unsafe impl Sync for GpuIndex {} // every FFI call that takes &self goes through search_lock; writers hold try_write

impl GpuIndex {
    fn last_error(&self) -> HNSWError {
        let p = unsafe { ffi::hnsw_last_error(self.h) };
        HNSWError::String(unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned())
    }

    /// id of a name that sits in the stored hnswindex value: a name whose removal is still queued first (it left `ids`
    /// when the command ran; a node re-added under the same name enters the stored value only when ITS entry is applied)
    fn stored_id(&self, name: &str) -> Option<&u32> {
        self.gone.get(name).or_else(|| self.ids.get(name))
    }

    /// Index::new(name, mfunc, data_dim, m, ef_construction)  core.rs:322-347 (the metric is Euclidean, :332)
    pub fn new(name: &str, data_dim: usize, m: usize, ef_construction: usize) -> Result<Self, HNSWError> {
        let mut h = std::ptr::null_mut();
        let seed: u64 = rand::random(); // core.rs:344 seeds from entropy
        let st = unsafe { ffi::hnsw_create(data_dim as u32, m as u32, ef_construction as u32, seed, 0, &mut h) };
        let idx = GpuIndex {
            name: name.to_owned(),
            data_dim,
            m,
            m_max: m,
            m_max_0: 2 * m,
            ef_construction,
            level_mult: 1.0 / (m as f64).ln(),
            h,
            names: Vec::new(),
            ids: HashMap::new(),
            levels: Vec::new(),
            pos_nodes: Vec::new(),
            pos_layer: Vec::new(),
            pending: Vec::new(),
            gone: HashMap::new(),
            search_lock: Mutex::new(()),
        };
        if st != ffi::HNSW_OK {
            return Err(idx.last_error());
        }
        Ok(idx)
    }

    pub fn node_count(&self) -> usize {
        self.ids.len()
    }
    pub fn node_names(&self) -> impl Iterator<Item = &String> {
        self.ids.keys()
    }
    pub fn contains(&self, name: &str) -> bool {
        self.ids.contains_key(name)
    }
    pub fn node(&self, name: &str) -> Option<NodeView> {
        self.ids.get(name).map(|&id| NodeView { index: self, id })
    }
    pub fn info(&self) -> ffi::hnsw_info {
        let mut info = ffi::hnsw_info::default();
        unsafe { ffi::hnsw_get_info(self.h, &mut info) };
        info
    }
    pub fn enterpoint(&self) -> Option<String> {
        let e = self.info().enterpoint;
        if e < 0 { None } else { self.names[e as usize].clone() }
    }
    /// IndexRedis.layers: layer l holds the names of the nodes whose TOP layer is l -- a node is in exactly one
    /// set (core.rs:596 `self.layers[l].insert`, src/types.rs:73-82), which is what delete_node relies on when it
    /// removes a node from one set and breaks (core.rs:426-430).  Used for replies (HNSW.GET) and for the value
    /// HNSW.NEW stores; the stored value is kept current by sync_redis, not rebuilt from here.
    pub fn layers(&self) -> Vec<Vec<String>> {
        let top = self.info().max_layer as usize;
        let mut out = vec![Vec::new(); if self.ids.is_empty() { 0 } else { top + 1 }];
        for (id, name) in self.names.iter().enumerate() {
            if let Some(n) = name {
                out[self.levels[id] as usize].push(n.clone());
            }
        }
        out
    }

    /// update_index (src/lib.rs:317-332) without `index.clone().into()`: bring the stored hnswindex value up to date
    /// with what the commands since the last call did.  The reference re-serialises all names and layer sets on every
    /// HNSW.NODE.ADD (O(N) per command, O(N^2) per load); here an add appends one name to `nodes` and to the set of its
    /// top layer, a delete swap-removes the name from both, and the header fields are overwritten.
    pub fn sync_redis(&mut self, ir: &mut IndexRedis) {
        let info = self.info();
        let pending = std::mem::replace(&mut self.pending, Vec::new());
        for ch in pending {
            match ch {
                Change::Added(id) => {
                    let id = id as usize;
                    let name = match &self.names[id] {
                        Some(n) => n.clone(),
                        None => continue, // added and deleted again before the value was synchronised
                    };
                    let l = self.levels[id] as usize;
                    self.pos_nodes[id] = ir.nodes.len() as u32;
                    ir.nodes.push(name.clone());
                    while ir.layers.len() < l + 1 {
                        ir.layers.push(Vec::new()); // core.rs:590-592
                    }
                    self.pos_layer[id] = ir.layers[l].len() as u32;
                    ir.layers[l].push(name); // core.rs:596
                }
                Change::Removed(id) => {
                    let id = id as usize;
                    if self.pos_nodes[id] == NOT_STORED {
                        continue; // never reached the stored value (see Added above)
                    }
                    // positions are only ever edited here, so they describe `ir` as it is now
                    let (p, l, q) = (self.pos_nodes[id] as usize, self.levels[id] as usize, self.pos_layer[id] as usize);
                    self.pos_nodes[id] = NOT_STORED;
                    let name = ir.nodes.swap_remove(p);
                    if self.gone.get(&name) == Some(&(id as u32)) {
                        self.gone.remove(&name);
                    }
                    if p < ir.nodes.len() {
                        // the element moved into the hole may itself be pending removal (its name has left `ids`
                        // already, its own Removed entry comes later in this queue): stored_id still knows it
                        if let Some(&moved) = self.stored_id(&ir.nodes[p]) {
                            self.pos_nodes[moved as usize] = p as u32;
                        }
                    }
                    ir.layers[l].swap_remove(q); // core.rs:426-430: the one set that holds it
                    if q < ir.layers[l].len() {
                        if let Some(&moved) = self.stored_id(&ir.layers[l][q]) {
                            self.pos_layer[moved as usize] = q as u32;
                        }
                    }
                }
            }
        }
        ir.node_count = info.node_count as usize;
        ir.max_layer = info.max_layer as usize;
        // core.rs:453-466: empty top layers are popped when the enterpoint goes; with no node left there is no layer
        ir.layers.truncate(if self.ids.is_empty() { 0 } else { info.max_layer as usize + 1 });
        ir.enterpoint = self.enterpoint();
        self.gone.clear();
    }

    /// add_node(&mut self, name, data, update_fn)  core.rs:383-412
    pub fn add_node(&mut self, name: &str, data: &[f32], update_fn: impl Fn(String, NodeView)) -> Result<(), HNSWError> {
        if data.len() != self.data_dim {
            return Err(format!("data dimension: {} does not match Index", data.len()).into()); // core.rs:389-391
        }
        if !self.ids.is_empty() && self.ids.contains_key(name) {
            return Err(format!("Node: {:?} already exists", name).into()); // core.rs:407-409 (after the first-node branch)
        }
        let (mut id, mut nt) = (0u32, 0u32);
        let mut touched = vec![0u32; TOUCHED_CAP];
        let st = unsafe {
            ffi::hnsw_add(self.h, data.as_ptr(), data.len() as u32, -1, &mut id, touched.as_mut_ptr(),
                          touched.len() as u32, &mut nt)
        };
        if st != ffi::HNSW_OK {
            return Err(self.last_error());
        }
        // status OK = the graph holds the node: record it BEFORE anything else can fail, or names and ids drift apart
        debug_assert_eq!(id as usize, self.names.len());
        let mut level = 0u32;
        unsafe { ffi::hnsw_get_level(self.h, id, &mut level) }; // O(1); the level the engine drew (core.rs:601-605)
        self.names.push(Some(name.to_owned()));
        self.ids.insert(name.to_owned(), id);
        self.levels.push(level);
        self.pos_nodes.push(NOT_STORED);
        self.pos_layer.push(NOT_STORED);
        self.pending.push(Change::Added(id));
        if nt as usize > touched.len() {
            // the list is incomplete (u32::MAX: the engine could not produce it): the node keys of this command
            // cannot all be rewritten -- surface it; the index itself is consistent
            return Err(format!("update_fn list of {} ids does not fit the buffer", nt).into());
        }
        for &t in &touched[..nt as usize] {
            // core.rs:580-584: every node whose links changed is written through (write_node, src/lib.rs:351-353)
            update_fn(self.names[t as usize].clone().unwrap(), NodeView { index: self, id: t });
        }
        Ok(())
    }

    /// delete_node(&mut self, name, update_fn)  core.rs:414-475
    pub fn delete_node(&mut self, name: &str, update_fn: impl Fn(String, NodeView)) -> Result<(), HNSWError> {
        let id = match self.ids.get(name) {
            Some(&id) => id,
            None => return Err(format!("Node: {:?} does not exist", name).into()), // core.rs:419-422
        };
        let (mut nt, mut touched) = (0u32, vec![0u32; TOUCHED_CAP]);
        let st = unsafe { ffi::hnsw_delete(self.h, id, touched.as_mut_ptr(), touched.len() as u32, &mut nt) };
        if st != ffi::HNSW_OK {
            return Err(self.last_error());
        }
        self.ids.remove(name); // status OK = the node is gone from the graph (see add_node)
        // keep the FIRST id removed under this name since the last sync_redis: that is the element the stored value
        // still holds (del X(id1), add X(id2), del X(id2) must leave gone[X] = id1)
        self.gone.entry(name.to_owned()).or_insert(id);
        self.names[id as usize] = None;
        self.pending.push(Change::Removed(id));
        if nt as usize > touched.len() {
            return Err(format!("update_fn list of {} ids does not fit the buffer", nt).into());
        }
        for &t in &touched[..nt as usize] {
            if let Some(n) = self.names[t as usize].clone() {
                update_fn(n, NodeView { index: self, id: t }); // core.rs:441-446
            }
        }
        Ok(())
    }

    /// search_knn(&self, data, k)  core.rs:477-486 -> :865-892
    pub fn search_knn(&self, data: &[f32], k: usize) -> Result<Vec<SearchResult>, HNSWError> {
        if data.len() != self.data_dim {
            return Err(format!("data dimension: {} does not match Index", data.len()).into()); // core.rs:478-480
        }
        if self.ids.is_empty() || k == 0 {
            return Ok(Vec::new()); // core.rs:481-483
        }
        let (mut ids, mut sims, mut n) = (vec![0u32; k], vec![0f32; k], 0u32);
        let _one_caller = self.search_lock.lock().unwrap(); // the handle's staging, stream and error string are per call
        let st = unsafe {
            ffi::hnsw_search(self.h, data.as_ptr(), data.len() as u32, k as u32, ids.as_mut_ptr(), sims.as_mut_ptr(), &mut n)
        };
        if st != ffi::HNSW_OK {
            return Err(self.last_error());
        }
        Ok((0..n as usize)
            .map(|i| SearchResult {
                sim: sims[i],
                name: self.names[ids[i] as usize].as_ref().unwrap().rsplit('.').next().unwrap().to_owned(),
                id: ids[i],
            })
            .collect())
    }

    /// make_index (src/lib.rs:252-315): rebuild from the per-node keys with ONE upload.
    /// nodes[i] = (key, data, neighbour keys per layer) in the order of IndexRedis.nodes; ids follow that order.
    /// A node's level is the layer SET it is in (`layers`, src/lib.rs:287-299; core.rs:596), never its row count: the
    /// reference saves a node that was promoted to enterpoint with l > l_max with rows 0..=l_max only (the loop of
    /// core.rs:523 starts at min(l_max, l); upper rows appear lazily, core.rs:642), and the first node of an index
    /// with no row at all (core.rs:393-405).  Missing rows are empty rows.
    pub fn from_keys(name: &str, data_dim: usize, m: usize, ef_construction: usize,
                     nodes: Vec<(String, Vec<f32>, Vec<Vec<String>>)>, layers: &[Vec<String>], max_layer: usize,
                     enterpoint: Option<String>)
                     -> Result<Self, HNSWError> {
        let mut idx = GpuIndex::new(name, data_dim, m, ef_construction)?;
        let n = nodes.len();
        if n == 0 {
            return Ok(idx);
        }
        for (i, (key, _, _)) in nodes.iter().enumerate() {
            idx.ids.insert(key.clone(), i as u32);
        }
        let mut levels = vec![NOT_STORED; n];
        let mut pos_layer = vec![NOT_STORED; n];
        for (l, set) in layers.iter().enumerate() {
            for (q, key) in set.iter().enumerate() {
                // src/lib.rs:290-293: a name in a layer set that is not a node of the index is an error
                let i = *idx.ids.get(key).ok_or_else(|| format!("Node: {} does not exist", key))? as usize;
                levels[i] = l as u32;
                pos_layer[i] = q as u32;
            }
        }
        if let Some(i) = levels.iter().position(|&l| l == NOT_STORED) {
            return Err(format!("Node: {} is in no layer set", nodes[i].0).into());
        }
        let n_layers = (max_layer + 1).max(layers.len()).max(1);
        let mut vectors = Vec::with_capacity(n * data_dim);
        let mut row_ptr = vec![vec![0u64; n + 1]; n_layers];
        let mut col: Vec<Vec<u32>> = vec![Vec::new(); n_layers];
        for (i, (key, data, nbrs)) in nodes.iter().enumerate() {
            if data.len() != data_dim {
                return Err(format!("data dimension: {} does not match Index", data.len()).into());
            }
            vectors.extend_from_slice(data);
            if nbrs.iter().skip(levels[i] as usize + 1).any(|row| !row.is_empty()) {
                return Err(format!("Node: {} has links above its layer", key).into());
            }
            for l in 0..n_layers {
                if let Some(layer) = nbrs.get(l) {
                    for nb in layer {
                        // src/lib.rs:277-281: an unknown neighbour key is an error, not a skip
                        let j = *idx.ids.get(nb).ok_or_else(|| format!("Node: {} does not exist", nb))?;
                        col[l].push(j);
                    }
                }
                row_ptr[l][i + 1] = col[l].len() as u64;
            }
        }
        let ep = match &enterpoint {
            Some(key) => *idx.ids.get(key).ok_or_else(|| format!("Node: {} does not exist", key))? as i64,
            None => return Err(HNSWError::Str("an index with nodes has an enterpoint")),
        };
        let rp: Vec<*const u64> = row_ptr.iter().map(|r| r.as_ptr()).collect();
        let cl: Vec<*const u32> = col.iter().map(|c| c.as_ptr()).collect();
        let st = unsafe {
            ffi::hnsw_import(idx.h, n as u32, vectors.as_ptr(), levels.as_ptr(), ep, n_layers as u32, rp.as_ptr(), cl.as_ptr())
        };
        if st != ffi::HNSW_OK {
            return Err(idx.last_error());
        }
        idx.names = nodes.into_iter().map(|x| Some(x.0)).collect();
        idx.levels = levels;
        idx.pos_nodes = (0..n as u32).collect(); // ids follow IndexRedis.nodes
        idx.pos_layer = pos_layer;
        Ok(idx)
    }
}

impl Drop for GpuIndex {
    fn drop(&mut self) {
        unsafe { ffi::hnsw_destroy(self.h) }
    }
}
