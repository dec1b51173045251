//! `hnsw::template::HNSW` on libhnsw_mi355x.so -- the reference API of hnsw/src/template.rs.
//!
//! Error mapping: the C ABI never unwinds.  Where the reference returns `Err(String)` the shim
//! returns `Err(hnsw_last_error())`; where it panics (dimension mismatch template.rs:253-262, NaN
//! distance graph/src/dist.rs:32, save I/O template.rs:45-71) the shim panics with the same text.
use std::ffi::CString;
use std::path::Path;
use std::ptr;

use crate::ffi::*;
use crate::params::{NodeID, Params};

pub struct HNSW {
    handle: *mut HnswIndex,
    /// public field in the reference (template.rs:37); refreshed after every mutating call
    pub params: Params,
}

// `ann_by_vector(&self)` is re-entrant in the reference; the native search is too
unsafe impl Send for HNSW {}
unsafe impl Sync for HNSW {}

/// What `get_point(id)` hands out (points/src/point.rs:6-10); `get_vals` is VecBase::get_vals.
pub struct Point {
    pub id: NodeID,
    pub level: u8,
    vals: Vec<f32>,
}
impl Point {
    pub fn get_vals(&self) -> Vec<f32> {
        self.vals.clone()
    }
}

/// Read-only view of a layer (graph/src/graph.rs:9-16)
pub struct Graph<'a> {
    index: &'a HNSW,
    pub level: usize,
    pub m: usize,
}
impl<'a> Graph<'a> {
    pub fn nb_nodes(&self) -> usize {
        unsafe { hnsw_layer_nb_nodes(self.index.handle, self.level as u32) as usize }
    }
    pub fn iter_nodes(&self) -> impl Iterator<Item = NodeID> {
        let n = self.nb_nodes();
        let mut out = vec![0u32; n.max(1)];
        let mut cnt = 0u64;
        unsafe { hnsw_layer_nodes(self.index.handle, self.level as u32, out.as_mut_ptr(), n as u64, &mut cnt) };
        out.truncate(n);
        out.into_iter()
    }
    pub fn neighbors(&self, node: NodeID) -> Result<std::collections::HashSet<NodeID>, String> {
        // a row has no fixed length (degree may exceed the layer's cap, graph.rs:37-52 never checks it): ask for the
        // degree first, then for exactly that many ids
        let mut deg = 0u32;
        let rc = unsafe { hnsw_neighbors(self.index.handle, self.level as u32, node, ptr::null_mut(), 0, &mut deg) };
        if rc != HNSW_OK {
            return Err(last_error()); // GraphError::NodeNotInGraph
        }
        let mut buf = vec![0u32; (deg as usize).max(1)];
        let cap = buf.len() as u32;
        let rc = unsafe { hnsw_neighbors(self.index.handle, self.level as u32, node, buf.as_mut_ptr(), cap, &mut deg) };
        if rc != HNSW_OK {
            return Err(last_error());
        }
        buf.truncate(deg as usize);
        Ok(buf.into_iter().collect())
    }
    pub fn degree(&self, node: NodeID) -> Result<usize, String> {
        let mut deg = 0u32;
        let rc = unsafe { hnsw_neighbors(self.index.handle, self.level as u32, node, ptr::null_mut(), 0, &mut deg) };
        if rc != HNSW_OK {
            Err(last_error())
        } else {
            Ok(deg as usize)
        }
    }
}

impl HNSW {
    fn refresh_params(&mut self) {
        let mut p = HnswParams::default();
        unsafe { hnsw_get_params(self.handle, &mut p) };
        self.params = Params::from(&p);
    }

    fn from_handle(handle: *mut HnswIndex) -> Self {
        let mut s = HNSW {
            handle,
            params: Params::unset(),
        };
        s.refresh_params();
        s
    }

    /// template.rs:133-144
    pub fn new(m: usize, ef_cons: Option<usize>, dim: usize) -> Self {
        let mut h = ptr::null_mut();
        let rc = unsafe { hnsw_create(m as u32, ef_cons.unwrap_or(0) as u32, dim as u32, HNSW_VEC_QUANT8, &mut h) };
        assert_eq!(rc, HNSW_OK, "{}", last_error());
        Self::from_handle(h)
    }

    /// Not in the reference: route later `insert_bulk` calls to the on-device build (searches,
    /// heuristic, connect and prune on the GPU).  The graph is a valid HNSW graph
    /// of the same recall, not the identical one the CPU algorithm would produce.
    pub fn gpu_build(self, on: bool) -> Self {
        let key = std::ffi::CString::new("gpu_build").unwrap();
        let rc = unsafe { hnsw_set_option(self.handle, key.as_ptr(), if on { 2 } else { 0 }) };
        assert_eq!(rc, HNSW_OK, "{}", last_error());
        self
    }

    /// template.rs:388-444 (consumes and returns the index)
    pub fn insert_bulk(mut self, vectors: Vec<Vec<f32>>, nb_threads: usize, verbose: bool) -> Result<HNSW, String> {
        let dim = self.params.dim;
        let mut flat = Vec::with_capacity(vectors.len() * dim);
        for v in vectors.iter() {
            if v.len() != dim {
                // check_points_dim, template.rs:253-262
                panic!("The current index dimension is {0}, but tried inserting points of dimension {1}", dim, v.len());
            }
            flat.extend_from_slice(v);
        }
        let rc = unsafe {
            hnsw_insert_bulk(self.handle, flat.as_ptr(), vectors.len() as u64, nb_threads as u32, verbose as i32)
        };
        self.refresh_params();
        match rc {
            HNSW_OK => Ok(self),
            HNSW_ERR_NAN_INPUT | HNSW_ERR_BAD_DIM => panic!("{}", last_error()),
            _ => Err(last_error()),
        }
    }

    /// template.rs:165-173
    pub fn insert_vec(&mut self, vector: &Vec<f32>) -> Result<NodeID, String> {
        if vector.len() != self.params.dim {
            panic!("The current index dimension is {0}, but tried inserting points of dimension {1}", self.params.dim, vector.len());
        }
        let mut id = 0u32;
        let rc = unsafe { hnsw_insert_vec(self.handle, vector.as_ptr(), &mut id) };
        self.refresh_params();
        match rc {
            HNSW_OK => Ok(id),
            HNSW_ERR_NAN_INPUT => panic!("{}", last_error()),
            _ => Err(last_error()),
        }
    }

    /// template.rs:306-335 -- one query per call, ids only, fewer than n when ef < n
    pub fn ann_by_vector(&self, vector: &Vec<f32>, n: usize, ef: usize) -> Result<Vec<NodeID>, String> {
        assert_eq!(vector.len(), self.params.dim, "query dimension"); // stricter than the reference (Q4)
        let mut ids = vec![0u32; n.max(1)];
        let mut count = 0u32;
        let rc = unsafe { hnsw_search(self.handle, vector.as_ptr(), n as u32, ef as u32, ids.as_mut_ptr(), &mut count) };
        match rc {
            HNSW_OK => {
                ids.truncate(count as usize);
                Ok(ids)
            }
            HNSW_ERR_NAN_INPUT => panic!("{}", last_error()), // Dist::cmp unwrap, graph/src/dist.rs:32
            _ => Err(last_error()),
        }
    }

    /// New: the batched GPU entry point (`nq` queries, row-major), ids padded with u32::MAX
    pub fn ann_by_vectors(&self, queries: &[f32], n: usize, ef: usize) -> Result<(Vec<NodeID>, Vec<f32>), String> {
        let nq = queries.len() / self.params.dim;
        let mut ids = vec![u32::MAX; nq * n];
        let mut dists = vec![f32::INFINITY; nq * n];
        let rc = unsafe {
            hnsw_search_batch(self.handle, queries.as_ptr(), nq as u64, n as u32, ef as u32, ids.as_mut_ptr(),
                              dists.as_mut_ptr(), ptr::null_mut(), ptr::null_mut())
        };
        if rc != HNSW_OK {
            return Err(last_error());
        }
        Ok((ids, dists))
    }

    pub fn len(&self) -> usize {
        unsafe { hnsw_len(self.handle) as usize } // template.rs:146
    }

    pub fn distance(&self, a: NodeID, b: NodeID) -> Option<f32> {
        let mut d = 0f32; // template.rs:150-152
        if unsafe { hnsw_distance(self.handle, a, b, &mut d) } == HNSW_OK { Some(d) } else { None }
    }

    pub fn get_point(&self, point_id: NodeID) -> Option<Point> {
        let mut vals = vec![0f32; self.params.dim]; // template.rs:154-156
        let mut level = 0u32;
        unsafe {
            if hnsw_get_vector(self.handle, point_id, vals.as_mut_ptr()) != HNSW_OK {
                return None;
            }
            hnsw_get_level(self.handle, point_id, &mut level);
        }
        Some(Point { id: point_id, level: level as u8, vals })
    }

    pub fn get_layer(&self, layer_nb: usize) -> Graph<'_> {
        let n = unsafe { hnsw_layer_count(self.handle) } as usize; // template.rs:192-194
        if layer_nb >= n {
            panic!("Layer {layer_nb} not found in the structure."); // layers.rs:25-30
        }
        Graph { index: self, level: layer_nb, m: unsafe { hnsw_layer_m(self.handle, layer_nb as u32) } as usize }
    }

    /// template.rs:158-163: one line per node of the layer with its degree
    pub fn layer_degrees(&self, layer_nb: usize) {
        let layer = self.get_layer(layer_nb);
        for node in layer.iter_nodes() {
            println!("{}", layer.degree(node).unwrap());
        }
    }

    /// template.rs:372-384: the same lines in the same order
    pub fn print_index(&self) {
        println!("m = {}", self.params.m);
        println!("mmax = {}", self.params.mmax);
        println!("mmax0 = {}", self.params.mmax0);
        println!("ml = {}", self.params.ml);
        println!("ef_cons = {}", self.params.ef_cons);
        let nb_layers = unsafe { hnsw_layer_count(self.handle) } as usize;
        println!("Nb. layers = {}", nb_layers);
        println!("Nb. of points = {}", self.len());
        for idx in 0..nb_layers {
            println!("NB. nodes in layer {idx}: {}", self.get_layer(idx).nb_nodes());
        }
        println!("ep: {:?}", self.params.ep);
    }

    /// New: exact k nearest stored points of each query under the index's own metric (the reference's
    /// `helpers::glove::brute_force_nns`, glove.rs:73-109, as one GPU scan): ids and distances, row-major
    pub fn brute_force(&self, queries: &[f32], k: usize) -> Result<(Vec<NodeID>, Vec<f32>), String> {
        let nq = queries.len() / self.params.dim;
        let mut ids = vec![u32::MAX; nq * k];
        let mut dists = vec![f32::INFINITY; nq * k];
        let rc = unsafe {
            hnsw_brute_force_fast(self.handle, queries.as_ptr(), nq as u64, k as u32, ids.as_mut_ptr(), dists.as_mut_ptr())
        };
        if rc != HNSW_OK {
            return Err(last_error());
        }
        Ok((ids, dists))
    }

    pub fn assert_param_compliance(&self) {
        let mut ok = 0;
        unsafe { hnsw_check_param_compliance(self.handle, &mut ok) };
        if ok != 0 {
            println!("Index complies with params.") // template.rs:367-369
        }
    }

    /// template.rs:43-73 (panics on I/O errors like the reference)
    pub fn save(&self, dir: &Path) {
        let c = CString::new(dir.to_str().expect("utf-8 path")).unwrap();
        let rc = unsafe { hnsw_save(self.handle, c.as_ptr()) };
        if rc != HNSW_OK {
            panic!("{}", last_error());
        }
    }

    /// template.rs:75-131
    pub fn load(dir: &Path) -> Result<Self, String> {
        let c = CString::new(dir.to_str().ok_or("non utf-8 path")?).map_err(|e| e.to_string())?;
        let mut h = ptr::null_mut();
        let rc = unsafe { hnsw_load(c.as_ptr(), &mut h) };
        if rc != HNSW_OK {
            return Err(last_error());
        }
        Ok(Self::from_handle(h))
    }
}

/// template.rs:623-628: M = 12, one build thread
pub fn make_rand_index_full(n: usize, dim: usize) -> HNSW {
    HNSW::new(12, None, dim).insert_bulk(make_rand_vectors(n, dim), 1, false).unwrap()
}

/// template.rs:630-638: n vectors of dim values uniform in [0, 1).  The reference draws from
/// `rand::thread_rng()` (unseeded); the shim has no dependencies, so it runs a splitmix64 stream seeded
/// from the clock and takes the top 24 bits of each word, the same `bits * 2^-24` mapping rand uses.
pub fn make_rand_vectors(n: usize, dim: usize) -> Vec<Vec<f32>> {
    let mut state = std::time::SystemTime::now()
        .duration_since(std::time::UNIX_EPOCH)
        .map(|d| d.as_nanos() as u64)
        .unwrap_or(0x5EED_0001)
        ^ (&n as *const usize as u64);
    let mut next = move || {
        state = state.wrapping_add(0x9E37_79B9_7F4A_7C15);
        let mut z = state;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
        z ^ (z >> 31)
    };
    (0..n)
        .map(|_| (0..dim).map(|_| (next() >> 40) as f32 * (1.0 / 16_777_216.0)).collect())
        .collect()
}

impl Clone for HNSW {
    fn clone(&self) -> Self {
        let mut h = ptr::null_mut(); // #[derive(Clone)], template.rs:35
        let rc = unsafe { hnsw_clone(self.handle, &mut h) };
        assert_eq!(rc, HNSW_OK, "{}", last_error());
        Self::from_handle(h)
    }
}

impl Drop for HNSW {
    fn drop(&mut self) {
        unsafe { hnsw_free(self.handle) }
    }
}

impl std::fmt::Debug for HNSW {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("HNSW").field("params", &self.params).field("len", &self.len()).finish()
    }
}
