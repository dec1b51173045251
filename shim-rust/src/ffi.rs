//! extern "C" declarations of include/hnsw_mi355x.h (only what the shim needs).
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct HnswIndex {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct HnswParams {
    pub ep: u32,
    pub vec_kind: u32,
    pub m: u64,
    pub mmax: u64,
    pub mmax0: u64,
    pub ml: f32,
    pub _pad: u32,
    pub ef_cons: u64,
    pub dim: u64,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct HnswQueryStats {
    pub n_dist: u32,
    pub n_exp: u32,
    pub sum_deg: u32,
    pub status: i32,
}

pub const HNSW_OK: c_int = 0;
pub const HNSW_ERR_BAD_DIM: c_int = -1;
pub const HNSW_ERR_NAN_INPUT: c_int = -2;
pub const HNSW_VEC_QUANT8: c_int = 0;

extern "C" {
    pub fn hnsw_last_error() -> *const c_char;
    pub fn hnsw_create(m: u32, ef_cons: u32, dim: u32, vec_kind: c_int, out: *mut *mut HnswIndex) -> c_int;
    pub fn hnsw_free(h: *mut HnswIndex);
    pub fn hnsw_clone(h: *const HnswIndex, out: *mut *mut HnswIndex) -> c_int;
    pub fn hnsw_get_params(h: *const HnswIndex, out: *mut HnswParams) -> c_int;
    pub fn hnsw_set_ep(h: *mut HnswIndex, ep: u32) -> c_int;
    pub fn hnsw_insert_bulk(h: *mut HnswIndex, rows: *const f32, n: u64, nb_threads: u32, verbose: c_int) -> c_int;
    pub fn hnsw_insert_bulk_device(
        h: *mut HnswIndex, rows: *const f32, n: u64, nb_threads: u32, verbose: c_int, levels: *const u8,
    ) -> c_int;
    pub fn hnsw_set_option(h: *mut HnswIndex, key: *const c_char, value: i64) -> c_int;
    pub fn hnsw_insert_vec(h: *mut HnswIndex, v: *const f32, out_id: *mut u32) -> c_int;
    pub fn hnsw_search(h: *mut HnswIndex, q: *const f32, n: u32, ef: u32, ids: *mut u32, count: *mut u32) -> c_int;
    pub fn hnsw_search_batch(
        h: *mut HnswIndex, q: *const f32, nq: u64, n: u32, ef: u32, ids: *mut u32, dists: *mut f32,
        counts: *mut u32, stats: *mut HnswQueryStats,
    ) -> c_int;
    pub fn hnsw_search_batch_device(
        h: *mut HnswIndex, d_q: *const f32, nq: u64, n: u32, ef: u32, d_ids: *mut u32, d_dists: *mut f32,
        d_counts: *mut u32, d_stats: *mut HnswQueryStats, stream: *mut c_void,
    ) -> c_int;
    pub fn hnsw_search_batch_device_finish(
        h: *mut HnswIndex, d_q: *const f32, nq: u64, n: u32, ef: u32, d_ids: *mut u32, d_dists: *mut f32,
        d_counts: *mut u32, d_stats: *mut HnswQueryStats, stream: *mut c_void,
    ) -> c_int;
    pub fn hnsw_brute_force_fast(h: *mut HnswIndex, q: *const f32, nq: u64, k: u32, ids: *mut u32, dists: *mut f32) -> c_int;
    pub fn hnsw_len(h: *const HnswIndex) -> u64;
    pub fn hnsw_distance(h: *const HnswIndex, a: u32, b: u32, out: *mut f32) -> c_int;
    pub fn hnsw_get_vector(h: *const HnswIndex, id: u32, out: *mut f32) -> c_int;
    pub fn hnsw_get_level(h: *const HnswIndex, id: u32, out: *mut u32) -> c_int;
    pub fn hnsw_layer_count(h: *const HnswIndex) -> u32;
    pub fn hnsw_layer_nb_nodes(h: *const HnswIndex, layer: u32) -> u64;
    pub fn hnsw_layer_m(h: *const HnswIndex, layer: u32) -> u32;
    pub fn hnsw_layer_nodes(h: *const HnswIndex, layer: u32, out: *mut u32, cap: u64, n: *mut u64) -> c_int;
    pub fn hnsw_neighbors(h: *const HnswIndex, layer: u32, id: u32, buf: *mut u32, cap: u32, deg: *mut u32) -> c_int;
    pub fn hnsw_check_param_compliance(h: *const HnswIndex, ok: *mut c_int) -> c_int;
    pub fn hnsw_save(h: *const HnswIndex, dir: *const c_char) -> c_int;
    pub fn hnsw_load(dir: *const c_char, out: *mut *mut HnswIndex) -> c_int;
}

pub fn last_error() -> String {
    unsafe {
        let p = hnsw_last_error();
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}
