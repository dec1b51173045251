//! `hnsw::params::Params` (reference: hnsw/src/params.rs:5-13), filled from the native handle.
pub type NodeID = u32; // graph/src/lib.rs:1

#[derive(Debug, Clone)]
pub struct Params {
    pub ep: NodeID,
    pub m: usize,
    pub mmax: usize,
    pub mmax0: usize,
    pub ml: f32,
    pub ef_cons: usize,
    pub dim: usize,
}

pub fn get_default_ml(m: usize) -> f32 {
    1.0 / (m as f32).ln() // hnsw/src/params.rs:15-17
}
