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

/// What `hnsw_get_params` reports (include/hnsw_mi355x.h `hnsw_params`: the native handle owns the index) as the
/// reference's public field.  `vec_kind` has no counterpart: the reference fixes it at compile time
/// (points/src/point.rs:4 `type VecType = QuantVec`).
impl From<&crate::ffi::HnswParams> for Params {
    fn from(p: &crate::ffi::HnswParams) -> Self {
        Params {
            ep: p.ep,
            m: p.m as usize,
            mmax: p.mmax as usize,
            mmax0: p.mmax0 as usize,
            ml: p.ml,
            ef_cons: p.ef_cons as usize,
            dim: p.dim as usize,
        }
    }
}

impl Params {
    /// the handle has not been asked yet (`HNSW::from_handle` fills it in at once)
    pub(crate) fn unset() -> Self {
        Params { ep: 0, m: 0, mmax: 0, mmax0: 0, ml: 0.0, ef_cons: 0, dim: 0 }
    }
}
