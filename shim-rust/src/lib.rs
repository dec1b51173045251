//! Drop-in replacement for the reference's `hnsw` crate (`hnsw/src/lib.rs`): same module paths
//! `hnsw::template::HNSW` and `hnsw::params::Params`, same method names and signatures,
//! implemented on the C ABI of libhnsw_mi355x.so (include/hnsw_mi355x.h).  Callers such as
//! `eval_glove/src/main.rs:37-41` and `hnsw/benches/hnsw_benchmarks.rs:16-25` compile against it
//! unchanged.  (The files are named after what they hold -- an FFI handle wrapper -- and mounted at
//! the reference's module paths here.)
pub mod ffi;
#[path = "gpu_params.rs"]
pub mod params;
#[path = "gpu_index.rs"]
pub mod template;
