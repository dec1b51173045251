//! Drop-in replacement for the reference's `hnsw` crate (`hnsw/src/lib.rs`): same module path
//! `hnsw::template::HNSW`, same method names and signatures, implemented on the C ABI of
//! libhnsw_mi355x.so (include/hnsw_mi355x.h).  Callers such as `eval_glove/src/main.rs:37-41`
//! and `hnsw/benches/hnsw_benchmarks.rs:16-25` compile against it unchanged.
pub mod ffi;
pub mod params;
pub mod template;
