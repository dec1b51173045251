//! `hnsw::helpers` (reference: hnsw/src/helpers/mod.rs): the two modules the reference's callers import --
//! `helpers::args` (eval_glove/src/main.rs:8) and `helpers::glove` (eval_glove/src/main.rs:9,
//! hnsw/benches/hnsw_benchmarks.rs:2).  Not carried over: `helpers::data` (`split` has no caller anywhere in the
//! reference and `load_bf_data` reads the author's home directory) and `helpers::get_progress_bar` (returns an
//! indicatif type; progress bars are out of scope).
pub mod args;
pub mod glove;
