//! Drop-in replacement for the reference's `hnsw` crate (`hnsw/src/lib.rs`): the module paths its
//! callers import -- `hnsw::template::HNSW`, `hnsw::params::Params`, `hnsw::helpers::glove::load_glove_array`,
//! `hnsw::helpers::args::parse_args_eval` (eval_glove/src/main.rs:8-11, hnsw/benches/hnsw_benchmarks.rs:2-3)
//! -- with the same names and signatures, implemented on the C ABI of libhnsw_mi355x.so
//! (include/hnsw_mi355x.h).  Source only: the build image has no rustc, so this crate has never been
//! compiled (tests/test_shim_surface.py checks the public surface against the reference's by name).
//! The files sit where the reference's modules sit (`params.rs`, `template.rs`, `helpers/args.rs`,
//! `helpers/glove.rs`).  Not carried over: `hnsw::disk` (private dead code in the reference),
//! `helpers::get_progress_bar` (returns an indicatif type), `helpers::data` (`split` has no caller in the
//! reference, `load_bf_data` reads the author's home directory).
pub mod ffi;
pub mod helpers;
pub mod params;
pub mod template;
