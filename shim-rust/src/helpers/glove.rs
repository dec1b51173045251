//! `hnsw::helpers::glove` (reference: hnsw/src/helpers/glove.rs:14-71): the GloVe text loader the
//! reference's binaries and benches import (eval_glove/src/main.rs:9, hnsw_benchmarks.rs:2).
//!
//! Format: one embedding per line, `word v1 v2 ... vd`, single spaces.  Behaviour kept from the
//! reference: `lim == 0` reads every line; a token after the word that does not parse as `f32` is glued
//! onto the word (multi-token "words"); rows are checked against the length of the FIRST row from the
//! third row on (the reference's `embeddings.len() > 1` guard, glove.rs:57) and a mismatch panics with
//! the reference's message.  The progress display is a plain stderr counter (the shim has no
//! dependencies, so no indicatif bar).
//!
//! The reference's `brute_force_nns` takes `Arc<SimplePoints>` and an indicatif bar -- types of crates
//! the shim does not re-export; exact ground truth on the GPU is `HNSW::brute_force` in `template`.
use std::fs::File;
use std::io::{BufRead, BufReader, Result};

pub fn load_glove_array(lim: usize, file: File, verbose: bool) -> Result<(Vec<String>, Vec<Vec<f32>>)> {
    let mut words: Vec<String> = Vec::new();
    let mut rows: Vec<Vec<f32>> = Vec::new();
    for (line_no, line) in BufReader::new(file).lines().enumerate() {
        if lim != 0 && line_no >= lim {
            break;
        }
        let line = line?;
        let mut tokens = line.split(' ');
        let mut word = tokens.next().expect("Empty line").to_string();
        let mut vals: Vec<f32> = Vec::new();
        for tok in tokens {
            match tok.parse::<f32>() {
                Ok(x) => vals.push(x),
                Err(_) => word.push_str(tok),
            }
        }
        if rows.len() > 1 && rows[0].len() != vals.len() {
            panic!(
                "Line {0}: vector is not the same size as others. Len: {1}, Word {2}",
                line_no + 1,
                vals.len(),
                word
            );
        }
        rows.push(vals);
        words.push(word);
        if verbose && (line_no + 1) % 10_000 == 0 {
            eprint!("\rLoading Embeddings {}", line_no + 1);
        }
    }
    if verbose {
        eprintln!("\rLoading Embeddings {} done", rows.len());
    }
    Ok((words, rows))
}
