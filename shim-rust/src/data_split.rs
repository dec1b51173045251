//! `hnsw::helpers::data::split` (reference: hnsw/src/helpers/data.rs:6-33): `0..nb_elements` cut into
//! `nb_splits` consecutive runs of `nb_elements / nb_splits`, the last run taking the remainder.
//! (`load_bf_data` of the same reference file reads JSON from the author's home directory and is not
//! carried over.)
pub fn split(nb_elements: usize, nb_splits: usize) -> Vec<Vec<usize>> {
    let each = nb_elements / nb_splits;
    let runs: Vec<Vec<usize>> = (0..nb_splits)
        .map(|k| {
            let lo = k * each;
            let hi = if k + 1 == nb_splits { nb_elements } else { lo + each };
            (lo..hi).collect()
        })
        .collect();
    let total: usize = runs.iter().map(|r| r.len()).sum();
    assert!(total == nb_elements, "Total elements: {nb_elements}, sum of splits: {total}");
    runs
}
