//! `hnsw::helpers::args` (reference: hnsw/src/helpers/args.rs:3-47): the four positional-argument
//! readers of the reference's binaries, same names, same return types, same messages.  Written around
//! one generic field reader instead of four copies of the parsing code.
use std::env;
use std::str::FromStr;

/// `argv[idx]` parsed as `T`; a missing or malformed field panics with the reference's
/// `expect` text for that field ("Could not parse ...").
fn field<T: FromStr>(argv: &[String], idx: usize, what: &str) -> T {
    match argv.get(idx).map(|s| s.parse::<T>()) {
        Some(Ok(v)) => v,
        _ => panic!("Could not parse {what}"),
    }
}

fn argv_exactly(n_positional: usize, complaint: &'static str) -> Result<Vec<String>, &'static str> {
    let argv: Vec<String> = env::args().collect();
    if argv.len() == n_positional + 1 {
        Ok(argv)
    } else {
        Err(complaint)
    }
}

/// `dim lim` (args.rs:3-12)
pub fn parse_args_bf() -> Result<(usize, usize), &'static str> {
    let argv = argv_exactly(2, "Expected exactly 2 positional arguments.")?;
    Ok((field(&argv, 1, "dimention"), field(&argv, 2, "limit")))
}

/// `dim lim`, no count check: too few arguments panic (args.rs:14-19 indexes out of bounds there)
pub fn parse_args() -> (usize, usize) {
    let argv: Vec<String> = env::args().collect();
    (field(&argv, 1, "dimention"), field(&argv, 2, "limit"))
}

/// `lim m` (args.rs:21-31) -- imported by eval_glove/src/main.rs:8
pub fn parse_args_eval() -> Result<(usize, usize), &'static str> {
    let argv = argv_exactly(2, "Expected exactly 2 positional arguments.")?;
    Ok((field(&argv, 1, "limit"), field(&argv, 2, "M")))
}

/// `dim lim m ef_cons` (args.rs:33-47; the message says 3 although 4 are wanted, kept as is)
pub fn parse_args_eval_ef_cons() -> Result<(u32, usize, u8, u32), &'static str> {
    let argv = argv_exactly(4, "Expected exactly 3 positional arguments.")?;
    Ok((
        field(&argv, 1, "dimention"),
        field(&argv, 2, "limit"),
        field(&argv, 3, "M"),
        field(&argv, 4, "ef construction"),
    ))
}
