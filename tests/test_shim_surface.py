"""The Rust shim (shim-rust/, source only: no rustc in the image) must offer, by name, every public
item of the reference's `hnsw` crate that its callers can reach: every `pub fn` of
hnsw/src/template.rs, and every `use hnsw::...` path of eval_glove/src/main.rs and
hnsw/benches/hnsw_benchmarks.rs.  The lists below are data read off the reference (file:line in the
comments); when the reference tree is present (this container, not the GPU box) the test also checks
that the lists themselves are complete."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim-rust", "src")
REF = "/root/reference"

# hnsw/src/template.rs: save :43, load :75, new :133, len :146, distance :150, get_point :154,
# layer_degrees :158, insert_vec :165, get_layer :192, ann_by_vector :306, assert_param_compliance :341,
# print_index :372, insert_bulk :388, make_rand_index_full :623, make_rand_vectors :630
TEMPLATE_PUB_FNS = ["save", "load", "new", "len", "distance", "get_point", "layer_degrees", "insert_vec",
                    "get_layer", "ann_by_vector", "assert_param_compliance", "print_index", "insert_bulk",
                    "make_rand_index_full", "make_rand_vectors"]
# eval_glove/src/main.rs:8-11, hnsw/benches/hnsw_benchmarks.rs:2-3
CALLER_IMPORTS = ["hnsw::helpers::args::parse_args_eval", "hnsw::helpers::glove::load_glove_array",
                  "hnsw::template::HNSW"]
# hnsw/src/helpers/args.rs:3,14,21,33
ARGS_PUB_FNS = ["parse_args_bf", "parse_args", "parse_args_eval", "parse_args_eval_ef_cons"]


def read(name):
    return open(os.path.join(SHIM, name)).read()


def module_file(path):
    """source file of shim module `a::b::c` (crate-relative): the standard layout -- every module is declared
    `pub mod x;` by its parent (lib.rs, or the parent's mod.rs) and lives in x.rs or x/mod.rs beside it"""
    rel, decl = "", "lib.rs"
    for mod in path:
        assert re.search(r"pub mod %s;" % re.escape(mod), read(decl)), "module %s is not declared in %s" % ("::".join(path), decl)
        flat, nested = os.path.join(rel, mod + ".rs"), os.path.join(rel, mod, "mod.rs")
        if os.path.exists(os.path.join(SHIM, nested)):
            rel, decl = os.path.join(rel, mod), nested
        else:
            assert os.path.exists(os.path.join(SHIM, flat)), "%s not found" % flat
            decl = flat
    return decl


def pub_fns(src):
    return set(re.findall(r"pub fn\s+([A-Za-z_0-9]+)", src))


def test_every_pub_fn_of_template_rs_is_in_the_shim():
    have = pub_fns(read(module_file(["template"])))
    missing = [f for f in TEMPLATE_PUB_FNS if f not in have]
    assert not missing, "shim template lacks: %s" % missing


def test_every_caller_import_resolves():
    lib = read("lib.rs")
    for imp in CALLER_IMPORTS:
        parts = imp.split("::")[1:]
        item, mods = parts[-1], parts[:-1]
        src = read(module_file(mods))  # (asserts that every module on the path is declared by its parent)
        assert re.search(r"pub (fn|struct|type) %s\b" % item, src), "%s: `%s` not public in %s" % (
            imp, item, module_file(mods))


def test_args_parsers_and_signatures():
    src = read(module_file(["helpers", "args"]))
    assert set(ARGS_PUB_FNS) <= pub_fns(src)
    # return types as in the reference (args.rs:3,14,21,33)
    assert re.search(r"pub fn parse_args_eval\(\)\s*->\s*Result<\(usize, usize\), &'static str>", src)
    assert re.search(r"pub fn parse_args\(\)\s*->\s*\(usize, usize\)", src)
    assert re.search(r"pub fn parse_args_eval_ef_cons\(\)\s*->\s*Result<\(u32, usize, u8, u32\), &'static str>", src)
    glove = read(module_file(["helpers", "glove"]))
    # glove.rs:14-18
    assert re.search(r"pub fn load_glove_array\(\s*lim: usize,\s*file: File,\s*verbose: bool\s*\)\s*->\s*"
                     r"Result<\(Vec<String>, Vec<Vec<f32>>\)>", glove)


def test_hnsw_method_signatures_match_the_reference():
    src = read(module_file(["template"]))
    for sig in [r"pub fn new\(m: usize, ef_cons: Option<usize>, dim: usize\) -> Self",            # :133
                r"pub fn insert_bulk\(mut self, vectors: Vec<Vec<f32>>, nb_threads: usize, verbose: bool\) -> Result<HNSW, String>",  # :388
                r"pub fn insert_vec\(&mut self, vector: &Vec<f32>\) -> Result<NodeID, String>",     # :165
                r"pub fn ann_by_vector\(&self, vector: &Vec<f32>, n: usize, ef: usize\) -> Result<Vec<NodeID>, String>",  # :306
                r"pub fn save\(&self, dir: &Path\)",                                                # :43
                r"pub fn load\(dir: &Path\) -> Result<Self, String>",                               # :75
                r"pub fn distance\(&self, a: NodeID, b: NodeID\) -> Option<f32>",                   # :150
                r"pub fn layer_degrees\(&self, layer_nb: usize\)",                                  # :158
                r"pub fn print_index\(&self\)",                                                     # :372
                r"pub fn make_rand_index_full\(n: usize, dim: usize\) -> HNSW",                     # :623
                r"pub fn make_rand_vectors\(n: usize, dim: usize\) -> Vec<Vec<f32>>"]:              # :630
        assert re.search(sig, src), "signature not found: " + sig


def test_ffi_block_names_only_exported_symbols():
    """every extern "C" fn the shim declares is a symbol of include/hnsw_mi355x.h"""
    hdr = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    declared = set(re.findall(r"\b(hnsw_[a-z0-9_]+)\s*\(", hdr))
    used = set(re.findall(r"pub fn (hnsw_[a-z0-9_]+)\(", read("ffi.rs")))
    assert used and used <= declared, "not in the header: %s" % sorted(used - declared)


def test_lists_are_complete_against_the_reference_tree():
    if not os.path.isdir(REF):
        import pytest
        pytest.skip("reference tree not present (GPU box)")
    tmpl = open(os.path.join(REF, "hnsw/src/template.rs")).read()
    assert pub_fns(tmpl) == set(TEMPLATE_PUB_FNS)
    assert pub_fns(open(os.path.join(REF, "hnsw/src/helpers/args.rs")).read()) == set(ARGS_PUB_FNS)
    imports = set()
    for f in ("eval_glove/src/main.rs", "hnsw/benches/hnsw_benchmarks.rs"):
        for line in open(os.path.join(REF, f)):
            m = re.match(r"\s*use (hnsw::[A-Za-z_:0-9]+);", line)
            if m:
                imports.add(m.group(1))
    assert imports == set(CALLER_IMPORTS)
