fn main() {
    // HNSW_MI355X_LIB_DIR = directory holding libhnsw_mi355x.so (hnsw_rs_amd/ in this repository)
    if let Ok(dir) = std::env::var("HNSW_MI355X_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=hnsw_mi355x");
}
