// build.rs -- link the MI355X engine (libhnsw_mi355x.so, C ABI in include/hnsw_mi355x.h).
// HNSW_MI355X_LIB_DIR = the directory `python -m redis_hnsw_amd.build` wrote the library to.
fn main() {
    let dir = std::env::var("HNSW_MI355X_LIB_DIR")
        .expect("set HNSW_MI355X_LIB_DIR to the directory that holds libhnsw_mi355x.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=hnsw_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=HNSW_MI355X_LIB_DIR");
}
