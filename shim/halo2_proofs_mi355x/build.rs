// build.rs of the patched halo2_proofs: link the MI355X backend (include/zkmi355.h).
fn main() {
    println!("cargo:rerun-if-env-changed=ZKMI355_LIB_DIR");
    if let Ok(dir) = std::env::var("ZKMI355_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=zkmi355");
}
