// Points the linker at libavian_mi355x.so: AVIAN_MI355X_LIB_DIR, or the in-tree build directory of this repository.
fn main() {
    let dir = std::env::var("AVIAN_MI355X_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(env!("CARGO_MANIFEST_DIR"));
        here.join("../../../avian_amd/csrc").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=avian_mi355x");
    println!("cargo:rerun-if-env-changed=AVIAN_MI355X_LIB_DIR");
}
