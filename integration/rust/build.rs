// build.rs — link the MI355X library (UNTESTED: no Rust toolchain in the build image).
// RTX_LIB_DIR = directory holding librtx_hip.so (the repo's ray-tracer_amd/).
fn main() {
    let dir = std::env::var("RTX_LIB_DIR").unwrap_or_else(|_| "ray-tracer_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtx_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTX_LIB_DIR");
}
