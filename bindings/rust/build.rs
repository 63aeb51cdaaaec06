// build.rs — tells cargo where librtx.so lives.  Add `build = "build.rs"` to [package] in Cargo.toml
// (the reference's Cargo.toml:1-7 has no build script today).
//
//   RTX_LIB_DIR=/path/to/repo/rust-raytracer_amd cargo build --release
//
// librtx.so itself is built by `make -C rust-raytracer_amd/csrc` (hipcc --offload-arch=gfx950).
use std::env;

fn main() {
    let dir = env::var("RTX_LIB_DIR").unwrap_or_else(|_| "/usr/local/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtx");
    // run-time search path, so the binary finds librtx.so without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTX_LIB_DIR");
}
