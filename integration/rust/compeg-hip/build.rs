// Links against the library built by `make -C compeg_amd/csrc`.
// COMPEG_HIP_LIB_DIR overrides the directory that holds libcompeg_hip.so.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("COMPEG_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        // integration/rust/compeg-hip -> repo root -> compeg_amd
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../compeg_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=compeg_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=COMPEG_HIP_LIB_DIR");
}
