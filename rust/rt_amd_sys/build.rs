// Builds librt_amd.so with hipcc (through the Makefile next to the kernels) and links it.
//
// Written without a Rust toolchain at hand (none is installed where the kernels were
// developed): it has not been compiled.  It only shells out to `make` and prints cargo
// directives, so there is little to get wrong, but treat it as a reviewed sketch.
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    // <repo>/rust/rt_amd_sys -> <repo>/raytracer-rust_amd
    let native = manifest.join("..").join("..").join("raytracer-rust_amd");
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    let arch = env::var("RT_AMD_ARCH").unwrap_or_else(|_| "gfx950".to_string());

    let status = Command::new("make")
        .arg("-C").arg(&native)
        .arg(format!("ROCM={rocm}"))
        .arg(format!("ARCH={arch}"))
        .arg("librt_amd.so")
        .status()
        .expect("failed to run make (is hipcc installed?)");
    assert!(status.success(), "building librt_amd.so failed");

    println!("cargo:rustc-link-search=native={}", native.display());
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", native.display());
    println!("cargo:rerun-if-changed={}", native.join("csrc").display());
    println!("cargo:rerun-if-changed={}", manifest.join("..").join("..").join("include").join("rt_amd.h").display());
}
