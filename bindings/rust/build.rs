// Links liblbft_hip.so (built by `python -m librabft_simulator_amd.build`, hipcc --offload-arch=gfx950).
// LBFT_HIP_DIR = the directory that holds the library (default: ../librabft_simulator_amd next to this crate).
fn main() {
    let dir = std::env::var("LBFT_HIP_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../librabft_simulator_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=lbft_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=LBFT_HIP_DIR");
}
