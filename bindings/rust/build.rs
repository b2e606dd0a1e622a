// was (reference build.rs:1-3): println!("cargo:rustc-link-search=native=./mcl-rust/build/lib");
// BPP_AMD_LIB_DIR: the directory that holds libbpp_amd.so (bulletproofsplus_amd/ of this repository after
// `make -C bulletproofsplus_amd/csrc`).  Never run in this repository's build image (no Rust toolchain there).
fn main() {
    let dir = std::env::var("BPP_AMD_LIB_DIR").unwrap_or_else(|_| "../../bulletproofsplus_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=bpp_amd");
    println!("cargo:rerun-if-env-changed=BPP_AMD_LIB_DIR");
}
