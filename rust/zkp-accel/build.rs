// Link against libzkp_accel.so.  ZKP_ACCEL_LIB_DIR = directory holding the library (ckb_zkp_amd/lib of the backend
// repository, or wherever it was installed); the HIP runtime it depends on lives in /opt/rocm/lib.
fn main() {
    let dir = std::env::var("ZKP_ACCEL_LIB_DIR").unwrap_or_else(|_| "/usr/local/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-search=native=/opt/rocm/lib");
    println!("cargo:rustc-link-lib=dylib=zkp_accel");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rustc-link-arg=-Wl,-rpath,/opt/rocm/lib");
    println!("cargo:rerun-if-env-changed=ZKP_ACCEL_LIB_DIR");
}
