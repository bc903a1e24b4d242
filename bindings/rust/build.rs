// Links libecfft_hip.so (built by `python -m ecfft_amd.build` in the MI355X repo).  ECFFT_HIP_DIR = directory holding it.
fn main() {
    let dir = std::env::var("ECFFT_HIP_DIR").unwrap_or_else(|_| "../../ecfft_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ecfft_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ECFFT_HIP_DIR");
}
