// Links libfmx.so when the `gpu` feature is on.  FMX_LIB_DIR = the directory that holds libfmx.so
// (fm_index_amd/ of the fmx repository after `make -C fm_index_amd/csrc`).
fn main() {
    println!("cargo:rerun-if-env-changed=FMX_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_GPU").is_some() {
        if let Some(dir) = std::env::var_os("FMX_LIB_DIR") {
            let dir = dir.to_string_lossy();
            println!("cargo:rustc-link-search=native={dir}");
            println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        }
        println!("cargo:rustc-link-lib=dylib=fmx");
    }
}
