// Link against libfrw.so (built by `make -C falcon-r1cs_amd/csrc`); FRW_LIB_DIR names the directory that holds it.
fn main() {
    if let Ok(dir) = std::env::var("FRW_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=frw");
    println!("cargo:rerun-if-env-changed=FRW_LIB_DIR");
}
