// build.rs — link the MI355X engine (libgsv_engine.so: hipcc --offload-arch=gfx950 kernels + host runtime, include/gsv_engine.h).
fn main() {
    let dir = std::env::var("GSV_ENGINE_LIB_DIR").expect("set GSV_ENGINE_LIB_DIR to the directory that holds libgsv_engine.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=gsv_engine");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=GSV_ENGINE_LIB_DIR");
}
