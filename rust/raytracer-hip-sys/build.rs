// Links librt_hip.so.  RT_HIP_LIB_DIR names the directory that holds it (the `raytracer_amd/` directory of the
// raytracer-amd checkout after `python -c "import __graft_entry__ as g; g.build()"`); the HIP runtime it depends on
// comes from ROCM_PATH (default /opt/rocm).  Nothing is compiled here: the library is built by hipcc for gfx950.
use std::env;

fn main() {
    let dir = env::var("RT_HIP_LIB_DIR").expect("set RT_HIP_LIB_DIR to the directory that holds librt_hip.so");
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{rocm}/lib");
    println!("cargo:rerun-if-env-changed=RT_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
