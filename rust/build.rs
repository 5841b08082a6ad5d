// Build script of the patched `zk_stark` crate: compiles the HIP kernels and the C ABI
// (stark_amd/csrc, include/stark_hip.h of this repository, vendored under hip/) for gfx950 and
// links them.  No CUDA shims and no second code path: without a gfx950 device the first call
// fails in stark_ctx_create and the crate panics.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let src = PathBuf::from(env::var("STARK_HIP_SRC").unwrap_or_else(|_| "hip/csrc".into()));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let mut objs = vec![];
    for f in ["api.hip", "ntt_kernels.hip", "hash_kernels.hip", "poly_kernels.hip"] {
        let o = out.join(f).with_extension("o");
        let ok = Command::new(&hipcc)
            .args(["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c"])
            .arg(src.join(f))
            .arg("-o")
            .arg(&o)
            .status()
            .expect("hipcc not found")
            .success();
        assert!(ok, "hipcc failed on {f}");
        objs.push(o);
    }
    let lib = out.join("libstark_hip.a");
    assert!(Command::new("ar").arg("crs").arg(&lib).args(&objs).status().unwrap().success());
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=static=stark_hip");
    println!("cargo:rustc-link-search=native=/opt/rocm/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=dylib=stdc++");
    println!("cargo:rerun-if-changed={}", src.display());
    println!("cargo:rerun-if-env-changed=HIPCC");
}
