// SOURCE ONLY (never compiled here).  Point BBS_SIGN_AMD_DIR at the directory holding libbbs_sign_amd.so.
fn main() {
    if let Ok(dir) = std::env::var("BBS_SIGN_AMD_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=bbs_sign_amd");
}
