// Links libzkt_plonk_hip.so (built in-tree by `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    let dir = std::env::var("ZKT_PLONK_LIB_DIR").unwrap_or_else(|_| "../zkt-plonk_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkt_plonk_hip");
    println!("cargo:rerun-if-env-changed=ZKT_PLONK_LIB_DIR");
}
