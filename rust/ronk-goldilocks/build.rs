// Links libronk_ntt.so (the C ABI of include/ronk_ntt.h).  RONK_NTT_DIR = directory that holds the library
// (`<repo>/ronkathon_amd` after `make`); an rpath is added so `cargo test` finds it without LD_LIBRARY_PATH.
fn main() {
  let dir = std::env::var("RONK_NTT_DIR").expect("set RONK_NTT_DIR to the directory containing libronk_ntt.so");
  println!("cargo:rustc-link-search=native={dir}");
  println!("cargo:rustc-link-lib=dylib=ronk_ntt");
  println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
  println!("cargo:rerun-if-env-changed=RONK_NTT_DIR");
}
