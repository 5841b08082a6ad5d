//! FFI declarations of the MI355X hot path (include/stark_hip.h), the process-wide context and the
//! packing helpers.  New file of the patched `zk_stark` crate: `src/modules/hip_ffi.rs`, declared in
//! src/lib.rs next to the reference's nine modules (src/lib.rs:1-11).
//!
//! Every function of the C header is declared here, one to one; tests/test_rust_ffi_cpu.py of the HIP
//! repository checks names and argument counts against the header.
#![allow(dead_code)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_void};
use std::sync::OnceLock;

use num_bigint::{BigInt, Sign};

use crate::modules::field::FieldElement;

#[repr(C)]
pub struct StarkCtx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct StarkTree {
    _p: [u8; 0],
}
#[repr(C)]
pub struct StarkFriLayers {
    _p: [u8; 0],
}
/// stark_transcript_cb: called once per FRI round with the layer's Merkle root; returns alpha unless `last`.
pub type TranscriptCb =
    extern "C" fn(user: *mut c_void, round: u32, last: i32, root: *const u8, alpha_out: *mut u64) -> i32;

extern "C" {
    pub fn stark_ctx_create(device: i32, out: *mut *mut StarkCtx) -> i32;
    pub fn stark_ctx_destroy(ctx: *mut StarkCtx);
    pub fn stark_ctx_set_stream(ctx: *mut StarkCtx, hip_stream: *mut c_void) -> i32;
    pub fn stark_ctx_reset_stream(ctx: *mut StarkCtx) -> i32;
    pub fn stark_ctx_synchronize(ctx: *mut StarkCtx) -> i32;
    pub fn stark_ctx_trim(ctx: *mut StarkCtx) -> i32;
    pub fn stark_last_error() -> *const c_char;
    pub fn stark_dev_alloc(ctx: *mut StarkCtx, bytes: usize, dptr: *mut *mut c_void) -> i32;
    pub fn stark_dev_free(ctx: *mut StarkCtx, dptr: *mut c_void) -> i32;
    pub fn stark_copy_h2d(ctx: *mut StarkCtx, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> i32;
    pub fn stark_copy_d2h(ctx: *mut StarkCtx, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> i32;
    pub fn stark_copy_d2d(ctx: *mut StarkCtx, dst_dev: *mut c_void, src_dev: *const c_void, bytes: usize) -> i32;
    pub fn stark_ntt(ctx: *mut StarkCtx, data: *mut u64, n: usize, inverse: i32) -> i32;
    pub fn stark_ntt_dev(ctx: *mut StarkCtx, data_dev: *mut u64, n: usize, inverse: i32) -> i32;
    pub fn stark_ntt_batch(ctx: *mut StarkCtx, data: *mut u64, n: usize, n_cols: usize, col_stride: usize,
                           inverse: i32) -> i32;
    pub fn stark_ntt_batch_dev(ctx: *mut StarkCtx, data_dev: *mut u64, n: usize, n_cols: usize, col_stride: usize,
                               inverse: i32) -> i32;
    pub fn stark_coset_lde_batch(ctx: *mut StarkCtx, coeffs: *const u64, n_coeffs: usize, coeff_stride: usize,
                                 n_cols: usize, n_out: usize, offset: *const u64, out: *mut u64, out_stride: usize) -> i32;
    pub fn stark_coset_lde_batch_dev(ctx: *mut StarkCtx, coeffs_dev: *const u64, n_coeffs: usize,
                                     coeff_stride: usize, n_cols: usize, n_out: usize, offset: *const u64,
                                     out_dev: *mut u64, out_stride: usize) -> i32;
    pub fn stark_vec_mul(ctx: *mut StarkCtx, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> i32;
    pub fn stark_vec_mul_dev(ctx: *mut StarkCtx, a_dev: *const u64, b_dev: *const u64, out_dev: *mut u64, n: usize) -> i32;
    pub fn stark_vec_read_dev(ctx: *mut StarkCtx, vec_dev: *const u64, n: usize, indices: *const u64, count: usize,
                              out: *mut u64) -> i32;
    pub fn stark_poly_compose_linear_dev(ctx: *mut StarkCtx, coeffs_dev: *mut u64, n: usize, n_cols: usize,
                                         col_stride: usize, s: *const u64) -> i32;
    pub fn stark_coset_interpolate_batch_dev(ctx: *mut StarkCtx, data_dev: *mut u64, n: usize, n_cols: usize,
                                             col_stride: usize, offset: *const u64) -> i32;
    pub fn stark_vec_inv_dev(ctx: *mut StarkCtx, a_dev: *const u64, out_dev: *mut u64, n: usize) -> i32;
    pub fn stark_poly_lincomb_dev(ctx: *mut StarkCtx, n_terms: usize, srcs_dev: *const *const u64,
                                  lens: *const usize, shifts: *const usize, weights: *const u64, out_dev: *mut u64,
                                  n_out: usize) -> i32;
    pub fn stark_mpoly_eval_dev(ctx: *mut StarkCtx, n_vars: usize, cols_dev: *const *const u64,
                                rotations: *const usize, n_terms: usize, coeffs: *const u64, exps: *const u8,
                                mul_dev: *const u64, out_dev: *mut u64, n: usize) -> i32;
    pub fn stark_poly_degree_dev(ctx: *mut StarkCtx, coeffs_dev: *const u64, n: usize, degree_out: *mut usize,
                                 is_zero_out: *mut i32) -> i32;
    pub fn stark_poly_degree_post_dev(ctx: *mut StarkCtx, coeffs_dev: *const u64, n: usize, result_dev: *mut u64) -> i32;
    pub fn stark_poly_eval_points_dev(ctx: *mut StarkCtx, coeffs_dev: *const u64, n: usize, n_polys: usize, stride: usize,
                                      points: *const u64, n_points: usize, out: *mut u64) -> i32;
    pub fn stark_sample_bytes_dev(ctx: *mut StarkCtx, bytes_dev: *const u8, width: usize, n: usize, reversed: i32,
                                  out_dev: *mut u64) -> i32;
    pub fn stark_fourstep_twiddle_dev(ctx: *mut StarkCtx, data_dev: *mut u64, n_total: usize, row_start: usize,
                                      n_rows: usize, n_cols: usize, inverse: i32) -> i32;
    pub fn stark_fourstep_pack_dev(ctx: *mut StarkCtx, src_dev: *const u64, n_rows: usize, n_cols: usize,
                                   world: usize, n_total: usize, row_start: usize, twiddle: i32, dst_dev: *mut u64) -> i32;
    pub fn stark_fourstep_unpack_dev(ctx: *mut StarkCtx, src_dev: *const u64, world: usize, n_rows: usize,
                                     run: usize, dst_dev: *mut u64) -> i32;
    pub fn stark_merkle_commit_fe(ctx: *mut StarkCtx, elems: *const u64, n: usize, root: *mut u8,
                                  tree_out: *mut *mut StarkTree) -> i32;
    pub fn stark_merkle_commit_fe_dev(ctx: *mut StarkCtx, elems_dev: *const u64, n: usize, root: *mut u8,
                                      tree_out: *mut *mut StarkTree) -> i32;
    pub fn stark_merkle_roots_fe_batch_dev(ctx: *mut StarkCtx, elems_dev: *const u64, n: usize, n_cols: usize,
                                           col_stride: usize, roots_out: *mut u8) -> i32;
    pub fn stark_merkle_root_fe_static_dev(ctx: *mut StarkCtx, elems_dev: *const u64, n: usize,
                                           root_dev_out: *mut u8) -> i32;
    pub fn stark_merkle_subtree_roots_dev(ctx: *mut StarkCtx, elems_dev: *const u64, chunk_len: usize,
                                          n_chunks: usize, roots_dev_out: *mut u8, forest_out: *mut *mut StarkTree) -> i32;
    pub fn stark_merkle_root_of_digests_dev(ctx: *mut StarkCtx, digests_dev: *const u8, m: usize, root: *mut u8,
                                            tree_out: *mut *mut StarkTree) -> i32;
    pub fn stark_merkle_commit_bytes(ctx: *mut StarkCtx, flat: *const u8, offsets: *const u64, n: usize,
                                     root: *mut u8, tree_out: *mut *mut StarkTree) -> i32;
    pub fn stark_merkle_open(tree: *mut StarkTree, index: usize, path_out: *mut u8, depth_out: *mut usize) -> i32;
    pub fn stark_merkle_open_many(tree: *mut StarkTree, indices: *const u64, count: usize, paths_out: *mut u8,
                                  depth_out: *mut usize) -> i32;
    pub fn stark_tree_num_leaves(tree: *const StarkTree) -> usize;
    pub fn stark_tree_download(tree: *mut StarkTree, out: *mut u8) -> i32;
    pub fn stark_tree_free(tree: *mut StarkTree);
    pub fn stark_fri_fold(ctx: *mut StarkCtx, codeword: *const u64, N: usize, alpha: *const u64,
                          offset: *const u64, omega: *const u64, out: *mut u64) -> i32;
    pub fn stark_fri_fold_dev(ctx: *mut StarkCtx, codeword_dev: *const u64, N: usize, alpha: *const u64,
                              offset: *const u64, omega: *const u64, out_dev: *mut u64) -> i32;
    pub fn stark_fri_fold_shard_dev(ctx: *mut StarkCtx, local_dev: *const u64, local_len: usize, alpha: *const u64,
                                    offset: *const u64, omega: *const u64, chunk_len: usize, period: usize,
                                    base: usize, out_dev: *mut u64, chunk_roots_dev_out: *mut u8,
                                    forest_out: *mut *mut StarkTree) -> i32;
    pub fn stark_fri_num_rounds(domain_length: usize, expansion_factor: usize, num_colinearity_tests: usize) -> usize;
    pub fn stark_fri_commit(ctx: *mut StarkCtx, codeword: *const u64, N: usize, offset: *const u64,
                            omega: *const u64, expansion_factor: usize, num_colinearity_tests: usize,
                            cb: TranscriptCb, user: *mut c_void, layers_out: *mut *mut StarkFriLayers) -> i32;
    pub fn stark_fri_commit_dev(ctx: *mut StarkCtx, codeword_dev: *const u64, N: usize, offset: *const u64,
                                omega: *const u64, expansion_factor: usize, num_colinearity_tests: usize,
                                cb: TranscriptCb, user: *mut c_void, layers_out: *mut *mut StarkFriLayers) -> i32;
    pub fn stark_fri_layers_count(l: *const StarkFriLayers) -> usize;
    pub fn stark_fri_layer_len(l: *const StarkFriLayers, layer: usize) -> usize;
    pub fn stark_fri_layer_root(l: *const StarkFriLayers, layer: usize, root: *mut u8) -> i32;
    pub fn stark_fri_layer_download(l: *mut StarkFriLayers, layer: usize, out: *mut u64) -> i32;
    pub fn stark_fri_layer_read(l: *mut StarkFriLayers, layer: usize, indices: *const u64, count: usize,
                                out: *mut u64) -> i32;
    pub fn stark_fri_layer_open(l: *mut StarkFriLayers, layer: usize, index: usize, path_out: *mut u8,
                                depth_out: *mut usize) -> i32;
    pub fn stark_fri_layer_open_many(l: *mut StarkFriLayers, layer: usize, indices: *const u64, count: usize,
                                     paths_out: *mut u8, depth_out: *mut usize) -> i32;
    pub fn stark_fri_layer_query(l: *mut StarkFriLayers, layer: usize, c_indices: *const u64, count: usize,
                                 values_out: *mut u64, paths_out: *mut u8, depth_out: *mut usize,
                                 next_depth_out: *mut usize) -> i32;
    pub fn stark_fri_query_all(l: *mut StarkFriLayers, top_indices: *const u64, count: usize, values_out: *mut u64,
                               paths_out: *mut u8) -> i32;
    pub fn stark_fri_layers_free(l: *mut StarkFriLayers);
}

/// One context per process (the reference has no state).  `cargo test` runs tests on several threads; the
/// library serialises the calls on one context itself and makes the context's device current on the calling
/// thread, so sharing the pointer is sound.
pub struct CtxPtr(pub *mut StarkCtx);
unsafe impl Sync for CtxPtr {}
unsafe impl Send for CtxPtr {}
static CTX: OnceLock<CtxPtr> = OnceLock::new();

pub fn ctx() -> *mut StarkCtx {
    CTX.get_or_init(|| {
        let device = std::env::var("STARK_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut p: *mut StarkCtx = std::ptr::null_mut();
        check(unsafe { stark_ctx_create(device, &mut p) });
        CtxPtr(p)
    })
    .0
}

/// A non-zero return code becomes the panic the reference raises at the same place: the library's
/// message is the reference's panic text (ntt.rs:15,16,26,32; merkle.rs:18,56; fri.rs:34,116).
pub fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(stark_last_error()) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}

/// FieldElement.value (a canonical BigInt, field.rs:11-26) -> two little-endian u64 words.
pub fn pack_one(x: &FieldElement, out: &mut [u64]) {
    let (_, digits) = x.value.to_u64_digits();
    out[0] = *digits.first().unwrap_or(&0);
    out[1] = *digits.get(1).unwrap_or(&0);
}
pub fn pack(v: &[FieldElement]) -> Vec<u64> {
    let mut out = vec![0u64; 2 * v.len()];
    for (i, x) in v.iter().enumerate() {
        pack_one(x, &mut out[2 * i..2 * i + 2]);
    }
    out
}
pub fn unpack_one(w: &[u64]) -> FieldElement {
    let digits = [w[0] as u32, (w[0] >> 32) as u32, w[1] as u32, (w[1] >> 32) as u32];
    FieldElement { value: BigInt::from_slice(Sign::Plus, &digits) }
}
pub fn unpack(p: &[u64]) -> Vec<FieldElement> {
    p.chunks_exact(2).map(unpack_one).collect()
}
