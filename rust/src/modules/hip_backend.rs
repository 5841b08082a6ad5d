//! New bodies of the reference's hot-path functions over the C ABI (hip_ffi.rs).  New file of the patched
//! `zk_stark` crate: `src/modules/hip_backend.rs`.  The reference's methods keep their signatures and
//! delegate here with one line each (every function below names the method it is the body of); nothing else in the reference
//! changes.  Not compiled where it was written (no Rust toolchain there): tests/golden_parity.rs is the
//! first thing to run on a machine that has cargo.
#![allow(dead_code)]
use std::os::raw::c_void;

use generic_array::GenericArray;

use crate::modules::field::FieldElement;
use crate::modules::fri::Fri;
use crate::modules::hip_ffi::*;
use crate::modules::merkle::HashOutput;
use crate::modules::proof_stream::ProofStream;

const ONE: [u64; 2] = [1, 0];

// ---- ntt.rs ------------------------------------------------------------------------------------
/// NTT::forward / NTT::inverse (ntt.rs:25-40): in place, natural order; the n^-1 scaling of the inverse
/// is inside the library.  The caller keeps its `assert_eq!(len, self.n, "Input size must match NTT size")`.
pub fn ntt_in_place(data: &mut Vec<FieldElement>, inverse: bool) {
    let mut buf = pack(data);
    check(unsafe { stark_ntt(ctx(), buf.as_mut_ptr(), data.len(), inverse as i32) });
    *data = unpack(&buf);
}

/// NTT::evaluate (ntt.rs:101-107): zero-pad or TRUNCATE to n coefficients (lowest first), then forward.
pub fn ntt_evaluate(n: usize, coefficients: &[FieldElement]) -> Vec<FieldElement> {
    let used = coefficients.len().min(n);
    let coeffs = pack(&coefficients[..used]);
    let mut out = vec![0u64; 2 * n];
    check(unsafe {
        stark_coset_lde_batch(ctx(), coeffs.as_ptr(), used, used, 1, n, ONE.as_ptr(), out.as_mut_ptr(), n)
    });
    unpack(&out)
}

/// NTT::multiply (ntt.rs:110-135): transforms of length next_power_of_two(len_a + len_b - 1), pointwise
/// product, inverse, truncate.
pub fn ntt_multiply(a: &[FieldElement], b: &[FieldElement]) -> Vec<FieldElement> {
    if a.is_empty() || b.is_empty() {
        return vec![];
    }
    let result_len = a.len() + b.len() - 1;
    let n = result_len.next_power_of_two();
    let (pa, pb) = (pack(a), pack(b));
    let (mut fa, mut fb) = (vec![0u64; 2 * n], vec![0u64; 2 * n]);
    unsafe {
        check(stark_coset_lde_batch(ctx(), pa.as_ptr(), a.len(), a.len(), 1, n, ONE.as_ptr(), fa.as_mut_ptr(), n));
        check(stark_coset_lde_batch(ctx(), pb.as_ptr(), b.len(), b.len(), 1, n, ONE.as_ptr(), fb.as_mut_ptr(), n));
        check(stark_vec_mul(ctx(), fa.as_ptr(), fb.as_ptr(), fa.as_mut_ptr(), n));
        check(stark_ntt(ctx(), fa.as_mut_ptr(), n, 1));
    }
    let mut out = unpack(&fa);
    out.truncate(result_len);
    out
}

// ---- univariate_poly.rs ------------------------------------------------------------------------------
/// The fast path of Polynomial::eval_domain (univariate_poly.rs:44-54) for a domain
/// domain[i] = domain[0] * omega_n^i, n a power of two >= 8.  `coeffs` are the polynomial's, HIGHEST degree
/// first (:27).  Returns None when the domain has another shape or when Horner and the transform would
/// differ (the reference's NTT path truncates to n coefficients, :63, its Horner path does not): the
/// caller then runs the reference's own loop (:53).  One running product checks the domain instead of the
/// reference's n calls of pow (:180-194).
pub fn eval_domain_fast(coeffs: &[FieldElement], domain: &[FieldElement]) -> Option<Vec<FieldElement>> {
    let n = domain.len();
    if n < 8 || !n.is_power_of_two() || domain[0].is_zero() {
        return None;
    }
    let omega = FieldElement::primitive_nth_root(n as u128);
    let mut cur = domain[0].clone();
    for x in domain {
        if *x != cur {
            return None;
        }
        cur = cur * omega.clone();
    }
    let unshifted = domain[0] == FieldElement::one();
    if !unshifted && coeffs.len() > n {
        return None;
    }
    let mut low: Vec<FieldElement> = coeffs.iter().rev().cloned().collect();
    low.truncate(n); // only reached for the unshifted domain: the reference's resize (:63)
    let packed = pack(&low);
    let mut offset = [0u64; 2];
    pack_one(&domain[0], &mut offset);
    let mut out = vec![0u64; 2 * n];
    check(unsafe {
        stark_coset_lde_batch(ctx(), packed.as_ptr(), low.len(), low.len(), 1, n, offset.as_ptr(), out.as_mut_ptr(), n)
    });
    Some(unpack(&out))
}

/// Polynomial::lagrange_ntt (univariate_poly.rs:167-177): inverse transform, then reverse to highest-first;
/// exactly n coefficients, leading zeros kept.
pub fn lagrange_ntt(evaluations: &[FieldElement]) -> Vec<FieldElement> {
    let mut v = evaluations.to_vec();
    ntt_in_place(&mut v, true);
    v.reverse();
    v
}

// ---- merkle.rs -----------------------------------------------------------------------------------------
fn flatten(data_array: &[Vec<u8>]) -> (Vec<u8>, Vec<u64>) {
    let mut offsets = Vec::with_capacity(data_array.len() + 1);
    let mut flat = Vec::new();
    offsets.push(0u64);
    for d in data_array {
        flat.extend_from_slice(d);
        offsets.push(flat.len() as u64);
    }
    if flat.is_empty() {
        flat.push(0);
    }
    (flat, offsets)
}

/// Merkle::commit (merkle.rs:38-51) for arbitrary byte leaves.
pub fn merkle_commit(data_array: &Vec<Vec<u8>>) -> HashOutput {
    let (flat, offsets) = flatten(data_array);
    let mut root = [0u8; 32];
    check(unsafe {
        stark_merkle_commit_bytes(ctx(), flat.as_ptr(), offsets.as_ptr(), data_array.len(), root.as_mut_ptr(),
                                  std::ptr::null_mut())
    });
    GenericArray::clone_from_slice(&root)
}

/// Merkle::commit over the leaves bincode::serialize(&FieldElement) of a codeword (fri.rs:119,
/// stark.rs:302,358) without serialising on the host: the leaf bytes are formed in the hash kernel.
pub fn merkle_commit_field(codeword: &[FieldElement]) -> HashOutput {
    let packed = pack(codeword);
    let mut root = [0u8; 32];
    check(unsafe {
        stark_merkle_commit_fe(ctx(), packed.as_ptr(), codeword.len(), root.as_mut_ptr(), std::ptr::null_mut())
    });
    GenericArray::clone_from_slice(&root)
}

/// Merkle::open (merkle.rs:79-93): the tree is built once on the device and the path gathered from it.
pub fn merkle_open(index: usize, data_array: &Vec<Vec<u8>>) -> Vec<HashOutput> {
    let (flat, offsets) = flatten(data_array);
    let mut root = [0u8; 32];
    let mut tree: *mut StarkTree = std::ptr::null_mut();
    let mut path = vec![0u8; 32 * 64];
    let mut depth = 0usize;
    unsafe {
        check(stark_merkle_commit_bytes(ctx(), flat.as_ptr(), offsets.as_ptr(), data_array.len(), root.as_mut_ptr(),
                                        &mut tree));
        let rc = stark_merkle_open(tree, index, path.as_mut_ptr(), &mut depth);
        stark_tree_free(tree);
        check(rc);
    }
    path.chunks_exact(32).take(depth).map(GenericArray::clone_from_slice).collect()
}

// ---- fri.rs ----------------------------------------------------------------------------------------------
/// The host keeps the Fiat-Shamir transcript: hex::encode(root) is pushed (fri.rs:120) and, unless this is
/// the last round, alpha = FieldElement::sample(prover_fiat_shamir(32)) goes back to the device (:126-127).
extern "C" fn on_root(user: *mut c_void, _round: u32, last: i32, root: *const u8, alpha_out: *mut u64) -> i32 {
    let ps = unsafe { &mut *(user as *mut ProofStream) };
    let root = unsafe { std::slice::from_raw_parts(root, 32) };
    ps.push(hex::encode(root));
    if last == 0 {
        let alpha = FieldElement::sample(ps.prover_fiat_shamir(32));
        let out = unsafe { std::slice::from_raw_parts_mut(alpha_out, 2) };
        pack_one(&alpha, out);
    }
    0
}

/// Layers of one commit, resident on the device until dropped.
pub struct Layers(pub *mut StarkFriLayers);
impl Drop for Layers {
    fn drop(&mut self) {
        unsafe { stark_fri_layers_free(self.0) }
    }
}
impl Layers {
    pub fn count(&self) -> usize {
        unsafe { stark_fri_layers_count(self.0) }
    }
    pub fn len(&self, layer: usize) -> usize {
        unsafe { stark_fri_layer_len(self.0, layer) }
    }
    pub fn download(&self, layer: usize) -> Vec<FieldElement> {
        let mut buf = vec![0u64; 2 * self.len(layer)];
        check(unsafe { stark_fri_layer_download(self.0, layer, buf.as_mut_ptr()) });
        unpack(&buf)
    }
}

/// The commit loop of Fri::commit (fri.rs:112-143) on the device; the stream receives the roots.  The last
/// codeword is pushed as JSON by the caller side below (:146).
pub fn fri_commit_resident(fri: &Fri, codeword: &[FieldElement], proof_stream: &mut ProofStream) -> Layers {
    let packed = pack(codeword);
    let (mut offset, mut omega) = ([0u64; 2], [0u64; 2]);
    pack_one(&fri.offset, &mut offset);
    pack_one(&fri.omega, &mut omega);
    let mut layers: *mut StarkFriLayers = std::ptr::null_mut();
    check(unsafe {
        stark_fri_commit(ctx(), packed.as_ptr(), codeword.len(), offset.as_ptr(), omega.as_ptr(), fri.expansion_factor,
                         fri.num_colinearity_tests, on_root, proof_stream as *mut ProofStream as *mut c_void,
                         &mut layers)
    });
    let l = Layers(layers);
    let last = l.download(l.count() - 1);
    proof_stream.push(serde_json::to_string(&last).unwrap()); // fri.rs:146
    l
}

/// Fri::commit (fri.rs:100-152) with its by-value signature: every codeword comes back to the host.
pub fn fri_commit(fri: &Fri, codeword: Vec<FieldElement>, proof_stream: &mut ProofStream) -> Vec<Vec<FieldElement>> {
    let l = fri_commit_resident(fri, &codeword, proof_stream);
    (0..l.count()).map(|r| l.download(r)).collect()
}

/// One round of Fri::query (fri.rs:155-209) answered from the resident layers in one round trip: the
/// three values and three authentication paths of every colinearity test.
#[allow(dead_code)]
fn fri_query_resident(fri: &Fri, l: &Layers, layer: usize, c_indices: &[usize], proof_stream: &mut ProofStream) -> Vec<usize> {
    let t = fri.num_colinearity_tests;
    let cs: Vec<u64> = c_indices.iter().take(t).map(|&i| i as u64).collect();
    let half = l.len(layer) / 2;
    let mut values = vec![0u64; 2 * 3 * t];
    let mut paths = vec![0u8; 32 * 64 * 3 * t];
    let (mut depth, mut next_depth) = (0usize, 0usize);
    check(unsafe {
        stark_fri_layer_query(l.0, layer, cs.as_ptr(), t, values.as_mut_ptr(), paths.as_mut_ptr(), &mut depth,
                              &mut next_depth)
    });
    let v = unpack(&values); // [all a | all b | all c]
    for s in 0..t {
        proof_stream.push(
            serde_json::to_string(&(v[s].to_string(), v[t + s].to_string(), v[2 * t + s].to_string())).unwrap(),
        ); // fri.rs:166-179
    }
    let path_at = |base: usize, k: usize, d: usize| -> Vec<HashOutput> {
        paths[32 * (base + k * d)..32 * (base + (k + 1) * d)].chunks_exact(32).map(GenericArray::clone_from_slice).collect()
    };
    for s in 0..t {
        // fri.rs:182-204: a, b, c paths of test s
        proof_stream.push(serde_json::to_string(&path_at(0, s, depth)).unwrap());
        proof_stream.push(serde_json::to_string(&path_at(t * depth, s, depth)).unwrap());
        proof_stream.push(serde_json::to_string(&path_at(2 * t * depth, s, next_depth)).unwrap());
    }
    let mut out: Vec<usize> = cs.iter().map(|&c| c as usize).collect();
    out.extend(cs.iter().map(|&c| c as usize + half));
    out
}

/// Fri::prove (fri.rs:212-254): commit on the device, queries from the retained codewords and trees.
/// Only the first num_colinearity_tests indices matter from round to round: the list the reference passes
/// on doubles every round (:160-163,207-208) and only its first entries are ever read (:166-203), so round
/// i asks for top[s] mod len(i)/2 and the reads of all rounds come back in ONE round trip
/// (stark_fri_query_all); fri_query_resident above is the same thing one round at a time.
pub fn fri_prove(fri: &Fri, codeword: Vec<FieldElement>, proof_stream: &mut ProofStream) -> Vec<usize> {
    let l = fri_commit_resident(fri, &codeword, proof_stream);
    let layers = l.count();
    let t = fri.num_colinearity_tests;
    let top = fri.sample_indices(&proof_stream.prover_fiat_shamir(32), l.len(0) / 2, l.len(layers - 1), t);
    if layers > 1 && t > 0 {
        let depth: Vec<usize> = (0..layers).map(|i| l.len(i).trailing_zeros() as usize).collect();
        let digests: usize = (0..layers - 1).map(|i| t * (2 * depth[i] + depth[i + 1])).sum();
        let top64: Vec<u64> = top.iter().take(t).map(|&i| i as u64).collect();
        let mut values = vec![0u64; 2 * 3 * t * (layers - 1)];
        let mut paths = vec![0u8; 32 * digests + 32];
        check(unsafe { stark_fri_query_all(l.0, top64.as_ptr(), t, values.as_mut_ptr(), paths.as_mut_ptr()) });
        let v = unpack(&values); // per round [all a | all b | all c]
        let mut at = 0usize;
        for i in 0..layers - 1 {
            let e = &v[i * 3 * t..(i + 1) * 3 * t];
            for s in 0..t {
                proof_stream.push(
                    serde_json::to_string(&(e[s].to_string(), e[t + s].to_string(), e[2 * t + s].to_string())).unwrap(),
                ); // fri.rs:166-179
            }
            let path_at = |base: usize, k: usize, d: usize| -> Vec<HashOutput> {
                paths[32 * (at + base + k * d)..32 * (at + base + (k + 1) * d)]
                    .chunks_exact(32)
                    .map(GenericArray::clone_from_slice)
                    .collect()
            };
            for s in 0..t {
                // fri.rs:182-204: a, b, c paths of test s
                proof_stream.push(serde_json::to_string(&path_at(0, s, depth[i])).unwrap());
                proof_stream.push(serde_json::to_string(&path_at(t * depth[i], s, depth[i])).unwrap());
                proof_stream.push(serde_json::to_string(&path_at(2 * t * depth[i], s, depth[i + 1])).unwrap());
            }
            at += t * (2 * depth[i] + depth[i + 1]);
        }
    }
    let mut out = top.clone();
    out.extend(top.iter().map(|idx| idx + l.len(0) / 2));
    out
}
