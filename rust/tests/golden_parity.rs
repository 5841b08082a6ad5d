//! First check on a machine that has cargo and a gfx950 device: the patched crate against the vectors of
//! the HIP repository (tests/golden/oracle_vectors.json, set STARK_GOLDEN to its path).  The vectors come
//! from a restatement of the reference, not from the reference itself (no Rust toolchain existed where
//! they were made), so this test is what pins the byte encodings: bincode(FieldElement), Merkle roots,
//! the Fri::commit transcript and the Fri::prove stream.  If an assertion fails on the ENCODING (the first
//! test, which touches no device code), the reference wins: fix the restatement and the leaf kernel and
//! regenerate the vectors.
use num_bigint::BigInt;
use serde_json::Value;
use sha2::{Digest, Sha256};
use zk_stark::modules::field::FieldElement;
use zk_stark::modules::fri::Fri;
use zk_stark::modules::merkle::Merkle;
use zk_stark::modules::ntt::NTT;
use zk_stark::modules::proof_stream::ProofStream;

fn golden() -> Value {
    let path = std::env::var("STARK_GOLDEN").unwrap_or_else(|_| "tests/golden/oracle_vectors.json".into());
    serde_json::from_str(&std::fs::read_to_string(path).expect("golden vectors not found")).unwrap()
}

/// packed {lo, hi} little-endian field vector (hex) -> elements
fn field_vec(hex_str: &str) -> Vec<FieldElement> {
    hex::decode(hex_str)
        .unwrap()
        .chunks_exact(16)
        .map(|c| FieldElement::new(BigInt::from_bytes_le(num_bigint::Sign::Plus, c)))
        .collect()
}

fn sha256_hex(bytes: &[u8]) -> String {
    hex::encode(Sha256::digest(bytes))
}

#[test]
fn leaf_encoding_is_bincode_of_field_element() {
    // fri.rs:119: the Merkle leaf of a codeword element.  Pure host code: this is the reference's own
    // serialisation against the layout the leaf kernel synthesises.
    let g = golden();
    for pair in g["leaf_bytes"].as_array().unwrap() {
        let v: BigInt = pair[0].as_str().unwrap().parse().unwrap();
        let want = pair[1].as_str().unwrap();
        assert_eq!(hex::encode(bincode::serialize(&FieldElement::new(v.clone())).unwrap()), want, "value {v}");
    }
    for pair in g["leaf_json"].as_array().unwrap() {
        let v: BigInt = pair[0].as_str().unwrap().parse().unwrap();
        assert_eq!(serde_json::to_string(&FieldElement::new(v)).unwrap(), pair[1].as_str().unwrap());
    }
}

/// The synthetic field vector of the vectors' generator: a SplitMix64 stream, two words per candidate
/// {lo, hi}, candidates not below p rejected.
fn splitmix_field_vector(n: usize, seed: u64) -> Vec<FieldElement> {
    let p = FieldElement::modulus();
    let mut state = seed;
    let mut next = || {
        state = state.wrapping_add(0x9E3779B97F4A7C15);
        let mut z = state;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
        z ^ (z >> 31)
    };
    let mut out = vec![];
    while out.len() < n {
        let lo = next();
        let hi = next();
        let v = (BigInt::from(hi) << 64) + BigInt::from(lo);
        if v < p {
            out.push(FieldElement::new(v));
        }
    }
    out
}

#[test]
fn ntt_matches_vectors() {
    // NTT::forward / NTT::inverse (ntt.rs:25-40) through the device against the stored transforms
    let g = golden();
    for n in ["8", "64", "1024"] {
        let e = &g["ntt"][n];
        let size: usize = n.parse().unwrap();
        let x = splitmix_field_vector(size, e["seed"].as_u64().unwrap());
        let ntt = NTT::new(size);
        let mut fwd = x.clone();
        ntt.forward(&mut fwd);
        assert!(fwd == field_vec(e["forward"].as_str().unwrap()), "forward at n = {n}");
        let mut inv = x.clone();
        ntt.inverse(&mut inv);
        assert!(inv == field_vec(e["inverse"].as_str().unwrap()), "inverse at n = {n}");
        ntt.inverse(&mut fwd);
        assert!(fwd == x, "round trip at n = {n}");
    }
}

#[test]
fn merkle_commit_matches_vectors() {
    let g = golden();
    for n in ["2", "8", "256"] {
        let e = &g["merkle_fe"][n];
        let elems = field_vec(e["elems"].as_str().unwrap());
        let leaves: Vec<Vec<u8>> = elems.iter().map(|x| bincode::serialize(x).unwrap()).collect();
        assert_eq!(hex::encode(Merkle::commit(&leaves)), e["root"].as_str().unwrap(), "root at n = {n}");
        let path: Vec<String> = Merkle::open(1, &leaves).iter().map(hex::encode).collect();
        let want: Vec<String> = e["open_1"].as_array().unwrap().iter().map(|s| s.as_str().unwrap().to_string()).collect();
        assert_eq!(path, want, "open(1) at n = {n}");
    }
    let e = &g["merkle_bytes"];
    let leaves: Vec<Vec<u8>> = e["leaves"].as_array().unwrap().iter().map(|s| hex::decode(s.as_str().unwrap()).unwrap()).collect();
    assert_eq!(hex::encode(Merkle::commit(&leaves)), e["root"].as_str().unwrap());
}

#[test]
fn fri_commit_and_prove_match_vectors() {
    // the parameters of the reference's own FRI test (fri.rs:428-504)
    let g = golden();
    let c = &g["fri_commit_ref_test"];
    let n0 = c["n0"].as_u64().unwrap() as usize;
    let fri = Fri::new(
        FieldElement::generator(),
        FieldElement::primitive_nth_root(n0 as u128),
        n0,
        c["expansion"].as_u64().unwrap() as usize,
        c["ncol"].as_u64().unwrap() as usize,
    );
    let codeword = field_vec(c["codeword"].as_str().unwrap());
    let mut ps = ProofStream::new();
    let codewords = fri.commit(codeword.clone(), &mut ps);
    let want: Vec<String> = c["objects"].as_array().unwrap().iter().map(|s| s.as_str().unwrap().to_string()).collect();
    assert_eq!(ps.objects, want, "Fri::commit transcript");
    let lens: Vec<usize> = codewords.iter().map(|cw| cw.len()).collect();
    let want_lens: Vec<usize> = c["layer_lens"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect();
    assert_eq!(lens, want_lens);
    assert!(*codewords.last().unwrap() == field_vec(c["last_codeword"].as_str().unwrap()));

    let p = &g["fri_prove_ref_test"];
    let mut ps = ProofStream::new();
    let indices = fri.prove(codeword, &mut ps);
    let want_idx: Vec<usize> = p["indices"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect();
    assert_eq!(indices, want_idx);
    assert_eq!(ps.objects.len() as u64, p["num_objects"].as_u64().unwrap());
    assert_eq!(sha256_hex(&ps.serialize()), p["serialized_sha256"].as_str().unwrap(), "Fri::prove stream");
    let mut verifier_stream = ProofStream::deserialize(&ps.serialize());
    let mut values = vec![];
    assert!(fri.verify(&mut verifier_stream, &mut values));
}
