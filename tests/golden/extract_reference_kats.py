"""Collects the known-answer DATA the reference's own unit tests hold for the
field core and writes tests/golden/reference_kats.json.

Run in the authoring container only (needs /root/reference).  Only numeric
constants and expected values are extracted -- no source text is kept.
Sources: rescue_prime.rs:27-169 (parameters), :415-422 (hash vectors),
stark.rs:65 (128-th root), field.rs:29,32 (generator, modulus),
multivariate_poly.rs:277-296 (Lagrange coefficients), field.rs:163-209,250-257
(small identities, restated as triples).
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
rp = open(os.path.join(REF, "src/modules/rescue_prime.rs")).read()
new_body = rp[rp.index("pub fn new()"):rp.index("// hash")]

nums = [int(x) for x in re.findall(r"\b(\d{6,})\b", new_body)]
alpha_inv = int(re.search(r"alpha_inv: BigInt = BigInt::from\((\d+)", new_body).group(1))
mds_block = new_body[new_body.index("let MDS:"):new_body.index("// MDS inverse")]
mds = [int(x) for x in re.findall(r"BigInt::from\((\d+)", mds_block)]
mdsi_block = new_body[new_body.index("let MDS_inv:"):new_body.index("// pre field element")]
mds_inv = [int(x) for x in re.findall(r"BigInt::from\((\d+)", mdsi_block)]
rc_block = new_body[new_body.index("round_constant_values"):new_body.index("// convert round constant")]
rcs = [int(x) for x in re.findall(r"^\s*(\d+),?\s*$", rc_block, flags=re.M)]
assert len(rcs) == 108, len(rcs)
assert len(mds) == 4 and len(mds_inv) == 4

tests = rp[rp.index("fn test_rescue_prime"):rp.index("fn test_trace")]
vec = [int(x) for x in re.findall(r"BigInt::from\((\d+) as u128", tests)]
# (one -> out0), (in1 -> out1)
hash_kats = [[1, vec[0]], [vec[1], vec[2]]]

stark = open(os.path.join(REF, "src/modules/stark.rs")).read()
omicron = int(re.search(r"omicron\.value == BigInt::from\((\d+)", stark).group(1)) \
    if re.search(r"omicron\.value == BigInt::from\((\d+)", stark) else \
    int(re.search(r"(65907963977709178563567092354521124432)", stark).group(1))

field = open(os.path.join(REF, "src/modules/field.rs")).read()
generator = int(re.search(r"BigInt::from\((\d+) as i128\)", field).group(1))

mp = open(os.path.join(REF, "src/modules/multivariate_poly.rs")).read()
lift = mp[mp.index("fn test_lift"):]
lift_coeffs = [int(x) for x in re.findall(r"coeff_\d\.value, BigInt::from\((\d+)", lift)]

out = {
    "_source": "numeric constants and expected values held by the reference's unit tests",
    "modulus_formula": "407*2^119+1",
    "generator": str(generator),
    "omicron_128th_root": str(omicron),
    "rescue_prime": {
        "m": 2, "N": 27, "alpha": 3, "alpha_inv": str(alpha_inv),
        "MDS": [str(x) for x in mds],
        "MDS_inv": [str(x) for x in mds_inv],
        "round_constants": [str(x) for x in rcs],
        "hash_kats": [[str(a), str(b)] for a, b in hash_kats],
    },
    "lagrange_lift": {"x": ["0", "1", "2"], "y": ["2", "5", "5"],
                      "coeffs_low_first": [str(c) for c in lift_coeffs]},
    "field_identities": [
        {"op": "add", "a": "p-1", "b": "10", "out": "9"},
        {"op": "sub", "a": "p-1", "b": "10", "out": "p-11"},
        {"op": "sub", "a": "2", "b": "3", "out": "p-1"},
        {"op": "pow", "a": "2", "b": "32", "out": str(2 ** 32)},
    ],
}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, "rcs", len(rcs))
