"""Writes tests/golden/scaled_stark.json: proofs of the Rescue-Prime chain at several trace
lengths from the CPU restatement oracle/stark_scaled_oracle.py (length, SHA-256 and the claimed
output of each proof; the proofs themselves are 60-200 KB and are regenerated, not stored).

    python tests/golden/gen_scaled_stark.py 6 8 10 12 14 16      # log2 of the trace lengths
    python tests/golden/gen_scaled_stark.py rows=100,e=8,c=3,in=7   # any length and parameters

Parameters of a plain case: expansion factor 4, 2 colinearity tests, input element 12345, random
bytes = SHAKE256("scaled-stark-<key>") read as one stream (stream_rng).  A case given as
rows=..,e=..,c=..,in=.. sets the trace length (any number of rows from 2), the expansion factor,
the number of colinearity tests and the input element, and is stored under that string."""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import stark_prover_oracle as sp      # noqa: E402
import stark_scaled_oracle as ss      # noqa: E402

INPUT = 12345
EXPANSION, NCOL, SECURITY = 4, 2, 2


def case(key):
    rp = sp.RescuePrime()
    expansion, ncol, inp = EXPANSION, NCOL, INPUT
    if "=" in key:
        kv = dict(item.split("=") for item in key.split(","))
        rows, expansion, ncol, inp = int(kv["rows"]), int(kv.get("e", EXPANSION)), int(kv.get("c", NCOL)), int(kv.get("in", INPUT))
        log_rows = None
    else:
        log_rows = int(key)
        rows = 1 << log_rows
    return _case(rp, key, log_rows, rows, expansion, ncol, inp)


def _case(rp, key, log_rows, rows, EXPANSION, NCOL, INPUT):
    st = ss.ScaledStark(EXPANSION, NCOL, SECURITY, rp.m, rows)
    trace = ss.rescue_chain_trace(rp, INPUT, rows)
    air = ss.rescue_column_air(rp)
    cols = ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
    n_random = 17 * (st.num_randomizers * rp.m + st.max_degree(air) + 1)
    seed = b"scaled-stark-" + key.encode()
    proof = st.prove(trace, air, cols, boundary, ss.stream_rng(seed, n_random))
    # stark.rs:44-58 sizes the domains for constraints of degree 2; where the Rescue-Prime constraints (degree 3)
    # push max_degree past fri_domain / expansion the reference's own verifier rejects its prover's proof
    verifies = st.verify(proof, air, cols, boundary)
    assert verifies == (st.max_degree(air) < st.fri_domain_length // EXPANSION)
    return {"log_rows": log_rows, "rows": rows, "verifies": verifies, "input": str(INPUT), "output": str(trace[-1][0]), "seed": seed.decode(),
            "random_bytes": n_random, "expansion": EXPANSION, "ncol": NCOL, "security": SECURITY,
            "fri_domain": st.fri_domain_length, "max_degree": st.max_degree(air),
            "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest()}


if __name__ == "__main__":
    path = os.path.join(HERE, "scaled_stark.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    for a in sys.argv[1:]:
        t0 = time.time()
        data[a] = case(a)
        print(a, "%.1f s" % (time.time() - t0), data[a]["proof_sha256"], flush=True)
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
