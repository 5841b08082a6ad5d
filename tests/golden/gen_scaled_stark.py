"""Writes tests/golden/scaled_stark.json: proofs of the Rescue-Prime chain at several trace
lengths from the CPU restatement oracle/stark_scaled_oracle.py (length, SHA-256 and the claimed
output of each proof; the proofs themselves are 60-200 KB and are regenerated, not stored).

    python tests/golden/gen_scaled_stark.py 6 8 10 12 14 16      # log2 of the trace lengths

Parameters of every case: expansion factor 4, 2 colinearity tests, input element 12345, random
bytes = SHAKE256("scaled-stark-<log2 rows>") read as one stream (stream_rng)."""
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


def case(log_rows):
    rp = sp.RescuePrime()
    rows = 1 << log_rows
    st = ss.ScaledStark(EXPANSION, NCOL, SECURITY, rp.m, rows)
    trace = ss.rescue_chain_trace(rp, INPUT, rows)
    air = ss.rescue_column_air(rp)
    cols = ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
    n_random = 17 * (st.num_randomizers * rp.m + st.max_degree(air) + 1)
    seed = b"scaled-stark-%d" % log_rows
    proof = st.prove(trace, air, cols, boundary, ss.stream_rng(seed, n_random))
    assert st.verify(proof, air, cols, boundary)
    return {"log_rows": log_rows, "input": str(INPUT), "output": str(trace[-1][0]), "seed": seed.decode(),
            "random_bytes": n_random, "expansion": EXPANSION, "ncol": NCOL, "security": SECURITY,
            "fri_domain": st.fri_domain_length, "max_degree": st.max_degree(air),
            "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest()}


if __name__ == "__main__":
    path = os.path.join(HERE, "scaled_stark.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    for a in sys.argv[1:]:
        t0 = time.time()
        data[a] = case(int(a))
        print(a, "%.1f s" % (time.time() - t0), data[a]["proof_sha256"], flush=True)
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
