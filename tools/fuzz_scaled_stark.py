"""Random trace lengths, expansion factors, colinearity counts and inputs: the device prover's proof against the
CPU restatement's, byte for byte, and the two verifiers' verdicts (on a GPU box).
usage: python3 tools/fuzz_scaled_stark.py [seed]"""
import ctypes as C, hashlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import host_mirror as hm
import stark_prover_oracle as sp, stark_scaled_oracle as ss, stark_oracle as so
P = so.P
lib = hm.lib
lib.sh_chain_new.restype = C.c_void_p
lib.sh_chain_random_bytes.restype = C.c_size_t
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
rp = sp.RescuePrime()
bad = 0
for case in range(24):
    rows = rng.choice([2, 3, 5, 27, 28, 29, 63, 64, 65, 100, 255, 257, 511, 700, 1023, 1025, 2047, 2500])
    e = rng.choice([4, 8, 16]); c = rng.choice([1, 2, 3, 4]); inp = rng.randrange(P)
    st = ss.ScaledStark(e, c, 2, rp.m, rows)
    trace = ss.rescue_chain_trace(rp, inp, rows)
    air, cols = ss.rescue_column_air(rp), ss.rescue_constant_columns(rp, rows)
    boundary = ss.rescue_chain_boundary(rp, rows, trace[-1][0])
    n_random = 17 * (st.num_randomizers * rp.m + st.max_degree(air) + 1)
    seed = b"fz-%d" % case
    want = st.prove(trace, air, cols, boundary, ss.stream_rng(seed, n_random))
    ok_ref = st.verify(want, air, cols, boundary)
    h = C.c_void_p(lib.sh_chain_new(hm.fe(inp), C.c_size_t(rows), C.c_size_t(e), C.c_size_t(c), C.c_size_t(2)))
    assert h, lib.sh_last_panic()
    assert lib.sh_chain_random_bytes(h) == n_random
    rb = hashlib.shake_256(seed).digest(n_random)
    buf = (C.c_uint8 * (1 << 21))(); plen = C.c_size_t(len(buf))
    rbuf = (C.c_uint8 * len(rb)).from_buffer_copy(rb)
    rc = lib.sh_chain_prove(h, rbuf, C.c_size_t(len(rb)), buf, C.byref(plen), None)
    assert rc >= 0, lib.sh_last_panic()
    got = bytes(buf[:plen.value])
    out = (C.c_uint64 * 2)(); lib.sh_chain_output(h, out)
    ok_dev = lib.sh_chain_verify(h, out, buf, plen)
    status = "ok" if (got == want and ok_dev == int(ok_ref)) else "MISMATCH"
    bad += status != "ok"
    print(case, rows, e, c, len(got), ok_ref, ok_dev, status, flush=True)
    lib.sh_chain_free(h)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
