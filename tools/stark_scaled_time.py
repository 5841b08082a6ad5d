"""Time the long-trace prover (stark_amd/host/stark_scaled.hpp): a Rescue-Prime chain of 2^k rows.
usage: python tools/stark_scaled_time.py [log2 rows = 20] [proofs = 3] [verify = 1] [by columns = 1]"""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_amd import hostlib as hm

log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
verify = int(sys.argv[3]) if len(sys.argv) > 3 else 1
by_columns = int(sys.argv[4]) if len(sys.argv) > 4 else 1
prove = hm.lib.sh_chain_prove_columns if by_columns else hm.lib.sh_chain_prove
hm.lib.sh_chain_new.restype = C.c_void_p
hm.lib.sh_chain_random_bytes.restype = C.c_size_t
t0 = time.perf_counter()
job = C.c_void_p(hm.lib.sh_chain_new(hm.fe(2024), C.c_size_t(1 << log_rows), C.c_size_t(4), C.c_size_t(2), C.c_size_t(2)))
assert job, hm.lib.sh_last_panic()
print(f"trace + tables: {time.perf_counter() - t0:.2f} s")
n_random = hm.lib.sh_chain_random_bytes(job)
buf = hashlib.shake_256(b"timing").digest(n_random)
rb = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
proof = (C.c_uint8 * (1 << 20))()
out = (C.c_uint64 * 2)()
hm.chk(hm.lib.sh_chain_output(job, out))
names = ("host", "interpolate", "quotients", "lde", "merkle", "fri", "openings", "total")
for rep in range(reps):
    plen = C.c_size_t(len(proof))
    tm = (C.c_double * 8)()
    t0 = time.perf_counter()
    hm.chk(prove(job, rb, C.c_size_t(len(buf)), proof, C.byref(plen), tm))
    t1 = time.perf_counter()
    line = f"prove {1e3 * (t1 - t0):.1f} ms  " + " ".join(f"{n}={v:.1f}" for n, v in zip(names, tm))
    if verify:
        ok = hm.lib.sh_chain_verify(job, out, proof, plen)
        line += f"  verify {1e3 * (time.perf_counter() - t1):.1f} ms accept={ok}"
    print(line + f"  proof {plen.value} bytes", flush=True)
hm.lib.sh_chain_free(job)
