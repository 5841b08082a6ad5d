"""Time the mirror's Stark::prove / verify on the reference's own test case (Rescue-Prime, 28 rows)."""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stark_amd import hostlib as hm

buf = hashlib.shake_256(b"timing").digest(17 * 256)
rb = (C.c_uint8 * len(buf)).from_buffer_copy(buf)
proof = (C.c_uint8 * (1 << 20))()
out = (C.c_uint64 * 2)()
for rep in range(3):
    plen = C.c_size_t(len(proof))
    t0 = time.perf_counter()
    hm.chk(hm.lib.sh_stark_rescue_prove(hm.fe(1 + rep), C.c_size_t(4), C.c_size_t(2), C.c_size_t(2), rb,
                                        C.c_size_t(len(buf)), proof, C.byref(plen), out))
    t1 = time.perf_counter()
    ok = hm.lib.sh_stark_rescue_verify(out, C.c_size_t(4), C.c_size_t(2), C.c_size_t(2), proof, plen)
    t2 = time.perf_counter()
    print(f"prove {1e3 * (t1 - t0):.1f} ms  verify {1e3 * (t2 - t1):.1f} ms  accept={ok}  proof {plen.value} bytes")
