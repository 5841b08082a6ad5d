"""Host build of the lazy 5-limb field core (stark_amd/csrc/field5.hpp, the
arithmetic inside the NTT kernels) against Python big integers.  No GPU."""
import ctypes as C
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "_build", "libfield5_host.so")
P = 407 * 2**119 + 1
R = 2**156
RINV = pow(R, -1, P)
P4 = 407 << 15
I5 = C.c_int32 * 5


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO,
                           os.path.join(ROOT, "tests", "host_field5_shim.cpp")])
    return C.CDLL(SO)


def val(l):
    return sum(int(x) << (26 * i) for i, x in enumerate(l))


def limbs(v):
    return [(v >> (26 * i)) & (2**26 - 1) for i in range(4)] + [v >> 104]


def _pack(lib, l):
    o2 = (C.c_uint64 * 2)()
    fl = C.c_uint32()
    lib.t_pack_canonical(I5(*l), o2, C.byref(fl))
    return o2[0] | (o2[1] << 64), fl.value


def test_unpack_roundtrip(lib):
    random.seed(3)
    for v in [0, 1, P - 1, 2**64, 2**104 - 1, 2**104, 2**26 - 1, 2**26] + [random.randrange(P) for _ in range(2000)]:
        out = I5()
        lib.t_unpack5((C.c_uint64 * 2)(v & (2**64 - 1), v >> 64), out)
        assert list(out) == limbs(v)
        assert _pack(lib, list(out))[0] == v


def test_reduce_slow_any_limbs(lib):
    random.seed(4)
    cases = [[random.randrange(-2**31, 2**31) for _ in range(5)] for _ in range(20000)]
    cases += [[1, 0, 0, 0, P4], [0, 0, 0, 0, 0], [0, 0, 0, 0, P4], [2**26, 2**26 - 1, 2**26 - 1, 2**26 - 1, P4 - 1],
              [2**26, 2**26 - 1, 2**26 - 1, 2**26 - 1, -1], [1, 0, 0, 0, -1], [0, 0, 0, 0, -1], [2, 0, 0, 0, P4],
              [0, 0, 0, 0, 2 * P4], [1, 0, 0, 0, 2 * P4], [-1, 0, 0, 0, 0], [0, -1, 0, 0, 0]]
    for a in cases:
        o2 = (C.c_uint64 * 2)()
        lib.t_reduce5(I5(*a), o2)
        assert (o2[0] | (o2[1] << 64)) == val(a) % P, a


def test_mont_mul5_lazy_inputs(lib):
    random.seed(1)
    special_w = [0, 1, P - 1, R % P, (R * R) % P, 2**104, 2**104 - 1]
    special_l = [0, 1, -1, 2**26 - 1, 2**26, -2**31, 2**31 - 1, 2**30]
    flagged = 0
    for it in range(60000):
        mode = it % 4
        if mode == 0:
            a = [random.randrange(-2**31, 2**31) for _ in range(5)]
        elif mode == 1:
            a = limbs(random.randrange(P))
        elif mode == 2:
            a = [random.choice(special_l) for _ in range(5)]
        else:
            a = limbs(random.choice([0, 1, P - 1, P - 2, 2]))
        wv = random.choice(special_w) if it % 3 == 0 else random.randrange(P)
        out = I5()
        lib.t_mont_mul5(I5(*a), I5(*limbs(wv)), out, 1)
        r = val(out)
        exp = (val(a) * wv * RINV) % P
        assert r % P == exp
        assert -2**107 < r <= P + 2**107
        assert 1 <= out[0] <= 2**26 and all(0 <= out[i] < 2**26 for i in (1, 2, 3))
        pc, fl = _pack(lib, list(out))
        assert pc == exp
        # the fast path is only taken for canonical normalised vectors
        if not fl:
            assert r == exp
        flagged += bool(fl)
    assert flagged > 0          # the slow path was exercised (zero products give r == p)


def test_zero_times_anything_is_flagged_and_exact(lib):
    out = I5()
    lib.t_mont_mul5(I5(0, 0, 0, 0, 0), I5(*limbs(12345)), out, 1)
    assert list(out) == [1, 0, 0, 0, P4]            # the representation p of zero
    assert _pack(lib, list(out)) == (0, 1)
    # the lazy representation p of zero as an input
    lib.t_mont_mul5(I5(1, 0, 0, 0, P4), I5(*limbs((R * R) % P)), out, 1)
    assert _pack(lib, list(out))[0] == 0


def table25(w):
    """limb k of (w * 2^(26 i + 52) mod p) at [5 k + i]: the tabulated form of a constant multiplier"""
    rows = [limbs((w << (26 * i + 52)) % P) for i in range(5)]
    return [rows[i][k] for k in range(5) for i in range(5)]


def test_mont_mul5_tabulated_constant(lib):
    """mont_mul5_c: a * w from the 25-limb table of w, for lazy a; same output contract as mont_mul5."""
    random.seed(11)
    I25 = C.c_int32 * 25
    special_w = [0, 1, P - 1, 2**104, 2**104 - 1, 2**52, pow(2, -52, P)]
    special_l = [0, 1, -1, 2**26 - 1, 2**26, -2**31, 2**31 - 1, 2**30]
    flagged = 0
    for it in range(40000):
        mode = it % 4
        if mode == 0:
            a = [random.randrange(-2**31, 2**31) for _ in range(5)]
        elif mode == 1:
            a = limbs(random.randrange(P))
        elif mode == 2:
            a = [random.choice(special_l) for _ in range(5)]
        else:
            a = limbs(random.choice([0, 1, P - 1, P - 2, 2]))
        wv = random.choice(special_w) if it % 3 == 0 else random.randrange(P)
        out = I5()
        lib.t_mont_mul5_c(I5(*a), I25(*table25(wv)), out, 1)
        r = val(out)
        exp = (val(a) * wv) % P
        assert r % P == exp
        assert -2**110 < r < P + 2**110
        assert 1 <= out[0] <= 2**26 and all(0 <= out[i] < 2**26 for i in (1, 2, 3))
        pc, fl = _pack(lib, list(out))
        assert pc == exp
        if not fl:
            assert r == exp
        flagged += bool(fl)
    assert flagged > 0


def test_products_without_the_final_plus_p(lib):
    """PLUS_P = false (the products whose results only feed further lazy sums and products): the same
    residue, p lower, limbs within the same bounds."""
    random.seed(12)
    I25 = C.c_int32 * 25
    for it in range(20000):
        a = [random.randrange(-2**31, 2**31) for _ in range(5)] if it % 2 else limbs(random.randrange(P))
        wv = random.randrange(P)
        o1, o0 = I5(), I5()
        lib.t_mont_mul5(I5(*a), I5(*limbs(wv)), o1, 1)
        lib.t_mont_mul5(I5(*a), I5(*limbs(wv)), o0, 0)
        assert val(o1) - val(o0) == P
        assert -P - 2**107 < val(o0) <= 2**107 and abs(o0[4]) <= P4 + 2**5
        assert all(0 <= o0[i] < 2**26 for i in range(4))
        lib.t_mont_mul5_c(I5(*a), I25(*table25(wv)), o1, 1)
        lib.t_mont_mul5_c(I5(*a), I25(*table25(wv)), o0, 0)
        assert val(o1) - val(o0) == P
        assert val(o0) % P == (val(a) * wv) % P
        assert -P - 2**110 < val(o0) < 2**110 and abs(o0[4]) <= P4 + 2**8
        assert all(0 <= o0[i] < 2**26 for i in range(4))


def test_tw16_literal_table_is_current():
    """stark_amd/csrc/tw16_tables.hpp is what tools/gen_tw16_tables.py writes, and its entries are the
    tabulated powers of primitive_nth_root(16) (field.rs:96-107) and of its inverse (ntt.rs:18-19)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_tw16", os.path.join(ROOT, "tools", "gen_tw16_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(gen.OUT).read() == gen.render()
    g = 85408008396924667383611388730472331217
    root = g
    for _ in range(119 - 4):
        root = root * root % P
    assert pow(root, 16, P) == 1 and pow(root, 8, P) == P - 1
    assert gen.table25(pow(root, 3, P)) == [[limbs((pow(root, 3, P) << (26 * i + 52)) % P)[k] for i in range(5)] for k in range(5)]
    assert ("{" + ", ".join(str(x) for x in gen.table25(pow(root, -1, P))[0]) + "}") in open(gen.OUT).read()
