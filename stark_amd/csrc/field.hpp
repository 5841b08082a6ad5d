// 128-bit prime field p = 407 * 2^119 + 1 (reference: src/modules/field.rs:32).
// One implementation for host (table generation) and device (kernels).
//
// Elements are 16 bytes {lo, hi}.  Values crossing the C-ABI are canonical
// integers in [0, p).  Inside kernels, CONSTANTS (twiddles, scale factors) are
// kept in Montgomery form x*2^128 mod p, so mont_mul(data, constant) returns
// the canonical product directly and data never needs converting.
//
// Reduction uses the shape of p: p == 1 (mod 2^64) gives -p^-1 == -1
// (mod 2^64), so each 64-bit Montgomery step is m = -t0 and
// m*p = m + ((m*407) << 119): one multiply by a 9-bit constant plus shifts.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define STARK_HD __host__ __device__ __forceinline__
#else
#define STARK_HD inline
#endif

namespace starkf {

typedef unsigned __int128 u128;
typedef uint64_t u64;

struct alignas(16) fe {
    u64 lo, hi;
};

constexpr u64 P_LO = 1ull;
constexpr u64 P_HI = 0xCB80000000000000ull;          // 407 << 55
constexpr u64 GEN_LO = 0xB5038F9C18F6F7D1ull;        // field.rs:29
constexpr u64 GEN_HI = 0x4040FBED12EE470Full;

STARK_HD u128 to_u128(fe a) { return ((u128)a.hi << 64) | a.lo; }
STARK_HD fe from_u128(u128 v) { return fe{(u64)v, (u64)(v >> 64)}; }
STARK_HD fe fe_zero() { return fe{0, 0}; }
STARK_HD bool fe_is_zero(fe a) { return (a.lo | a.hi) == 0; }
STARK_HD bool fe_eq(fe a, fe b) { return a.lo == b.lo && a.hi == b.hi; }

// true when (hi:lo) >= p
STARK_HD bool ge_p(u64 lo, u64 hi) { return hi > P_HI || (hi == P_HI && lo >= P_LO); }

// a + b mod p  (field.rs:120-123)
STARK_HD fe fe_add(fe a, fe b)
{
    u64 lo = a.lo + b.lo;
    u64 c = lo < a.lo;
    u64 hi = a.hi + b.hi;
    u64 c1 = hi < a.hi;
    u64 hi2 = hi + c;
    c1 |= (hi2 < hi);
    if (c1 || ge_p(lo, hi2)) {
        u64 nlo = lo - P_LO;
        u64 bor = lo < P_LO;
        hi2 = hi2 - P_HI - bor;
        lo = nlo;
    }
    return fe{lo, hi2};
}

// a - b mod p  (field.rs:125-132)
STARK_HD fe fe_sub(fe a, fe b)
{
    u64 lo = a.lo - b.lo;
    u64 bor = a.lo < b.lo;
    u64 hi = a.hi - b.hi - bor;
    bool neg = (a.hi < b.hi) || (a.hi == b.hi && bor);
    if (neg) {
        u64 nlo = lo + P_LO;
        u64 c = nlo < lo;
        hi = hi + P_HI + c;
        lo = nlo;
    }
    return fe{lo, hi};
}

STARK_HD fe fe_neg(fe a) { return fe_sub(fe_zero(), a); }

// a / 2 mod p: even -> shift; odd -> (a + p) / 2 = (a>>1) + (p>>1) + 1.
STARK_HD fe fe_half(fe a)
{
    u64 odd = a.lo & 1;
    u64 lo = (a.lo >> 1) | (a.hi << 63);
    u64 hi = a.hi >> 1;
    if (odd) {
        // (p >> 1) + 1 = {1, P_HI >> 1}
        u64 nlo = lo + 1;
        u64 c = nlo < lo;
        hi = hi + (P_HI >> 1) + c;
        lo = nlo;
    }
    return fe{lo, hi};
}

// Montgomery product a * b * 2^-128 mod p, inputs < p, output canonical < p.
STARK_HD fe mont_mul(fe a, fe b)
{
    u128 p00 = (u128)a.lo * b.lo;
    u128 p01 = (u128)a.lo * b.hi;
    u128 p10 = (u128)a.hi * b.lo;
    u128 p11 = (u128)a.hi * b.hi;
    u64 t0 = (u64)p00;
    u128 mid = (p00 >> 64) + (u64)p01 + (u64)p10;
    u64 t1 = (u64)mid;
    u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + p11;      // words t3:t2, cannot overflow (product < 2^256)

    // step 0: m0 = -t0; T = (T + m0*p) >> 64
    u64 m0 = 0 - t0;
    u128 q0 = ((u128)m0 * 407u) << 55;                            // m0 * P_HI
    u128 s0 = (u128)t1 + (u64)q0 + (u64)(t0 != 0);
    u64 u1 = (u64)s0;
    u128 hi1 = hi + (q0 >> 64) + (s0 >> 64);                      // < 2^128 (T/2^64 + p fits 192 bits)

    // step 1: m1 = -u1
    u64 m1 = 0 - u1;
    u128 q1 = ((u128)m1 * 407u) << 55;
    u128 s1 = (u128)(u64)hi1 + (u64)q1 + (u64)(u1 != 0);
    u64 r0 = (u64)s1;
    u128 s2 = (hi1 >> 64) + (q1 >> 64) + (s1 >> 64);              // up to 65 bits: result < 1.8p may exceed 2^128
    u64 r1 = (u64)s2;
    u64 carry = (u64)(s2 >> 64);
    if (carry || ge_p(r0, r1)) {
        u64 nlo = r0 - P_LO;
        u64 bor = r0 < P_LO;
        r1 = r1 - P_HI - bor;
        r0 = nlo;
    }
    return fe{r0, r1};
}

}  // namespace starkf
