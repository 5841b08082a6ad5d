// Single-block BLAKE2b-256 compression for Merkle leaves and nodes
// (reference: src/modules/merkle.rs:29-32,41-47 -- Blake2b<U32>, unkeyed;
// algorithm: RFC 7693).  Every hash on this path fits one 128-byte block:
// a leaf is at most 25 bytes, a node is 64 bytes.  Rounds are fully unrolled
// so the zero message words fold away at compile time.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define B2_HD __host__ __device__ __forceinline__
#else
#define B2_HD inline
#endif

namespace starkh {

typedef uint64_t u64;

struct digest {
    u64 w[4];
};

// 64-bit rotate right.  On the device a rotation by 24, 16 or 63 is two
// v_alignbit_b32 over the 32-bit halves and a rotation by 32 is a register
// rename; the generic shift/or form compiles to 64-bit shifts, which issue at
// less than half the rate on gfx950.
B2_HD u64 rotr64(u64 x, int r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (r == 32) return ((u64)lo << 32) | hi;
    uint32_t nlo, nhi;
    if (r < 32) {
        nlo = __builtin_amdgcn_alignbit(hi, lo, r);
        nhi = __builtin_amdgcn_alignbit(lo, hi, r);
    } else {
        nlo = __builtin_amdgcn_alignbit(lo, hi, r - 32);
        nhi = __builtin_amdgcn_alignbit(hi, lo, r - 32);
    }
    return ((u64)nhi << 32) | nlo;
#else
    return (x >> r) | (x << (64 - r));
#endif
}

#define B2_IV0 0x6a09e667f3bcc908ull
#define B2_IV1 0xbb67ae8584caa73bull
#define B2_IV2 0x3c6ef372fe94f82bull
#define B2_IV3 0xa54ff53a5f1d36f1ull
#define B2_IV4 0x510e527fade682d1ull
#define B2_IV5 0x9b05688c2b3e6c1full
#define B2_IV6 0x1f83d9abfb41bd6bull
#define B2_IV7 0x5be0cd19137e2179ull

template <int R> struct sigma;
#define B2_SIGMA(R, ...)                                   \
    template <> struct sigma<R> {                          \
        static constexpr int s[16] = {__VA_ARGS__};        \
    };
B2_SIGMA(0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
B2_SIGMA(1, 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
B2_SIGMA(2, 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
B2_SIGMA(3, 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
B2_SIGMA(4, 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
B2_SIGMA(5, 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
B2_SIGMA(6, 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
B2_SIGMA(7, 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
B2_SIGMA(8, 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
B2_SIGMA(9, 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
B2_SIGMA(10, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
B2_SIGMA(11, 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
#undef B2_SIGMA

// message word I of an NW-word message (words >= NW are zero)
template <int NW, int I> B2_HD u64 mword(const u64 (&m)[NW])
{
    if constexpr (I < NW) return m[I];
    else return 0;
}

#define B2_G(a, b, c, d, x, y)                                 \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32);  \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24);  \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16);  \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);

template <int NW, int R> B2_HD void b2_round(u64 (&v)[16], const u64 (&m)[NW])
{
    using S = sigma<R>;
    B2_G(0, 4, 8, 12, (mword<NW, S::s[0]>(m)), (mword<NW, S::s[1]>(m)));
    B2_G(1, 5, 9, 13, (mword<NW, S::s[2]>(m)), (mword<NW, S::s[3]>(m)));
    B2_G(2, 6, 10, 14, (mword<NW, S::s[4]>(m)), (mword<NW, S::s[5]>(m)));
    B2_G(3, 7, 11, 15, (mword<NW, S::s[6]>(m)), (mword<NW, S::s[7]>(m)));
    B2_G(0, 5, 10, 15, (mword<NW, S::s[8]>(m)), (mword<NW, S::s[9]>(m)));
    B2_G(1, 6, 11, 12, (mword<NW, S::s[10]>(m)), (mword<NW, S::s[11]>(m)));
    B2_G(2, 7, 8, 13, (mword<NW, S::s[12]>(m)), (mword<NW, S::s[13]>(m)));
    B2_G(3, 4, 9, 14, (mword<NW, S::s[14]>(m)), (mword<NW, S::s[15]>(m)));
}
#undef B2_G

// BLAKE2b-256 of a message of t bytes (t <= 128) held in the first NW words.
template <int NW> B2_HD digest blake2b256_1block(const u64 (&m)[NW], u64 t)
{
    const u64 h0 = B2_IV0 ^ 0x01010020ull;        // digest_length=32, fanout=1, depth=1
    u64 v[16] = {h0, B2_IV1, B2_IV2, B2_IV3, B2_IV4, B2_IV5, B2_IV6, B2_IV7,
                 B2_IV0, B2_IV1, B2_IV2, B2_IV3, B2_IV4 ^ t, B2_IV5, ~B2_IV6, B2_IV7};
    b2_round<NW, 0>(v, m);
    b2_round<NW, 1>(v, m);
    b2_round<NW, 2>(v, m);
    b2_round<NW, 3>(v, m);
    b2_round<NW, 4>(v, m);
    b2_round<NW, 5>(v, m);
    b2_round<NW, 6>(v, m);
    b2_round<NW, 7>(v, m);
    b2_round<NW, 8>(v, m);
    b2_round<NW, 9>(v, m);
    b2_round<NW, 10>(v, m);
    b2_round<NW, 11>(v, m);
    digest d;
    d.w[0] = h0 ^ v[0] ^ v[8];
    d.w[1] = B2_IV1 ^ v[1] ^ v[9];
    d.w[2] = B2_IV2 ^ v[2] ^ v[10];
    d.w[3] = B2_IV3 ^ v[3] ^ v[11];
    return d;
}

// Leaf: BLAKE2b-256(bincode::serialize(&FieldElement)) (fri.rs:119).
// Bytes: sign (0/1), u64 LE digit count d, then d u32 LE digits (least
// significant first, no trailing zero digit).  The digits start at byte 9, so
// message words 1..3 are the 128-bit value shifted left by 8 bits; absent high
// digits are zero bytes either way and only the length t = 9 + 4d varies.
B2_HD digest leaf_digest(u64 lo, u64 hi)
{
    unsigned d = hi ? ((hi >> 32) ? 4u : 3u) : (lo ? ((lo >> 32) ? 2u : 1u) : 0u);
    u64 m[4];
    m[0] = (d ? 1ull : 0ull) | ((u64)d << 8);
    m[1] = lo << 8;
    m[2] = (lo >> 56) | (hi << 8);
    m[3] = hi >> 56;
    return blake2b256_1block<4>(m, 9 + 4 * d);
}

// Node: BLAKE2b-256(left || right) (merkle.rs:29-32).
B2_HD digest node_digest(const digest& l, const digest& r)
{
    u64 m[8] = {l.w[0], l.w[1], l.w[2], l.w[3], r.w[0], r.w[1], r.w[2], r.w[3]};
    return blake2b256_1block<8>(m, 64);
}

}  // namespace starkh
