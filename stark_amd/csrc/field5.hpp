// Lazy 5-limb arithmetic for the butterfly kernels.
//
// gfx950 has no fast carry chain: a VALU instruction that consumes VCC needs
// wait states after the one that produced it, so 128-bit modular add/sub on
// 32-bit limbs costs ~75 cycles per wave and the 4-limb Montgomery product
// ~400.  Here an element is five SIGNED 32-bit limbs in base 2^26,
//     value = l0 + l1*2^26 + l2*2^52 + l3*2^78 + l4*2^104,
// add/sub are five independent 32-bit operations with no carries (limbs may
// grow to 31 bits between multiplications), and the product accumulates
// columns in 64-bit registers with v_mad_i64_i32, propagating carries by
// shifts.  p = 1 + P4*2^104 with P4 = 407*2^15, so a Montgomery digit step is
// m = digit, T -= m*p: the low digit cancels by construction and only one
// multiply-add (m * P4, four digits up) is needed.
//
// mont_mul5(a, w) returns a*w*2^-156 mod p up to a multiple of p, in
// (-2^107, p + 2^107], for ANY int32 limbs of a and a normalised w (limbs in
// [0, 2^26), value < p).  With PLUS_P = false the final "+ p" is left out and the
// result lies in (-p - 2^107, 2^107]: as good an operand for further lazy sums and
// products, two instructions cheaper, but not for pack_fast / finish, which expect a
// non-negative value.  Six digit steps (R = 2^156) rather than five make the
// result nearly canonical whatever the laziness of a.  Twiddles are stored as
// w*2^156 mod p (struct tw5).  pack_canonical() finishes the reduction.
#pragma once
#include <math.h>
#include <stdint.h>

#include "field.hpp"

namespace starkf {

constexpr int LIMB_BITS = 26;
constexpr int32_t LIMB_MASK = (1 << LIMB_BITS) - 1;
constexpr int32_t P4 = 407 << 15;             // p = 1 + P4 * 2^104

struct fe5 {
    int32_t l[5];
};

// a twiddle: five normalised limbs of (w * 2^156 mod p), padded to 32 bytes
struct alignas(32) tw5 {
    int32_t l[5];
    int32_t pad[3];
};

STARK_HD uint32_t alignbit32(uint32_t hi, uint32_t lo, uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}

// canonical 128-bit value -> normalised limbs
STARK_HD fe5 unpack5(fe a)
{
    const uint32_t w0 = (uint32_t)a.lo, w1 = (uint32_t)(a.lo >> 32), w2 = (uint32_t)a.hi, w3 = (uint32_t)(a.hi >> 32);
    fe5 r;
    r.l[0] = (int32_t)(w0 & LIMB_MASK);
    r.l[1] = (int32_t)(alignbit32(w1, w0, 26) & LIMB_MASK);
    r.l[2] = (int32_t)(alignbit32(w2, w1, 20) & LIMB_MASK);
    r.l[3] = (int32_t)(alignbit32(w3, w2, 14) & LIMB_MASK);
    r.l[4] = (int32_t)(w3 >> 8);
    return r;
}

STARK_HD fe5 add5(const fe5& a, const fe5& b)
{
    fe5 r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

STARK_HD fe5 sub5(const fe5& a, const fe5& b)
{
    fe5 r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}

// a * w * 2^-156 (mod p), result limbs: l0 in [1, 2^26], l1..l3 in [0, 2^26),
// l4 = small signed carry + P4; value in (-2^107, p + 2^107].
STARK_HD int64_t mad64(int32_t a, int32_t b, int64_t c) { return (int64_t)a * (int64_t)b + c; }

template <bool PLUS_P = true, class W>
STARK_HD fe5 mont_mul5(const fe5& a, const W& w)
{
    const int32_t a0 = a.l[0], a1 = a.l[1], a2 = a.l[2], a3 = a.l[3], a4 = a.l[4];
    const int32_t w0 = w.l[0], w1 = w.l[1], w2 = w.l[2], w3 = w.l[3], w4 = w.l[4];
    const int32_t NP4 = -P4;
    int64_t t;
    int32_t m0, m1, m2, m3, m4, m5;
    // each column is one chain of multiply-adds that starts from the carry of
    // the column below
    t = mad64(a0, w0, 0);
    m0 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(a0, w1, t);
    t = mad64(a1, w0, t);
    m1 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(a0, w2, t);
    t = mad64(a1, w1, t);
    t = mad64(a2, w0, t);
    m2 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(a0, w3, t);
    t = mad64(a1, w2, t);
    t = mad64(a2, w1, t);
    t = mad64(a3, w0, t);
    m3 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(a0, w4, t);
    t = mad64(a1, w3, t);
    t = mad64(a2, w2, t);
    t = mad64(a3, w1, t);
    t = mad64(a4, w0, t);
    t = mad64(m0, NP4, t);
    m4 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(a1, w4, t);
    t = mad64(a2, w3, t);
    t = mad64(a3, w2, t);
    t = mad64(a4, w1, t);
    t = mad64(m1, NP4, t);
    m5 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    fe5 r;
    t = mad64(a2, w4, t);
    t = mad64(a3, w3, t);
    t = mad64(a4, w2, t);
    t = mad64(m2, NP4, t);
    r.l[0] = ((int32_t)t & LIMB_MASK) + (PLUS_P ? 1 : 0);   // + p, low digit
    t >>= LIMB_BITS;
    t = mad64(a3, w4, t);
    t = mad64(a4, w3, t);
    t = mad64(m3, NP4, t);
    r.l[1] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(a4, w4, t);
    t = mad64(m4, NP4, t);
    r.l[2] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(m5, NP4, t);
    r.l[3] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    r.l[4] = (int32_t)t + (PLUS_P ? P4 : 0);          // + p, high digit
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Hand-scheduled form of mont_mul5 for the device: one serial column chain on a
// fixed accumulator pair, 31 v_mad_i64_i32 + 10 v_and + 10 v_ashrrev_i64.  The
// pair is named explicitly because the chain reads its low half on its own and
// inline asm has no sub-register operand; v[166:167] is the top of the
// 168-register budget of three waves per SIMD, so the pair itself never costs
// occupancy.  hipcc re-associates the C++ form into "sum the column's
// products, then add the carry", which costs one more 64-bit add per column.
// WS selects where the twiddle limbs live: "s" (wave-uniform, scalar registers)
// or "v".  VOP3 reads at most one scalar operand, which every line respects.
#define F5_ACC "v[166:167]"
#define F5_ACC_LO "v166"
#define F5_ACC_HI "v167"
#define STARK_MM5_BODY                                            \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a0], %[w0], 0\n\t"          \
    "v_and_b32 %[m0], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a0], %[w1], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a1], %[w0], " F5_ACC "\n\t" \
    "v_and_b32 %[m1], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a0], %[w2], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a1], %[w1], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a2], %[w0], " F5_ACC "\n\t" \
    "v_and_b32 %[m2], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a0], %[w3], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a1], %[w2], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a2], %[w1], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a3], %[w0], " F5_ACC "\n\t" \
    "v_and_b32 %[m3], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a0], %[w4], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a1], %[w3], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a2], %[w2], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a3], %[w1], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a4], %[w0], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[m0], %[np4], " F5_ACC "\n\t" \
    "v_and_b32 %[m0], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a1], %[w4], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a2], %[w3], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a3], %[w2], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a4], %[w1], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[m1], %[np4], " F5_ACC "\n\t" \
    "v_and_b32 %[m1], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a2], %[w4], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a3], %[w3], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a4], %[w2], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[m2], %[np4], " F5_ACC "\n\t" \
    "v_and_b32 %[r0], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a3], %[w4], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a4], %[w3], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[m3], %[np4], " F5_ACC "\n\t" \
    "v_and_b32 %[r1], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[a4], %[w4], " F5_ACC "\n\t" \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[m0], %[np4], " F5_ACC "\n\t" \
    "v_and_b32 %[r2], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_ashrrev_i64 " F5_ACC ", 26, " F5_ACC "\n\t"                \
    "v_mad_i64_i32 " F5_ACC ", vcc, %[m1], %[np4], " F5_ACC "\n\t" \
    "v_and_b32 %[r3], 0x3ffffff, " F5_ACC_LO "\n\t"                        \
    "v_alignbit_b32 %[r4], " F5_ACC_HI ", " F5_ACC_LO ", 26"

#define STARK_MM5_DEFINE(NAME, WS)                                                                            \
    template <bool PLUS_P = true, class W>                                                                    \
    __device__ __forceinline__ fe5 NAME(const fe5& a, const W& w)                                             \
    {                                                                                                         \
        int32_t r0, r1, r2, r3, r4, m0, m1, m2, m3;                                                           \
        asm(STARK_MM5_BODY                                                                                    \
            : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [m0] "=&v"(m0), \
              [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3)                                                  \
            : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]),       \
              [w0] WS(w.l[0]), [w1] WS(w.l[1]), [w2] WS(w.l[2]), [w3] WS(w.l[3]), [w4] WS(w.l[4]),            \
              [np4] "s"(-P4)                                                                                  \
            : "vcc", F5_ACC_LO, F5_ACC_HI);                                                                         \
        fe5 r;                                                                                                \
        r.l[0] = PLUS_P ? r0 + 1 : r0;                                                                        \
        r.l[1] = r1;                                                                                          \
        r.l[2] = r2;                                                                                          \
        r.l[3] = r3;                                                                                          \
        r.l[4] = PLUS_P ? r4 + P4 : r4;                                                                       \
        return r;                                                                                             \
    }
STARK_MM5_DEFINE(mont_mul5_v, "v")   // twiddle limbs in vector registers
STARK_MM5_DEFINE(mont_mul5_s, "s")   // twiddle limbs wave-uniform, in scalar registers
#undef STARK_MM5_DEFINE
#else
template <bool PLUS_P = true, class W> STARK_HD fe5 mont_mul5_v(const fe5& a, const W& w) { return mont_mul5<PLUS_P>(a, w); }
template <bool PLUS_P = true, class W> STARK_HD fe5 mont_mul5_s(const fe5& a, const W& w) { return mont_mul5<PLUS_P>(a, w); }
#endif

// A wave-uniform constant multiplier w, tabulated against every limb position of the other operand:
// l[k][i] is limb k of (w * 2^(26 i + 52) mod p), normalised.  With it a * w is the sum over i of
// a_i * (row i): 25 multiply-adds that fill only the five low columns, so two Montgomery digit steps
// (instead of six) and six digit extractions (instead of ten) bring the result into range.
struct alignas(16) tw25 {
    int32_t l[5][5];
    int32_t pad[3];
};

// a * w (w given as its tw25 table), same contract as mont_mul5's result: value = a * w mod p up to a
// multiple of p, in (-2^110, p + 2^110) for ANY int32 limbs of a; l0 in [1, 2^26], l1..l3 in [0, 2^26),
// l4 = small signed carry + P4.  PLUS_P = false: without the final "+ p", in (-p - 2^110, 2^110).
template <bool PLUS_P = true>
STARK_HD fe5 mont_mul5_c_ref(const fe5& a, const tw25& w)
{
    const int32_t NP4 = -P4;
    int64_t t = 0;
    int32_t m0, m1;
    fe5 r;
#pragma unroll
    for (int i = 0; i < 5; i++) t = mad64(a.l[i], w.l[0][i], t);
    m0 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
#pragma unroll
    for (int i = 0; i < 5; i++) t = mad64(a.l[i], w.l[1][i], t);
    m1 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
#pragma unroll
    for (int i = 0; i < 5; i++) t = mad64(a.l[i], w.l[2][i], t);
    r.l[0] = ((int32_t)t & LIMB_MASK) + (PLUS_P ? 1 : 0);
    t >>= LIMB_BITS;
#pragma unroll
    for (int i = 0; i < 5; i++) t = mad64(a.l[i], w.l[3][i], t);
    r.l[1] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
#pragma unroll
    for (int i = 0; i < 5; i++) t = mad64(a.l[i], w.l[4][i], t);
    t = mad64(m0, NP4, t);
    r.l[2] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    t = mad64(m1, NP4, t);
    r.l[3] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    r.l[4] = (int32_t)t + (PLUS_P ? P4 : 0);
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Device form: one asm block per column (the compiler would otherwise re-associate the chain into
// "sum the products, then add the carry"); the digit extraction between the blocks is plain C++.
// The table limbs are wave-uniform and sit in scalar registers; VOP3 reads one scalar operand.
#define STARK_MM5C_COL(T, K)                                                                                  \
    asm("v_mad_i64_i32 %[t], vcc, %[a0], %[w0], %[t]\n\t"                                                     \
        "v_mad_i64_i32 %[t], vcc, %[a1], %[w1], %[t]\n\t"                                                     \
        "v_mad_i64_i32 %[t], vcc, %[a2], %[w2], %[t]\n\t"                                                     \
        "v_mad_i64_i32 %[t], vcc, %[a3], %[w3], %[t]\n\t"                                                     \
        "v_mad_i64_i32 %[t], vcc, %[a4], %[w4], %[t]"                                                         \
        : [t] "+v"(T)                                                                                         \
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]),           \
          [w0] "s"(w.l[K][0]), [w1] "s"(w.l[K][1]), [w2] "s"(w.l[K][2]), [w3] "s"(w.l[K][3]), [w4] "s"(w.l[K][4]) \
        : "vcc")
template <bool PLUS_P = true>
__device__ __forceinline__ fe5 mont_mul5_c(const fe5& a, const tw25& w)
{
    int64_t t;
    int32_t m0, m1;
    fe5 r;
    asm("v_mad_i64_i32 %[t], vcc, %[a0], %[w0], 0\n\t"
        "v_mad_i64_i32 %[t], vcc, %[a1], %[w1], %[t]\n\t"
        "v_mad_i64_i32 %[t], vcc, %[a2], %[w2], %[t]\n\t"
        "v_mad_i64_i32 %[t], vcc, %[a3], %[w3], %[t]\n\t"
        "v_mad_i64_i32 %[t], vcc, %[a4], %[w4], %[t]"
        : [t] "=&v"(t)
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]),
          [w0] "s"(w.l[0][0]), [w1] "s"(w.l[0][1]), [w2] "s"(w.l[0][2]), [w3] "s"(w.l[0][3]), [w4] "s"(w.l[0][4])
        : "vcc");
    m0 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    STARK_MM5C_COL(t, 1);
    m1 = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    STARK_MM5C_COL(t, 2);
    r.l[0] = ((int32_t)t & LIMB_MASK) + (PLUS_P ? 1 : 0);
    t >>= LIMB_BITS;
    STARK_MM5C_COL(t, 3);
    r.l[1] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    STARK_MM5C_COL(t, 4);
    asm("v_mad_i64_i32 %[t], vcc, %[m], %[np4], %[t]" : [t] "+v"(t) : [m] "v"(m0), [np4] "s"(-P4) : "vcc");
    r.l[2] = (int32_t)t & LIMB_MASK;
    t >>= LIMB_BITS;
    asm("v_mad_i64_i32 %[t], vcc, %[m], %[np4], %[t]" : [t] "+v"(t) : [m] "v"(m1), [np4] "s"(-P4) : "vcc");
    r.l[3] = (int32_t)t & LIMB_MASK;
    r.l[4] = (int32_t)(t >> LIMB_BITS) + (PLUS_P ? P4 : 0);
    return r;
}
#undef STARK_MM5C_COL
#else
template <bool PLUS_P = true> STARK_HD fe5 mont_mul5_c(const fe5& a, const tw25& w) { return mont_mul5_c_ref<PLUS_P>(a, w); }
#endif

// Exact reduction of any limb vector with |value| < 2^150 into [0, p); the slow
// path behind pack_canonical.
STARK_HD fe reduce5_slow(const fe5& r)
{
    // carry-propagate into normalised digits with a signed top
    int64_t c = 0;
    int64_t d[5];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c += r.l[i];
        d[i] = c & LIMB_MASK;
        c >>= LIMB_BITS;
    }
    d[4] = c + r.l[4];                                  // signed, |d4| < 2^46
    // value = L + d4 * 2^104, 0 <= L < 2^104.  Remove q = floor(d4 / P4) copies of p:
    // p = 1 + P4*2^104, so value - q*p = (L - q) + (d4 - q*P4) * 2^104.
    int64_t q = d[4] / P4;
    if (d[4] % P4 < 0) q -= 1;
    int64_t top = d[4] - q * P4;                        // in [0, P4)
    // low part L - q with |q| < 2^23, fixed up against top so that the value
    // top*2^104 + low lands in [0, p).  (p itself does not fit a signed 128-bit
    // integer, so the two parts are handled separately.)
    const __int128 B104 = (__int128)1 << 104;
    __int128 low = (__int128)d[0] + ((__int128)d[1] << 26) + ((__int128)d[2] << 52) + ((__int128)d[3] << 78) - q;
    if (low < 0) {
        if (top > 0) { top -= 1; low += B104; }
        else { top = P4 - 1; low += B104 + 1; }         // value was negative: add p
    }
    if (low >= B104) { top += 1; low -= B104; }
    if (top == P4 && low >= 1) { top = 0; low -= 1; }   // value >= p: subtract p
    return from_u128(((unsigned __int128)(uint64_t)top << 104) | (unsigned __int128)low);
}

// limbs of a mont_mul5 result (or any normalised vector) -> canonical 128 bits
STARK_HD fe pack_fast(const fe5& r)
{
    const uint32_t l0 = (uint32_t)r.l[0], l1 = (uint32_t)r.l[1], l2 = (uint32_t)r.l[2], l3 = (uint32_t)r.l[3],
                   l4 = (uint32_t)r.l[4];
    const uint32_t w0 = (l1 << 26) | l0;
    const uint32_t w1 = (l2 << 20) | (l1 >> 6);
    const uint32_t w2 = (l3 << 14) | (l2 >> 12);
    const uint32_t w3 = (l4 << 8) | (l3 >> 18);
    return fe{((uint64_t)w1 << 32) | w0, ((uint64_t)w3 << 32) | w2};
}

// nonzero when r (a mont_mul5 result) is NOT already a canonical normalised
// vector: low limb overflowed its 26 bits, or the top limb is outside [0, P4).
STARK_HD uint32_t noncanonical5(const fe5& r)
{
    return ((uint32_t)r.l[0] >> LIMB_BITS) | (uint32_t)((uint32_t)r.l[4] >= (uint32_t)P4);
}

STARK_HD fe pack_canonical(const fe5& r)
{
    if (noncanonical5(r)) return reduce5_slow(r);
    return pack_fast(r);
}

// Any lazy limb vector with |value| < 2^134 (sums of a few products and canonical values) -> canonical
// 128 bits, without a multiplication: q ~ floor(value / p) from the two top limbs,
// value - q*p = (L - q) + (top - q*P4) * 2^104, one carry sweep, and the exact slow path whenever the top
// digit does not land in [0, P4) (q off by one, or value = p - 1).
STARK_HD fe finish_lazy(const fe5& x)
{
    const int32_t d_top = x.l[4] + (x.l[3] >> LIMB_BITS);
    const int32_t q = (int32_t)floorf((float)d_top * (1.0f / (float)P4));
    fe5 r;
    int32_t c = x.l[0] - q;
    r.l[0] = c & LIMB_MASK;
    c >>= LIMB_BITS;
    c += x.l[1];
    r.l[1] = c & LIMB_MASK;
    c >>= LIMB_BITS;
    c += x.l[2];
    r.l[2] = c & LIMB_MASK;
    c >>= LIMB_BITS;
    c += x.l[3];
    r.l[3] = c & LIMB_MASK;
    c >>= LIMB_BITS;
    r.l[4] = c + x.l[4] - q * P4;
    fe o = pack_fast(r);
    if (__builtin_expect((uint32_t)r.l[4] >= (uint32_t)P4, 0)) o = reduce5_slow(x);
    return o;
}

}  // namespace starkf
