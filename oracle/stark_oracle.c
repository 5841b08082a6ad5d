/*
 * CPU restatement, in plain C, of the reference prover hot path.
 *
 * TEST INFRASTRUCTURE ONLY: this library is the parity checker and the "port"
 * CPU baseline.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product library never links it.
 *
 * Each function names the reference function it restates (paths relative to
 * the reference checkout).  Field values are canonical integers in [0,p), so
 * any algebraically equal evaluation order gives bit-identical results; where
 * this file departs from the reference's evaluation order for speed (running
 * products instead of pow(i), one inverse instead of two per fold output) the
 * comment says so, and tests/ check those forms against the literal Python
 * restatement in stark_oracle.py.
 *
 * Pinning: field core pinned by the reference's known-answer values (see
 * tests/test_oracle_golden.py).  Encodings, Merkle roots, Fiat-Shamir bytes and
 * FRI codewords: PARITY UNPINNED (no golden value exists in the reference and
 * it cannot be built here: no Rust toolchain); they follow the documented wire
 * formats of bincode 1.3.3 / num-bigint 0.4.6 / serde_json 1.0.120 and the
 * BLAKE2b (RFC 7693) and SHAKE256 (FIPS 202) standards.
 *
 * Element layout everywhere: 16 bytes, two little-endian u64 {lo, hi}.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* Worker threads for the bulk loops (leaf and node hashing, fold, butterflies).  1 = the plain
 * sequential loops; more only changes who computes which element (the all-cores CPU baseline),
 * never a value: tests/ compare the two. */
static int g_threads = 1;
void so_set_threads(int t) { g_threads = t < 1 ? 1 : t; }
int so_get_threads(void) { return g_threads; }
#define PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)")

/* ------------------------------------------------------------------------ */
/* field.rs                                                                  */
/* ------------------------------------------------------------------------ */
#define P_HI 0xCB80000000000000ULL      /* 407 << 55 */
#define P_LO 0x0000000000000001ULL
static const u128 P = ((u128)P_HI << 64) | P_LO;                       /* field.rs:32 */
#define GEN_HI 0x4040FBED12EE470FULL      /* 85408008396924667383611388730472331217 >> 64 */
#define GEN_LO 0xB5038F9C18F6F7D1ULL

static inline u128 ld(const u64 *p) { return ((u128)p[1] << 64) | p[0]; }
static inline void st(u64 *p, u128 v) { p[0] = (u64)v; p[1] = (u64)(v >> 64); }

static inline u128 fe_add(u128 a, u128 b)          /* field.rs:120-123 */
{
    u128 s = a + b;                    /* a,b < p < 2^128: may wrap */
    if (s < a || s >= P) s -= P;
    return s;
}
static inline u128 fe_sub(u128 a, u128 b)          /* field.rs:125-132 */
{
    return (a >= b) ? a - b : a + (P - b);
}

/* Montgomery product a*b*2^-128 mod p.  p == 1 (mod 2^64) so -p^-1 == -1
 * (mod 2^64): each reduction step uses m = -t_i. */
static inline u128 mont_mul(u128 a, u128 b)
{
    u64 a0 = (u64)a, a1 = (u64)(a >> 64), b0 = (u64)b, b1 = (u64)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u64 t0 = (u64)p00;
    u128 mid = (p00 >> 64) + (u64)p01 + (u64)p10;
    u64 t1 = (u64)mid;
    u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (u64)p11;
    u64 t2 = (u64)hi;
    u64 t3 = (u64)((hi >> 64) + (p11 >> 64));
    u64 t4 = 0;
    /* step 0 */
    {
        u64 m = (u64)0 - t0;
        u128 c = ((u128)t0 + m) >> 64;                 /* low word becomes 0 */
        u128 mp = (u128)m * P_HI;
        u128 s = (u128)t1 + (u64)mp + c; t1 = (u64)s;
        s = (u128)t2 + (u64)(mp >> 64) + (s >> 64); t2 = (u64)s;
        s = (u128)t3 + (s >> 64); t3 = (u64)s;
        t4 += (u64)(s >> 64);
    }
    /* step 1 */
    {
        u64 m = (u64)0 - t1;
        u128 c = ((u128)t1 + m) >> 64;
        u128 mp = (u128)m * P_HI;
        u128 s = (u128)t2 + (u64)mp + c; t2 = (u64)s;
        s = (u128)t3 + (u64)(mp >> 64) + (s >> 64); t3 = (u64)s;
        t4 += (u64)(s >> 64);
    }
    u128 r = ((u128)t3 << 64) | t2;
    if (t4 || r >= P) r -= P;
    return r;
}

static u128 R2;          /* 2^256 mod p */
static int g_init = 0;
static void oracle_init(void)
{
    if (g_init) return;
    /* 2^128 mod p by doubling from 1, then square it the slow way. */
    u128 r = 1;
    for (int i = 0; i < 256; i++) r = fe_add(r, r);
    R2 = r;
    g_init = 1;
}

static inline u128 fe_mul(u128 a, u128 b)          /* field.rs:134-137 */
{
    return mont_mul(mont_mul(a, b), R2);
}
static inline u128 to_mont(u128 a) { return mont_mul(a, R2); }

static u128 fe_pow(u128 base, u128 exp)            /* field.rs:42-56 */
{
    u128 result = 1;
    while (exp) {
        if (exp & 1) result = fe_mul(result, base);
        base = fe_mul(base, base);
        exp >>= 1;
    }
    return result;
}

/* field.rs:67-93 computes the inverse by extended Euclid; the value equals
 * a^(p-2) for a != 0, and 0 for a == 0 (the reference's loop does not run). */
static u128 fe_inverse(u128 a)
{
    if (a == 0) return 0;
    return fe_pow(a, P - 2);
}

static u128 fe_generator(void) { return ((u128)GEN_HI << 64) | GEN_LO; }  /* field.rs:29 */

static u128 fe_primitive_nth_root(u128 n)          /* field.rs:96-107 */
{
    u128 root = fe_generator();
    u128 order = (u128)1 << 119;
    while (order != n) {
        root = fe_mul(root, root);
        order >>= 1;
    }
    return root;
}

/* field.rs:110-116: big-endian integer of all bytes, mod p.  Horner in the
 * field gives the same residue. */
static u128 fe_sample(const uint8_t *bytes, size_t len)
{
    u128 acc = 0;
    for (size_t i = 0; i < len; i++) acc = fe_add(fe_mul(acc, 256), bytes[i]);
    return acc;
}

/* exported scalar helpers ------------------------------------------------- */
void so_modulus(u64 out[2]) { st(out, P); }
void so_generator(u64 out[2]) { st(out, fe_generator()); }
void so_fe_add(const u64 a[2], const u64 b[2], u64 o[2]) { st(o, fe_add(ld(a), ld(b))); }
void so_fe_sub(const u64 a[2], const u64 b[2], u64 o[2]) { st(o, fe_sub(ld(a), ld(b))); }
void so_fe_mul(const u64 a[2], const u64 b[2], u64 o[2]) { oracle_init(); st(o, fe_mul(ld(a), ld(b))); }
void so_fe_pow(const u64 a[2], const u64 e[2], u64 o[2]) { oracle_init(); st(o, fe_pow(ld(a), ld(e))); }
void so_fe_inverse(const u64 a[2], u64 o[2]) { oracle_init(); st(o, fe_inverse(ld(a))); }
int so_primitive_nth_root(u64 n, u64 o[2])
{
    oracle_init();
    if (n == 0 || (n & (n - 1))) return -1;
    st(o, fe_primitive_nth_root(n));
    return 0;
}
void so_fe_sample(const uint8_t *b, size_t len, u64 o[2]) { oracle_init(); st(o, fe_sample(b, len)); }
void so_fe_mul_vec(const u64 *a, const u64 *b, u64 *o, size_t n)
{
    oracle_init();
    for (size_t i = 0; i < n; i++) st(o + 2 * i, fe_mul(ld(a + 2 * i), ld(b + 2 * i)));
}

/* ------------------------------------------------------------------------ */
/* ntt.rs                                                                    */
/* ------------------------------------------------------------------------ */
static void bit_reverse(u128 *a, size_t n)         /* ntt.rs:74-88 */
{
    size_t j = 0;
    for (size_t i = 1; i < n; i++) {
        size_t bit = n >> 1;
        while (j & bit) { j ^= bit; bit >>= 1; }
        j ^= bit;
        if (i < j) { u128 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* the same permutation with every index reversed by itself, so that threads can share the loop */
static void bit_reverse_mt(u128 *a, size_t n)
{
    int bits = 0;
    while (((size_t)1 << bits) < n) bits++;
    PAR_FOR
    for (size_t i = 1; i < n; i++) {
        size_t j = 0;
        for (int b = 0; b < bits; b++) j |= ((i >> b) & 1) << (bits - 1 - b);
        if (i < j) { u128 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

static u128 fe_pow(u128 base, u128 exp);

/* ntt.rs:43-71.  Same loop nest and the same running-product twiddle; the
 * twiddle is kept in Montgomery form so each butterfly costs one reduction.
 * With more than one thread a stage is cut into runs of butterflies, each
 * starting its running product at the power it would have reached. */
#define NTT_RUN 2048
static void ntt_core(u128 *a, size_t n, u128 omega)
{
    if (n <= 1) return;
    if (g_threads > 1) bit_reverse_mt(a, n); else bit_reverse(a, n);
    u128 one_m = to_mont(1);
    for (size_t len = 2; len <= n; len <<= 1) {
        u128 w = fe_pow(omega, n / len);
        u128 w_m = to_mont(w);
        size_t half = len >> 1;
        if (g_threads > 1 && half >= NTT_RUN) {
            size_t runs_per_block = half / NTT_RUN, items = (n / len) * runs_per_block;
            PAR_FOR
            for (size_t t = 0; t < items; t++) {
                size_t i = (t / runs_per_block) * len, j0 = (t % runs_per_block) * NTT_RUN;
                u128 wn_m = to_mont(fe_pow(w, j0));
                for (size_t j = j0; j < j0 + NTT_RUN; j++) {
                    u128 u = a[i + j];
                    u128 v = mont_mul(a[i + j + half], wn_m);
                    a[i + j] = fe_add(u, v);
                    a[i + j + half] = fe_sub(u, v);
                    wn_m = mont_mul(wn_m, w_m);
                }
            }
            continue;
        }
        PAR_FOR
        for (size_t b = 0; b < n / len; b++) {
            size_t i = b * len;
            u128 wn_m = one_m;
            for (size_t j = 0; j < half; j++) {
                u128 u = a[i + j];
                u128 v = mont_mul(a[i + j + half], wn_m);
                a[i + j] = fe_add(u, v);
                a[i + j + half] = fe_sub(u, v);
                wn_m = mont_mul(wn_m, w_m);
            }
        }
    }
}

/* NTT::forward / NTT::inverse (ntt.rs:25-40), in place, natural order. */
int so_ntt(u64 *data, size_t n, int inverse)
{
    oracle_init();
    if (n == 0 || (n & (n - 1))) return -1;      /* "NTT size must be power of 2" */
    if (n > ((size_t)1 << 30)) return -2;         /* ntt.rs:16 */
    u128 *a = (u128 *)malloc(n * sizeof(u128));
    if (!a) return -3;
    for (size_t i = 0; i < n; i++) a[i] = ld(data + 2 * i);
    u128 omega = fe_primitive_nth_root(n);
    if (inverse) {
        ntt_core(a, n, fe_inverse(omega));
        u128 n_inv_m = to_mont(fe_inverse((u128)n));
        PAR_FOR
        for (size_t i = 0; i < n; i++) a[i] = mont_mul(a[i], n_inv_m);
    } else {
        ntt_core(a, n, omega);
    }
    for (size_t i = 0; i < n; i++) st(data + 2 * i, a[i]);
    free(a);
    return 0;
}

/* NTT::evaluate (ntt.rs:101-107): pad or truncate to n, forward. */
int so_ntt_evaluate(const u64 *coeffs_low_first, size_t n_coeffs, size_t n, u64 *out)
{
    size_t m = n_coeffs < n ? n_coeffs : n;
    memcpy(out, coeffs_low_first, m * 16);
    memset(out + 2 * m, 0, (n - m) * 16);
    return so_ntt(out, n, 0);
}

/* Coset evaluation: out[i] = sum_j c_j (offset*omega_n^i)^j, the value the
 * reference obtains by Horner per point (univariate_poly.rs:53) for a shifted
 * domain; computed as the NTT of c_j*offset^j. */
int so_coset_lde(const u64 *coeffs_low_first, size_t n_coeffs, size_t n_out,
                 const u64 offset[2], u64 *out)
{
    oracle_init();
    if (n_coeffs > n_out) return -4;
    u128 off_m = to_mont(ld(offset));
    u128 s = 1;
    for (size_t j = 0; j < n_coeffs; j++) {
        st(out + 2 * j, fe_mul(ld(coeffs_low_first + 2 * j), s));
        s = mont_mul(s, off_m);
    }
    memset(out + 2 * n_coeffs, 0, (n_out - n_coeffs) * 16);
    return so_ntt(out, n_out, 0);
}

/* Horner per point, literally univariate_poly.rs:33-41,53 -- O(n*deg), small
 * sizes only. */
int so_coset_lde_horner(const u64 *coeffs_low_first, size_t n_coeffs, size_t n_out,
                        const u64 offset[2], u64 *out)
{
    oracle_init();
    u128 omega = fe_primitive_nth_root(n_out);
    u128 x = ld(offset);
    for (size_t i = 0; i < n_out; i++) {
        u128 val = 0;
        for (size_t j = n_coeffs; j-- > 0;) val = fe_add(fe_mul(val, x), ld(coeffs_low_first + 2 * j));
        st(out + 2 * i, val);
        x = fe_mul(x, omega);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* BLAKE2b (RFC 7693), unkeyed -- the blake2 0.10.6 crate's Blake2b<U32>/<U64> */
/* ------------------------------------------------------------------------ */
static const u64 B2_IV[8] = {
    0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
    {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
    {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
    {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
    {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

static inline u64 rotr64(u64 x, int r) { return (x >> r) | (x << (64 - r)); }

static void b2_compress(u64 h[8], const uint8_t block[128], u128 t, int last)
{
    u64 m[16], v[16];
    memcpy(m, block, 128);                 /* little-endian host assumed */
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= (u64)t;
    v[13] ^= (u64)(t >> 64);
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y)                       \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = B2_SIGMA[r];
        B2G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

void so_blake2b(const uint8_t *in, size_t len, uint8_t *out, size_t outlen)
{
    u64 h[8];
    uint8_t block[128];
    for (int i = 0; i < 8; i++) h[i] = B2_IV[i];
    h[0] ^= 0x01010000ULL ^ (u64)outlen;
    u128 t = 0;
    while (len > 128) {
        t += 128;
        b2_compress(h, in, t, 0);
        in += 128; len -= 128;
    }
    memset(block, 0, 128);
    memcpy(block, in, len);
    t += len;
    b2_compress(h, block, t, 1);
    memcpy(out, h, outlen);
}

/* ------------------------------------------------------------------------ */
/* SHAKE256 (FIPS 202) -- the sha3 0.10.8 crate's Shake256                   */
/* ------------------------------------------------------------------------ */
static const u64 KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14,
                                   27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4,
                                   15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
static inline u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }

static void keccak_f(u64 s[25])
{
    for (int round = 0; round < 24; round++) {
        u64 bc[5], t;
        for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
        }
        t = s[1];
        for (int i = 0; i < 24; i++) {
            int j = KECCAK_PIL[i];
            u64 b = s[j];
            s[j] = rotl64(t, KECCAK_ROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = s[j + i];
            for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        s[0] ^= KECCAK_RC[round];
    }
}

void so_shake256(const uint8_t *in, size_t len, uint8_t *out, size_t outlen)
{
    const size_t rate = 136;
    u64 s[25];
    uint8_t *sb = (uint8_t *)s;
    memset(s, 0, sizeof s);
    while (len >= rate) {
        for (size_t i = 0; i < rate; i++) sb[i] ^= in[i];
        keccak_f(s);
        in += rate; len -= rate;
    }
    for (size_t i = 0; i < len; i++) sb[i] ^= in[i];
    sb[len] ^= 0x1F;
    sb[rate - 1] ^= 0x80;
    keccak_f(s);
    while (outlen) {
        size_t k = outlen < rate ? outlen : rate;
        memcpy(out, sb, k);
        out += k; outlen -= k;
        if (outlen) keccak_f(s);
    }
}

/* ------------------------------------------------------------------------ */
/* Leaf bytes: bincode::serialize(&FieldElement) (fri.rs:119, stark.rs:302)  */
/* ------------------------------------------------------------------------ */
size_t so_bincode_fe(const u64 v[2], uint8_t out[25])
{
    uint32_t d[4] = {(uint32_t)v[0], (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32)};
    u64 nd = 4;
    while (nd > 0 && d[nd - 1] == 0) nd--;
    out[0] = nd ? 0x01 : 0x00;                    /* Sign as i8: NoSign=0, Plus=1 */
    memcpy(out + 1, &nd, 8);                      /* u64 LE digit count */
    memcpy(out + 9, d, 4 * nd);                   /* u32 LE digits, least significant first */
    return 9 + 4 * (size_t)nd;
}

/* ------------------------------------------------------------------------ */
/* merkle.rs                                                                 */
/* ------------------------------------------------------------------------ */
/* Tree storage used by this oracle and mirrored by the device library: levels
 * concatenated, leaf digests first: level 0 has n digests at offset 0, level
 * l has n>>l digests at offset (2n - (2n>>l)), root is digest index 2n-2. */
static void merkle_reduce(uint8_t *tree, size_t n)
{
    uint8_t *cur = tree;
    for (size_t m = n; m > 1; m >>= 1) {
        uint8_t *nxt = cur + m * 32;
        PAR_FOR
        for (size_t i = 0; i < m / 2; i++) so_blake2b(cur + 64 * i, 64, nxt + 32 * i, 32);   /* merkle.rs:29-32 */
        cur = nxt;
    }
}

/* Retained tree over field-element leaves; tree must hold (2n-1)*32 bytes. */
int so_merkle_tree_fe(const u64 *elems, size_t n, uint8_t *tree)
{
    if (n == 0 || (n & (n - 1))) return -1;       /* "length must be power of two" */
    PAR_FOR
    for (size_t i = 0; i < n; i++) {
        uint8_t leaf[25];
        size_t len = so_bincode_fe(elems + 2 * i, leaf);
        so_blake2b(leaf, len, tree + 32 * i, 32);                /* merkle.rs:41-47 */
    }
    merkle_reduce(tree, n);
    return 0;
}

/* Merkle::commit over bincode(FieldElement) leaves (fri.rs:119). */
int so_merkle_commit_fe(const u64 *elems, size_t n, uint8_t root[32])
{
    if (n == 0 || (n & (n - 1))) return -1;
    uint8_t *tree = (uint8_t *)malloc((2 * n - 1) * 32);
    if (!tree) return -3;
    so_merkle_tree_fe(elems, n, tree);
    memcpy(root, tree + (2 * n - 2) * 32, 32);
    free(tree);
    return 0;
}

/* Merkle::commit(&Vec<Vec<u8>>) (merkle.rs:38-51): leaf i is
 * flat[offsets[i] .. offsets[i+1]). */
int so_merkle_tree_bytes(const uint8_t *flat, const u64 *offsets, size_t n, uint8_t *tree)
{
    if (n == 0 || (n & (n - 1))) return -1;
    for (size_t i = 0; i < n; i++)
        so_blake2b(flat + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), tree + 32 * i, 32);
    merkle_reduce(tree, n);
    return 0;
}

int so_merkle_commit_bytes(const uint8_t *flat, const u64 *offsets, size_t n, uint8_t root[32])
{
    if (n == 0 || (n & (n - 1))) return -1;
    uint8_t *tree = (uint8_t *)malloc((2 * n - 1) * 32);
    if (!tree) return -3;
    so_merkle_tree_bytes(flat, offsets, n, tree);
    memcpy(root, tree + (2 * n - 2) * 32, 32);
    free(tree);
    return 0;
}

/* Merkle::open (merkle.rs:54-93): path[0] = sibling leaf digest, last =
 * sibling of the root's child; log2(n) digests. */
int so_merkle_open(const uint8_t *tree, size_t n, size_t index, uint8_t *path)
{
    if (n < 2 || (n & (n - 1))) return -1;
    if (index >= n) return -5;                    /* "cannot open invalid index" */
    const uint8_t *cur = tree;
    size_t k = 0;
    for (size_t m = n; m > 1; m >>= 1) {
        memcpy(path + 32 * k++, cur + 32 * (index ^ 1), 32);
        index >>= 1;
        cur += m * 32;
    }
    return (int)k;
}

/* Merkle::verify (merkle.rs:96-132). */
int so_merkle_verify(const uint8_t root[32], size_t index, const uint8_t *path, size_t depth,
                     const uint8_t *leaf, size_t leaf_len)
{
    uint8_t node[32], buf[64];
    if (depth == 0 || index >= ((size_t)1 << depth)) return -5;
    so_blake2b(leaf, leaf_len, node, 32);
    for (size_t k = 0; k < depth; k++) {
        if ((index & 1) == 0) { memcpy(buf, node, 32); memcpy(buf + 32, path + 32 * k, 32); }
        else { memcpy(buf, path + 32 * k, 32); memcpy(buf + 32, node, 32); }
        so_blake2b(buf, 64, node, 32);
        index >>= 1;
    }
    return memcmp(node, root, 32) == 0;
}

/* ------------------------------------------------------------------------ */
/* proof_stream.rs                                                           */
/* ------------------------------------------------------------------------ */
typedef struct {
    char **objs;
    size_t *lens;
    size_t count, cap;
} so_stream;

so_stream *so_ps_new(void) { return (so_stream *)calloc(1, sizeof(so_stream)); }
void so_ps_free(so_stream *ps)
{
    if (!ps) return;
    for (size_t i = 0; i < ps->count; i++) free(ps->objs[i]);
    free(ps->objs); free(ps->lens); free(ps);
}
void so_ps_push(so_stream *ps, const char *s, size_t len)          /* proof_stream.rs:23-25 */
{
    if (ps->count == ps->cap) {
        ps->cap = ps->cap ? ps->cap * 2 : 16;
        ps->objs = (char **)realloc(ps->objs, ps->cap * sizeof(char *));
        ps->lens = (size_t *)realloc(ps->lens, ps->cap * sizeof(size_t));
    }
    ps->objs[ps->count] = (char *)malloc(len + 1);
    memcpy(ps->objs[ps->count], s, len);
    ps->objs[ps->count][len] = 0;
    ps->lens[ps->count++] = len;
}
size_t so_ps_count(const so_stream *ps) { return ps->count; }
size_t so_ps_object_len(const so_stream *ps, size_t i) { return ps->lens[i]; }
const char *so_ps_object(const so_stream *ps, size_t i) { return ps->objs[i]; }

/* bincode(Vec<String>): u64 count, then u64 length + bytes per string
 * (proof_stream.rs:36-38).  Returns the size; writes when out != NULL. */
size_t so_ps_serialize(const so_stream *ps, size_t upto, uint8_t *out)
{
    size_t sz = 8;
    for (size_t i = 0; i < upto; i++) sz += 8 + ps->lens[i];
    if (!out) return sz;
    u64 c = upto;
    memcpy(out, &c, 8);
    size_t off = 8;
    for (size_t i = 0; i < upto; i++) {
        u64 l = ps->lens[i];
        memcpy(out + off, &l, 8); off += 8;
        memcpy(out + off, ps->objs[i], ps->lens[i]); off += ps->lens[i];
    }
    return sz;
}

/* prover_fiat_shamir (proof_stream.rs:50-58). */
void so_ps_fiat_shamir(const so_stream *ps, uint8_t *out, size_t nbytes)
{
    size_t sz = so_ps_serialize(ps, ps->count, NULL);
    uint8_t *buf = (uint8_t *)malloc(sz);
    so_ps_serialize(ps, ps->count, buf);
    so_shake256(buf, sz, out, nbytes);
    free(buf);
}

/* ------------------------------------------------------------------------ */
/* fri.rs                                                                    */
/* ------------------------------------------------------------------------ */
size_t so_fri_num_rounds(size_t domain_length, size_t expansion_factor, size_t num_colinearity_tests)
{                                                  /* fri.rs:39-51 */
    size_t len = domain_length, rounds = 0;
    while (len > expansion_factor && 4 * num_colinearity_tests < len) { len /= 2; rounds++; }
    return rounds;
}

/* One split-and-fold (fri.rs:133-138):
 *   new[i] = 2^-1 * ((1 + alpha/x_i) cw[i] + (1 - alpha/x_i) cw[N/2+i]),  x_i = offset*omega^i.
 * Evaluated as 2^-1 (a+b) + (2^-1 alpha offset^-1) omega^-i (a-b) with omega^-i
 * as a running product: same field value, no per-output inversion. */
void so_fri_fold(const u64 *cw, size_t N, const u64 alpha[2], const u64 offset[2],
                 const u64 omega[2], u64 *out)
{
    oracle_init();
    size_t half = N / 2;
    u128 two_inv = fe_inverse(2);
    u128 two_inv_m = to_mont(two_inv);
    u128 c_m = to_mont(fe_mul(fe_mul(two_inv, ld(alpha)), fe_inverse(ld(offset))));
    u128 winv_m = to_mont(fe_inverse(ld(omega)));
    u128 winv = fe_inverse(ld(omega));
    /* runs of outputs, each starting omega^-i at its own power (one run per 4096 outputs) */
    size_t run = 4096, runs = (half + run - 1) / run;
    PAR_FOR
    for (size_t t = 0; t < runs; t++) {
        size_t lo = t * run, hi = lo + run < half ? lo + run : half;
        u128 wi_m = to_mont(fe_pow(winv, lo));     /* omega^-i, Montgomery form */
        for (size_t i = lo; i < hi; i++) {
            u128 a = ld(cw + 2 * i), b = ld(cw + 2 * (half + i));
            u128 s = mont_mul(fe_add(a, b), two_inv_m);
            u128 d = mont_mul(fe_sub(a, b), mont_mul(wi_m, c_m));
            st(out + 2 * i, fe_add(s, d));
            wi_m = mont_mul(wi_m, winv_m);
        }
    }
}

static size_t u32_to_dec(uint32_t v, char *out)
{
    char tmp[12]; size_t n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (size_t i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    return n;
}

/* serde_json of Vec<FieldElement>: [{"value":[1,[d0,d1,..]]},...] (fri.rs:146). */
static char *json_field_vec(const u128 *v, size_t n, size_t *len_out)
{
    char *buf = (char *)malloc(n * 72 + 4);
    size_t o = 0;
    buf[o++] = '[';
    for (size_t i = 0; i < n; i++) {
        if (i) buf[o++] = ',';
        uint32_t d[4] = {(uint32_t)v[i], (uint32_t)(v[i] >> 32), (uint32_t)(v[i] >> 64), (uint32_t)(v[i] >> 96)};
        int nd = 4;
        while (nd > 0 && d[nd - 1] == 0) nd--;
        o += (size_t)sprintf(buf + o, "{\"value\":[%d,[", nd ? 1 : 0);
        for (int k = 0; k < nd; k++) {
            if (k) buf[o++] = ',';
            o += u32_to_dec(d[k], buf + o);
        }
        buf[o++] = ']'; buf[o++] = ']'; buf[o++] = '}';
    }
    buf[o++] = ']';
    buf[o] = 0;
    *len_out = o;
    return buf;
}

/* Fri::commit (fri.rs:100-152).  layers_out receives every codeword the
 * reference returns, concatenated (N + N/2 + ... elements); roots_out gets
 * num_rounds 32-byte roots; alphas_out gets num_rounds-1 elements.  The proof
 * stream receives exactly the reference's pushes (hex roots, last codeword
 * JSON).  Returns the number of rounds or a negative error. */
int so_fri_commit(const u64 *codeword, size_t N, const u64 offset_in[2], const u64 omega_in[2],
                  size_t expansion_factor, size_t num_colinearity_tests, so_stream *ps,
                  u64 *layers_out, uint8_t *roots_out, u64 *alphas_out)
{
    oracle_init();
    size_t rounds = so_fri_num_rounds(N, expansion_factor, num_colinearity_tests);
    if (rounds == 0) return -6;                   /* "Cannot do FRI with less than 1 round" */
    u128 omega = ld(omega_in), offset = ld(offset_in);
    u64 *cur = (u64 *)malloc(N * 16);
    u64 *nxt = (u64 *)malloc(N * 8 + 16);
    memcpy(cur, codeword, N * 16);
    size_t n = N, out_off = 0;
    static const char hexd[] = "0123456789abcdef";
    for (size_t r = 0; r < rounds; r++) {
        if (fe_pow(omega, n - 1) != fe_inverse(omega)) { free(cur); free(nxt); return -7; }   /* fri.rs:116 */
        uint8_t root[32];
        char hex[64];
        so_merkle_commit_fe(cur, n, root);                                  /* fri.rs:119 */
        for (int i = 0; i < 32; i++) { hex[2 * i] = hexd[root[i] >> 4]; hex[2 * i + 1] = hexd[root[i] & 15]; }
        so_ps_push(ps, hex, 64);                                            /* fri.rs:120 */
        if (roots_out) memcpy(roots_out + 32 * r, root, 32);
        if (r == rounds - 1) break;
        uint8_t fs[32];
        so_ps_fiat_shamir(ps, fs, 32);                                      /* fri.rs:126 */
        u128 alpha = fe_sample(fs, 32);                                     /* fri.rs:127 */
        if (alphas_out) st(alphas_out + 2 * r, alpha);
        if (layers_out) { memcpy(layers_out + 2 * out_off, cur, n * 16); out_off += n; }   /* fri.rs:130 */
        u64 a2[2], o2[2], w2[2];
        st(a2, alpha); st(o2, offset); st(w2, omega);
        so_fri_fold(cur, n, a2, o2, w2, nxt);                               /* fri.rs:133-139 */
        n /= 2;
        u64 *t = cur; cur = nxt; nxt = t;
        omega = fe_mul(omega, omega);                                       /* fri.rs:141-142 */
        offset = fe_mul(offset, offset);
    }
    u128 *last = (u128 *)malloc(n * sizeof(u128));
    for (size_t i = 0; i < n; i++) last[i] = ld(cur + 2 * i);
    size_t jl;
    char *js = json_field_vec(last, n, &jl);
    so_ps_push(ps, js, jl);                                                 /* fri.rs:146 */
    free(js); free(last);
    if (layers_out) memcpy(layers_out + 2 * out_off, cur, n * 16);          /* fri.rs:149 */
    free(cur); free(nxt);
    return (int)rounds;
}

/* Fri::sample_index (fri.rs:81-87): usize accumulator, so only the last 8
 * bytes survive. */
u64 so_fri_sample_index(const uint8_t *bytes, size_t len, u64 size)
{
    u64 acc = 0;
    for (size_t i = 0; i < len; i++) acc = (acc << 8) ^ bytes[i];
    return acc % size;
}

/* Fri::sample_indices (fri.rs:54-78). */
int so_fri_sample_indices(const uint8_t *seed, size_t seed_len, u64 size, u64 reduced_size,
                          size_t number, u64 *indices_out)
{
    u64 *reduced = (u64 *)malloc(number * sizeof(u64));
    uint8_t *buf = (uint8_t *)malloc(seed_len + 8);
    memcpy(buf, seed, seed_len);
    size_t have = 0;
    u64 counter = 0;
    while (have < number) {
        uint8_t h[64];
        memcpy(buf + seed_len, &counter, 8);
        so_blake2b(buf, seed_len + 8, h, 64);
        u64 index = so_fri_sample_index(h, 64, size);
        u64 red = index % reduced_size;
        counter++;
        int dup = 0;
        for (size_t i = 0; i < have; i++) if (reduced[i] == red) { dup = 1; break; }
        if (!dup) { indices_out[have] = index; reduced[have] = red; have++; }
    }
    free(reduced); free(buf);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Synthetic data: SplitMix64, two words per candidate, rejection below p.    */
/* ------------------------------------------------------------------------ */
void so_splitmix_field_vector(u64 *out, size_t n, u64 seed)
{
    u64 state = seed;
    size_t have = 0;
    while (have < n) {
        u64 w[2];
        for (int k = 0; k < 2; k++) {
            state += 0x9E3779B97F4A7C15ULL;
            u64 z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            w[k] = z ^ (z >> 31);
        }
        u128 v = ((u128)w[1] << 64) | w[0];
        if (v < P) { out[2 * have] = w[0]; out[2 * have + 1] = w[1]; have++; }
    }
}
