// NTT kernels for gfx950 (reference semantics: src/modules/ntt.rs:25-71 --
// natural order in, natural order out, X[k] = sum_j x[j] w^(jk)).
//
// Large transforms are a mixed-radix decimation-in-time factorisation
// n = R_0 R_1 ... R_{P-1} (R_q = 2^5..2^8) run as P global passes.  A workgroup
// owns a tile of 4096 elements = R_q sub-transform points x C columns, held in
// registers (16 elements per thread).  Each pass is two register rounds
// (radix R_q/16, then radix 16) with one exchange through LDS.  Bit reversal
// never happens as a data movement: digit reversal is folded into the pass
// addressing, and the last pass writes its result transposed into natural
// order.  Twiddles are read from small HBM/L2-resident tables in Montgomery
// form; the coset shift of a low-degree extension is applied to the loaded
// coefficients of the first pass, and the padding zeros are never read.
#include <hip/hip_runtime.h>


#include "field5.hpp"
#include "internal.hpp"
#include "tw16_tables.hpp"

namespace starkk {

using namespace starkf;

__device__ __forceinline__ constexpr int brev_c(int k, int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((k >> i) & 1) << (bits - 1 - i);
    return r;
}

// In-register decimation-in-frequency transform of M = 2^LOGM lazy elements at
// v[BASE .. BASE+M): natural order in, bit-reversed order out (output k sits in
// register BASE + brev(k)).  Adds and subtracts are carry-free limb operations;
// only the twiddle products reduce.  The multipliers are powers of the 16th root of unity (of its
// inverse when INV), constants of the field: the exponent is known after unrolling, so each product
// reads its tabulated multiplier (TW16C) as instruction literals.
template <int LOGM, int BASE, bool INV>
__device__ __forceinline__ void ntt_dif_reg(fe5 (&v)[16])
{
    constexpr int M = 1 << LOGM;
#pragma unroll
    for (int lg = LOGM; lg >= 1; lg--) {
        const int len = 1 << lg;
        const int half = len >> 1;
#pragma unroll
        for (int s = 0; s < M; s += len) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const fe5 a = v[BASE + s + j];
                const fe5 b = v[BASE + s + j + half];
                v[BASE + s + j] = add5(a, b);
                const fe5 d = sub5(a, b);
                const int e = j * (16 / len);
                v[BASE + s + j + half] = (e == 0) ? d : mont_mul5_c<false>(d, TW16C[INV ? 1 : 0][e]);
            }
        }
    }
}

template <int LOGM, int G, bool INV>
struct RoundOne {
    __device__ __forceinline__ static void run(fe5 (&v)[16])
    {
        ntt_dif_reg<LOGM, (G - 1) * (1 << LOGM), INV>(v);
        RoundOne<LOGM, G - 1, INV>::run(v);
    }
};
template <int LOGM, bool INV>
struct RoundOne<LOGM, 0, INV> {
    __device__ __forceinline__ static void run(fe5 (&)[16]) {}
};

// memory prefix (k_0 most significant) -> K = k_0 + R_0 k_1 + ...
__device__ __forceinline__ uint32_t prefix_to_k(uint32_t pre, const PassArgs& a)
{
    uint32_t K = 0;
    uint32_t below = 0;
    for (uint32_t u = 0; u < a.nprev; u++) below += a.rprev[u];
    for (int u = (int)a.nprev - 1; u >= 0; u--) {
        below -= a.rprev[u];
        uint32_t d = pre & ((1u << a.rprev[u]) - 1);
        pre >>= a.rprev[u];
        K |= d << below;
    }
    return K;
}

// K = k_0 + R_0 k_1 + ... -> memory prefix
__device__ __forceinline__ uint32_t k_to_prefix(uint32_t K, const PassArgs& a)
{
    uint32_t pre = 0;
    for (uint32_t u = 0; u < a.nprev; u++) {
        uint32_t d = K & ((1u << a.rprev[u]) - 1);
        K >>= a.rprev[u];
        pre = (pre << a.rprev[u]) | d;
    }
    return pre;
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also drains the
// vector-memory counter (vmcnt(0)); only LDS traffic has to be ordered here,
// so loads and stores in flight may stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// lazy element -> canonical 128-bit value (the slow path is rare: it is taken
// when the reduced value came out as p + small or small negative)
// (the sums of up to sixteen products that leave the last pass's butterflies go through finish_lazy of
// field5.hpp instead)
// The test is taken for the whole wave at once (two compares into scalar masks and a scalar branch): the
// per-lane form costs five vector instructions and an exec-mask round trip per element for a path that a
// wave takes about once in 2^14 elements.
__device__ __forceinline__ fe finish(const fe5& x)
{
    fe o = pack_fast(x);
    constexpr int UGT = 34, UGE = 35;                          // unsigned >, >= (LLVM's integer predicates)
    const uint64_t some = __builtin_amdgcn_uicmp((uint32_t)x.l[0], (uint32_t)LIMB_MASK, UGT) |
                          __builtin_amdgcn_uicmp((uint32_t)x.l[4], (uint32_t)P4, UGE);
    if (__builtin_expect(some != 0, 0)) {
        if (noncanonical5(x)) o = reduce5_slow(x);
    }
    return o;
}

// Which transform of the batch, and which column of it in this pass (row in PASS_LAST), a tile's column c is.
// Transforms of at least 4096 points own whole tiles: the transform index is the same for the whole
// workgroup.  Shorter ones (SMALL) share a tile, 4096 / n of them side by side, and the index varies
// with c; a tile may then reach past the end of the batch.
template <int RA, bool SMALL>
__device__ __forceinline__ void locate(const PassArgs& a, uint64_t tile, int c, uint64_t& batch, uint64_t& f)
{
    constexpr int C = 1 << (8 - RA);
    if constexpr (SMALL) {
        const uint32_t cl = a.log_n - a.r;                     // log2 of the columns of one transform in this pass
        const uint64_t fg = tile * C + (uint64_t)c;
        batch = fg >> cl;
        f = fg & (((uint64_t)1 << cl) - 1);
    } else {
        const uint32_t ltiles = a.log_n - 12;
        batch = tile >> ltiles;
        f = (tile & (((uint64_t)1 << ltiles) - 1)) * C + (uint64_t)c;
    }
}

// Last pass of an inverse transform: output k of the forward sums goes to index (n - k) mod n.  A tile's
// rows are whole columns b of the n / R it could own in any order, so it takes the columns one further
// on, b + 1 ... b + C (mod n / R): the C outputs of one k then land on an ALIGNED run of C elements,
// n - (k n/R + b) - C ... - 1, instead of one that straddles two (a 6 % slower pass at 2^24).
template <bool INV>
__device__ __forceinline__ uint64_t last_pass_column(const PassArgs& a, uint64_t b)
{
    if constexpr (INV) return (b + 1) & (((uint64_t)1 << (a.log_n - a.r)) - 1);
    else return b;
}

// A lane's fixed offset, handed out afresh for every access: if the compiler could see that it is the same
// value each time it would add it to the base once and then form "(base + lane offset) + rows" on the vector
// pipes for every access, instead of "(base + rows)" on the scalar unit plus the offset operand of the access.
__device__ __forceinline__ uint32_t lane_offset(uint32_t x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// Where a tile of a strided pass (first or middle, transforms of at least 4096 points) sits: the element
// index of its column 0, row 0, which is the same for the whole workgroup and stays in scalar registers,
// and this lane's distance from it in elements (column c of the tile, row `row` of the sub-transform).
// A lane is never more than 16 rows and one tile of columns away, so its distance fits 32 bits even in
// bytes (log_s <= 24, checked at launch), and an access is "uniform pointer + 32-bit lane offset": the
// form a global load or store takes with its base in scalar registers, with no address arithmetic on the
// vector pipes per access.
struct TileOrigin {
    uint64_t batch;      // which transform of the batch
    uint64_t elem;       // element index of (column 0, row 0) inside that transform
};
template <int RA>
__device__ __forceinline__ TileOrigin tile_origin(const PassArgs& a, uint64_t tile)
{
    TileOrigin o;
    uint64_t f0;
    locate<RA, false>(a, tile, 0, o.batch, f0);
    o.elem = ((f0 >> a.log_s) << (a.r + a.log_s)) + (f0 & (((uint64_t)1 << a.log_s) - 1));
    return o;
}
template <int RA>
__device__ __forceinline__ uint32_t lane_distance(const PassArgs& a, uint64_t tile, const TileOrigin& o, int c, int row)
{
    uint64_t batch, f;
    locate<RA, false>(a, tile, c, batch, f);
    const uint64_t base = ((f >> a.log_s) << (a.r + a.log_s)) + (f & (((uint64_t)1 << a.log_s) - 1));
    return (uint32_t)(base - o.elem) + ((uint32_t)row << a.log_s);
}

// Issue the 16 global loads of one thread for one tile (round-1 layout).
// tile enumerates (batch, tile-in-transform); rows past n_src of a first pass
// are the zero padding and are not read.
template <int RA, int MODE, bool SMALL, bool INV>
__device__ __forceinline__ void load_tile(fe (&raw)[16], const PassArgs& a, uint64_t tile, int t)
{
    constexpr int M1 = 1 << RA;
    constexpr int G1 = 16 / M1;
    constexpr int LOGC = 8 - RA;
    constexpr int C = 1 << LOGC;
#pragma unroll
    for (int g = 0; g < G1; g++) {
        const int q1 = t + TILE_THREADS * g;
        if constexpr (MODE == PASS_LAST) {
            const int c = q1 >> 4, jb = q1 & 15;
            uint64_t batch, b;                                     // b: low part of the output index
            locate<RA, SMALL>(a, tile, c, batch, b);
            b = last_pass_column<INV>(a, b);
            const fe* __restrict__ src = a.src + batch * a.src_batch_stride;
            const uint64_t row = (uint64_t)k_to_prefix((uint32_t)b, a) << a.r;
#pragma unroll
            for (int ja = 0; ja < M1; ja++) {
                if constexpr (SMALL) {
                    raw[g * M1 + ja] = fe_zero();
                    if (batch < a.n_batch) raw[g * M1 + ja] = src[row + ja * 16 + jb];
                } else {
                    raw[g * M1 + ja] = src[row + ja * 16 + jb];
                }
            }
        } else if constexpr (!SMALL) {
            const int jb = q1 >> LOGC, c = q1 & (C - 1);
            const TileOrigin o = tile_origin<RA>(a, tile);
            const uint32_t lane_el = lane_distance<RA>(a, tile, o, c, jb);
            const char* __restrict__ src = (const char*)(a.src + o.batch * a.src_batch_stride + o.elem);
#pragma unroll
            for (int ja = 0; ja < M1; ja++) {
                const uint64_t rows = (uint64_t)(ja * 16) << a.log_s;              // the same for every lane
                const fe* __restrict__ p = (const fe*)(src + rows * sizeof(fe) + lane_offset(lane_el * (uint32_t)sizeof(fe)));
                if constexpr (MODE == PASS_FIRST) {
                    // valid while origin + rows + lane_el < n_src: the bound on lane_el is the same for every lane
                    const uint64_t before = o.elem + rows;
                    const uint64_t room = a.n_src > before ? a.n_src - before : 0;
                    const uint32_t limit = room > 0xffffffffull ? 0xffffffffu : (uint32_t)room;
                    raw[g * M1 + ja] = fe_zero();
                    if (lane_el < limit) raw[g * M1 + ja] = *p;
                } else {
                    raw[g * M1 + ja] = *p;
                }
            }
        } else {
            const int jb = q1 >> LOGC, c = q1 & (C - 1);
            uint64_t batch, f;                                     // f: flat column of this pass
            locate<RA, SMALL>(a, tile, c, batch, f);
            const fe* __restrict__ src = a.src + batch * a.src_batch_stride;
            const uint64_t pre = f >> a.log_s;
            const uint64_t col = f & (((uint64_t)1 << a.log_s) - 1);
            const uint64_t base = (pre << (a.r + a.log_s)) + col;
#pragma unroll
            for (int ja = 0; ja < M1; ja++) {
                const uint64_t idx = base + ((uint64_t)(ja * 16 + jb) << a.log_s);
                if constexpr (MODE == PASS_FIRST || SMALL) {
                    raw[g * M1 + ja] = fe_zero();
                    bool ok = true;
                    if constexpr (MODE == PASS_FIRST) ok = idx < a.n_src;
                    if constexpr (SMALL) ok = ok && batch < a.n_batch;
                    if (ok) raw[g * M1 + ja] = src[idx];
                } else {
                    raw[g * M1 + ja] = src[idx];
                }
            }
        }
    }
}

// Geometry of one pass kernel instantiation.
template <int RA, int MODE>
struct Geo {
    static constexpr int M1 = 1 << RA;          // round-1 radix
    static constexpr int G1 = 16 / M1;          // round-1 groups per thread
    static constexpr int R = 16 * M1;           // sub-transform size of this pass
    static constexpr int LOGC = 8 - RA;
    static constexpr int C = 1 << LOGC;         // columns (rows in PASS_LAST) per tile
    static constexpr int PITCH = (MODE == PASS_LAST) ? (C + 1) : C;
    static constexpr int TWSHIFT = 8 - (RA + 4);
    static constexpr int KA_HALF = M1 / 2;      // the exchange runs in two phases of M1/2 values of ka
    static constexpr int LDS_ELEMS = RA == 0 ? 1 : (R / 2) * PITCH;   // a radix-16 pass has no exchange
};

// canonical loads -> lazy limbs, plus the coset twist of a first pass.  The
// twist factors are gathered four elements ahead of their use so that their
// latency overlaps the multiplications of the previous four.
template <int RA, int MODE, bool TWIST, bool SMALL>
__device__ __forceinline__ void unpack_tile(fe5 (&v)[16], const fe (&raw)[16], const PassArgs& a, uint64_t tile, int t)
{
    using G = Geo<RA, MODE>;
    if constexpr (MODE == PASS_FIRST && TWIST) {
        const tw5* __restrict__ thi = a.twist_hi;
        const tw5* __restrict__ tlo = a.twist_lo;
        constexpr int B = 2;
        tw5 wh[2][B], wl[2][B];
        auto index_of = [&](int i) -> uint64_t {
            const int g = i / G::M1, ja = i % G::M1;
            const int q1 = t + TILE_THREADS * g;
            const int jb = q1 >> G::LOGC, c = q1 & (G::C - 1);
            uint64_t batch, f;
            locate<RA, SMALL>(a, tile, c, batch, f);
            const uint64_t pre = f >> a.log_s;
            const uint64_t col = f & (((uint64_t)1 << a.log_s) - 1);
            const uint64_t base = (pre << (a.r + a.log_s)) + col;
            return base + ((uint64_t)(ja * 16 + jb) << a.log_s);
        };
        auto issue = [&](int bt) {
#pragma clang loop unroll(full)
            for (int i = 0; i < B; i++) {
                const uint64_t idx = index_of(bt * B + i);
                if (idx < a.n_src) {
                    wh[bt & 1][i] = thi[idx >> 8];
                    wl[bt & 1][i] = tlo[idx & 255];
                }
            }
        };
        issue(0);
#pragma clang loop unroll(full)
        for (int bt = 0; bt < 16 / B; bt++) {
            if (bt + 1 < 16 / B) issue(bt + 1);
#pragma clang loop unroll(full)
            for (int i = 0; i < B; i++) {
                const int e = bt * B + i;
                v[e] = unpack5(raw[e]);
                if (index_of(e) < a.n_src)
                    v[e] = mont_mul5_v(v[e], mont_mul5_v(*(const fe5*)&wh[bt & 1][i], wl[bt & 1][i]));
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = unpack5(raw[i]);
    }
}

// round 1 (radix M1 over ja), inner twiddle w_R^(jb*ka), exchange through LDS;
// on return v holds the 16 round-2 inputs of this thread.  The exchange runs in
// two phases, each carrying the half of the tile with ka in one half of its
// range, so the buffer is half a tile and three workgroups fit a CU.  Elements
// cross LDS as the five lazy limbs a multiplication leaves (20 bytes), which is
// cheaper than packing to 16 bytes and unpacking again.
// A thread's 16 inputs all have ka = t >> LOGC, so waves 0-1 collect theirs in
// phase 0 and waves 2-3 in phase 1.  Ends with every thread's LDS reads issued
// and waited for, but NOT with a barrier.
template <int RA, int MODE>
__device__ __forceinline__ void round1_exchange(fe5 (&v)[16], fe5* lds, const tw5* s_tw256, int t)
{
    using G = Geo<RA, MODE>;
    RoundOne<RA, G::G1, false>::run(v);
    const int ka2 = t >> G::LOGC, c2 = t & (G::C - 1);
    const int my_phase = ka2 / G::KA_HALF;
    fe5 rcv[16];
#pragma clang loop unroll(full)
    for (int ph = 0; ph < 2; ph++) {
        if (ph > 0) lds_barrier();          // phase 0 fully read before it is overwritten
        int tw_ = t;                        // LDS addresses are formed per phase rather than kept in registers
        asm volatile("" : "+v"(tw_));
#pragma clang loop unroll(full)
        for (int g = 0; g < G::G1; g++) {
            const int q1 = tw_ + TILE_THREADS * g;
            int jb, c;
            if constexpr (MODE == PASS_LAST) { c = q1 >> 4; jb = q1 & 15; }
            else { jb = q1 >> G::LOGC; c = q1 & (G::C - 1); }
#pragma clang loop unroll(full)
            for (int kh = 0; kh < G::KA_HALF; kh++) {
                const int ka = ph * G::KA_HALF + kh;
                const int reg = g * G::M1 + brev_c(ka, RA);
                lds[(kh * 16 + jb) * G::PITCH + c] = mont_mul5_v<false>(v[reg], s_tw256[(jb * ka) << G::TWSHIFT]);
            }
        }
        lds_barrier();
        if (my_phase == ph) {
#pragma clang loop unroll(full)
            for (int jb = 0; jb < 16; jb++) rcv[jb] = lds[((ka2 % G::KA_HALF) * 16 + jb) * G::PITCH + c2];
        }
    }
#pragma clang loop unroll(full)
    for (int jb = 0; jb < 16; jb++) v[jb] = rcv[jb];
}

// round 2 (radix 16 over jb), outgoing twiddle, store
// NEXT_LAST: 0 = the next pass is not the last, 1 = it is and the twiddle is formed as hi*lo, 2 = it is
// and the twiddle comes from the plane
// INV (last pass only): the transform is an inverse one.  It runs as the forward transform with the
// forward root's tables, and output k is stored at index (n - k) mod n: sum_j x[j] w^(-jk) is the forward
// sum at -k (ntt.rs:19,30-39; the factor n^-1 rides on the exchange twiddles of this pass).
template <int RA, int MODE, int NEXT_LAST, bool SMALL, bool INV>
__device__ __forceinline__ void round2_store(fe5 (&v)[16], const PassArgs& a, uint64_t tile, int t)
{
    using G = Geo<RA, MODE>;
    constexpr int M1 = G::M1;
    // The store addresses are recomputed here from values the compiler cannot
    // tie to the load addresses of the start of the pass; otherwise it keeps 16
    // address pairs alive across both rounds and spills some of them.
    asm volatile("" : "+v"(t), "+s"(tile));
    const int ka2 = t >> G::LOGC, c2 = t & (G::C - 1);
    uint64_t batch, f;
    locate<RA, SMALL>(a, tile, c2, batch, f);
    fe* __restrict__ dst = a.dst + batch * a.dst_batch_stride;
    bool live = true;                                      // a shared tile may reach past the end of the batch
    if constexpr (SMALL) live = batch < a.n_batch;
    if constexpr (MODE == PASS_LAST) {
        ntt_dif_reg<4, 0, false>(v);
        const uint64_t b = last_pass_column<INV>(a, f);
        const uint32_t sh = a.log_n - a.r;
        if constexpr (!SMALL) {
            // Output kb goes to index ((ka2 + M1 kb) << sh) + b: this lane's part (ka2 << sh) + b is below
            // n / 16 + n / R, the rest is the same for every lane, so a store is a scalar base plus a fixed
            // 32-bit lane offset.  Reversed (INV), index i becomes n - i: the lane part is counted down from a
            // bound above it, M1 << sh, which leaves the scalar part n - ((M1 (kb + 1)) << sh) >= 0; index 0
            // stays index 0 and can only be output 0 of a thread.
            char* __restrict__ out = (char*)dst;
            const uint32_t lane_el = ((uint32_t)ka2 << sh) + (uint32_t)b;
            const uint32_t lane_b = (INV ? ((uint32_t)M1 << sh) - lane_el : lane_el) * (uint32_t)sizeof(fe);
#pragma unroll
            for (int kb = 0; kb < 16; kb++) {
                const fe o16 = finish_lazy(v[brev_c(kb, 4)]);
                if constexpr (!INV) {
                    *(fe*)(out + (((uint64_t)(M1 * kb) << sh) * sizeof(fe)) + lane_offset(lane_b)) = o16;
                } else if (kb == 0) {
                    dst[(0 - (uint64_t)lane_el) & (((uint64_t)1 << a.log_n) - 1)] = o16;
                } else {
                    const uint64_t top = ((uint64_t)1 << a.log_n) - ((uint64_t)(M1 * (kb + 1)) << sh);
                    *(fe*)(out + top * sizeof(fe) + lane_offset(lane_b)) = o16;
                }
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < 16; kb++) {
                uint32_t k = ka2 + M1 * kb;
                asm volatile("" : "+v"(k));
                const fe o = finish_lazy(v[brev_c(kb, 4)]);
                uint64_t idx = ((uint64_t)k << sh) + b;
                if constexpr (INV) idx = (0 - idx) & (((uint64_t)1 << a.log_n) - 1);
                if (live) dst[idx] = o;
            }
        }
    } else if constexpr (!SMALL && NEXT_LAST != 1) {
        // Output kb of this thread is element origin + lane_el + ((M1 kb) << log_s) of the transform: the last
        // term is the same for every lane, so the store (and the plane load next to it) is a scalar base plus
        // this lane's fixed 32-bit offset, and the exponent of a tabulated twiddle advances by a fixed step.
        const TileOrigin o = tile_origin<RA>(a, tile);
        const uint32_t lane_el = lane_distance<RA>(a, tile, o, c2, ka2);
        const uint32_t lane_b = lane_el * (uint32_t)sizeof(fe);
        char* __restrict__ out = (char*)(a.dst + o.batch * a.dst_batch_stride + o.elem);
        auto rows_of = [&](int kb) { return ((uint64_t)(M1 * kb) << a.log_s); };
        if constexpr (NEXT_LAST == 2) {
            // the twiddle of output element i comes ready-made from the plane: a 20-byte load
            // next to the store instead of two gathers and a multiplication
            const char* __restrict__ plane = (const char*)(a.plane + 5 * o.elem);
            const uint32_t lane_p = lane_el * 20u;
            fe5 w[2];
            auto issue = [&](int kb) { w[kb & 1] = *(const fe5*)(plane + rows_of(kb) * 20 + lane_offset(lane_p)); };
            ntt_dif_reg<4, 0, false>(v);
            issue(0);
#pragma clang loop unroll(full)
            for (int kb = 0; kb < 16; kb++) {
                if (kb + 1 < 16) issue(kb + 1);
                const fe o16 = finish(mont_mul5_v(v[brev_c(kb, 4)], w[kb & 1]));
                *(fe*)(out + rows_of(kb) * sizeof(fe) + lane_offset(lane_b)) = o16;
            }
        } else if constexpr (NEXT_LAST == 3) {
            // the same from the plane of canonical 16-byte values
            const char* __restrict__ plane16 = (const char*)(a.plane16 + o.elem);
            fe w[2];
            auto issue = [&](int kb) { w[kb & 1] = *(const fe*)(plane16 + rows_of(kb) * sizeof(fe) + lane_offset(lane_b)); };
            ntt_dif_reg<4, 0, false>(v);
            issue(0);
#pragma clang loop unroll(full)
            for (int kb = 0; kb < 16; kb++) {
                if (kb + 1 < 16) issue(kb + 1);
                const fe o16 = finish(mont_mul5_v(v[brev_c(kb, 4)], unpack5(w[kb & 1])));
                *(fe*)(out + rows_of(kb) * sizeof(fe) + lane_offset(lane_b)) = o16;
            }
        } else {
            // twiddle of the next pass from the hi table: w_n^(j_next * K * S_next), K = K_prev + k * 2^bits_prev,
            // k = ka2 + M1 kb: the exponent of output kb is that of output 0 plus kb fixed steps
            const uint32_t pre = (uint32_t)(f >> a.log_s);
            const uint32_t col = (uint32_t)(f & (((uint64_t)1 << a.log_s) - 1));
            const uint32_t j_next = col >> a.log_s_next;
            uint32_t bits_prev = 0;
            for (uint32_t u = 0; u < a.nprev; u++) bits_prev += a.rprev[u];
            const uint32_t k_prev = prefix_to_k(pre, a);
            const uint32_t sn = a.log_s_next - a.h;                // S_next / 2^h
            const char* __restrict__ hi = (const char*)a.hi;
            uint32_t e_b = ((j_next * (k_prev + ((uint32_t)ka2 << bits_prev))) << sn) * (uint32_t)sizeof(tw5);
            const uint32_t step_b = ((j_next * ((uint32_t)M1 << bits_prev)) << sn) * (uint32_t)sizeof(tw5);
            constexpr int B = 2;
            tw5 w[2][B];
            auto issue = [&](int bt) {
#pragma clang loop unroll(full)
                for (int i = 0; i < B; i++) {
                    w[bt & 1][i] = *(const tw5*)(hi + e_b);
                    e_b += step_b;
                }
            };
            issue(0);
            ntt_dif_reg<4, 0, false>(v);
#pragma clang loop unroll(full)
            for (int bt = 0; bt < 16 / B; bt++) {
                if (bt + 1 < 16 / B) issue(bt + 1);
#pragma clang loop unroll(full)
                for (int i = 0; i < B; i++) {
                    const int kb = bt * B + i;
                    const fe o16 = finish(mont_mul5_v(v[brev_c(kb, 4)], w[bt & 1][i]));
                    *(fe*)(out + rows_of(kb) * sizeof(fe) + lane_offset(lane_b)) = o16;
                }
            }
        }
    } else {
        const uint32_t pre = (uint32_t)(f >> a.log_s);
        const uint64_t col = f & (((uint64_t)1 << a.log_s) - 1);
        const uint64_t base = ((uint64_t)pre << (a.r + a.log_s)) + col;
        // twiddle of the next pass: w_n^(j_next * K * S_next), K = K_prev + k * 2^bits_prev
        const uint32_t j_next = (uint32_t)(col >> a.log_s_next);
        uint32_t bits_prev = 0;
        for (uint32_t u = 0; u < a.nprev; u++) bits_prev += a.rprev[u];
        const uint32_t k_prev = prefix_to_k(pre, a);
        const tw5* __restrict__ hi = a.hi;
        const tw5* __restrict__ lo = a.lo;
        // The outgoing twiddles are gathered ahead of the butterflies (and, where
        // two tables are involved, four elements ahead of their use): a gather
        // issued next to its use would sit behind the previous element's store
        // in the in-order memory counter and expose a full round trip per element.
        if constexpr (NEXT_LAST == 2) {
            {
                // the twiddle of output element i comes ready-made from the plane: a 20-byte load
                // next to the store instead of two gathers and a multiplication
                const int32_t* __restrict__ plane = a.plane;
                fe5 w[2];
                auto issue = [&](int kb) {
                    uint32_t k = ka2 + M1 * kb;
                    asm volatile("" : "+v"(k));
                    w[kb & 1] = *(const fe5*)(plane + 5 * (base + ((uint64_t)k << a.log_s)));
                };
                ntt_dif_reg<4, 0, false>(v);
                issue(0);
#pragma clang loop unroll(full)
                for (int kb = 0; kb < 16; kb++) {
                    if (kb + 1 < 16) issue(kb + 1);
                    uint32_t k = ka2 + M1 * kb;
                    asm volatile("" : "+v"(k));
                    const fe o = finish(mont_mul5_v(v[brev_c(kb, 4)], w[kb & 1]));
                    if (live) dst[base + ((uint64_t)k << a.log_s)] = o;
                }
            }
        } else if constexpr (NEXT_LAST == 3) {
            // the same from the plane of canonical 16-byte values
            const fe* __restrict__ plane16 = a.plane16;
            fe w[2];
            auto issue = [&](int kb) {
                uint32_t k = ka2 + M1 * kb;
                asm volatile("" : "+v"(k));
                w[kb & 1] = plane16[base + ((uint64_t)k << a.log_s)];
            };
            ntt_dif_reg<4, 0, false>(v);
            issue(0);
#pragma clang loop unroll(full)
            for (int kb = 0; kb < 16; kb++) {
                if (kb + 1 < 16) issue(kb + 1);
                uint32_t k = ka2 + M1 * kb;
                asm volatile("" : "+v"(k));
                const fe o = finish(mont_mul5_v(v[brev_c(kb, 4)], unpack5(w[kb & 1])));
                if (live) dst[base + ((uint64_t)k << a.log_s)] = o;
            }
        } else if constexpr (NEXT_LAST == 1) {
            // S_next = 1: exponent j_next * K, split over the hi and lo tables
            constexpr int B = 1;
            tw5 wh[2][B], wl[2][B];
            auto issue = [&](int bt) {
#pragma clang loop unroll(full)
                for (int i = 0; i < B; i++) {
                    const uint32_t k = ka2 + M1 * (bt * B + i);
                    const uint32_t E = j_next * (k_prev + (k << bits_prev));
                    wh[bt & 1][i] = hi[E >> a.h];
                    wl[bt & 1][i] = lo[E & ((1u << a.h) - 1)];
                }
            };
            // (two tables: the first batch waits until the butterflies have released their registers)
            ntt_dif_reg<4, 0, false>(v);
            issue(0);
#pragma clang loop unroll(full)
            for (int bt = 0; bt < 16 / B; bt++) {
                if (bt + 1 < 16 / B) issue(bt + 1);
#pragma clang loop unroll(full)
                for (int i = 0; i < B; i++) {
                    const int kb = bt * B + i;
                    uint32_t k = ka2 + M1 * kb;
                    asm volatile("" : "+v"(k));        // address formed here, not 16 at once up front
                    const fe5 w = mont_mul5_v(*(const fe5*)&wh[bt & 1][i], wl[bt & 1][i]);
                    const fe o = finish(mont_mul5_v(v[brev_c(kb, 4)], w));
                    if (live) dst[base + ((uint64_t)k << a.log_s)] = o;
                }
            }
        } else {
            const uint32_t sn = a.log_s_next - a.h;                // S_next / 2^h
            constexpr int B = 2;
            tw5 w[2][B];
            auto issue = [&](int bt) {
#pragma clang loop unroll(full)
                for (int i = 0; i < B; i++) {
                    const uint32_t k = ka2 + M1 * (bt * B + i);
                    const uint32_t E = (j_next * (k_prev + (k << bits_prev))) << sn;
                    w[bt & 1][i] = hi[E];
                }
            };
            issue(0);
            ntt_dif_reg<4, 0, false>(v);
#pragma clang loop unroll(full)
            for (int bt = 0; bt < 16 / B; bt++) {
                if (bt + 1 < 16 / B) issue(bt + 1);
#pragma clang loop unroll(full)
                for (int i = 0; i < B; i++) {
                    const int kb = bt * B + i;
                    uint32_t k = ka2 + M1 * kb;
                    asm volatile("" : "+v"(k));        // address formed here, not 16 at once up front
                    const fe o = finish(mont_mul5_v(v[brev_c(kb, 4)], w[bt & 1][i]));
                    if (live) dst[base + ((uint64_t)k << a.log_s)] = o;
                }
            }
        }
    }
}

// Twiddle plane of the pass before the last (see PassArgs::plane): one thread per output element,
// the same index arithmetic as round2_store's.
__global__ __launch_bounds__(256) void ntt_plane_kernel(PassArgs a, int32_t* __restrict__ plane)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >> a.log_n) return;
    const uint32_t j_next = (uint32_t)(i & (((uint64_t)1 << a.log_s) - 1));
    const uint32_t k = (uint32_t)(i >> a.log_s) & ((1u << a.r) - 1);
    const uint32_t pre = (uint32_t)(i >> (a.r + a.log_s));
    uint32_t bits_prev = 0;
    for (uint32_t u = 0; u < a.nprev; u++) bits_prev += a.rprev[u];
    const uint32_t E = j_next * (prefix_to_k(pre, a) + (k << bits_prev));
    const tw5 wh = a.hi[E >> a.h];
    const fe5 w = mont_mul5_v(*(const fe5*)&wh, a.lo[E & ((1u << a.h) - 1)]);
#pragma unroll
    for (int l = 0; l < 5; l++) plane[5 * i + l] = w.l[l];
}

__global__ __launch_bounds__(256) void ntt_plane16_kernel(PassArgs a, fe* __restrict__ plane16)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >> a.log_n) return;
    const uint32_t j_next = (uint32_t)(i & (((uint64_t)1 << a.log_s) - 1));
    const uint32_t k = (uint32_t)(i >> a.log_s) & ((1u << a.r) - 1);
    const uint32_t pre = (uint32_t)(i >> (a.r + a.log_s));
    uint32_t bits_prev = 0;
    for (uint32_t u = 0; u < a.nprev; u++) bits_prev += a.rprev[u];
    const uint32_t E = j_next * (prefix_to_k(pre, a) + (k << bits_prev));
    const tw5 wh = a.hi[E >> a.h];
    plane16[i] = finish(mont_mul5_v(*(const fe5*)&wh, a.lo[E & ((1u << a.h) - 1)]));
}

hipError_t launch_ntt_plane16(const PassArgs& a, fe* plane16, hipStream_t s)
{
    const uint64_t n = (uint64_t)1 << a.log_n;
    hipLaunchKernelGGL(ntt_plane16_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, a, plane16);
    return hipGetLastError();
}

hipError_t launch_ntt_plane(const PassArgs& a, int32_t* plane, hipStream_t s)
{
    const uint64_t n = (uint64_t)1 << a.log_n;
    hipLaunchKernelGGL(ntt_plane_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, a, plane);
    return hipGetLastError();
}

// RA = log2(R) - 4.  MODE: PASS_FIRST (strided columns, optional coset twist,
// zero fill), PASS_MIDDLE (strided columns), PASS_LAST (contiguous rows,
// transposed store to natural order).  The twiddle that the NEXT pass needs is
// applied to this pass's outputs, so every element leaves such a pass through a
// multiplication, which is what brings the lazy limbs back to a canonical
// value; the last pass has nothing to multiply by (the inverse's n^-1 rides on
// its exchange twiddles) and reduces its lazy outputs directly (finish_lazy).
// Both directions run the forward root's butterflies and tables: an inverse
// transform differs in its last pass only (INV), which stores in reversed order.
// One workgroup per tile; three workgroups are resident per CU (168 registers,
// 48-52 KiB of LDS each) and hide one another's memory and barrier waits.
template <int RA, int MODE, bool TWIST, int NEXT_LAST, bool SMALL, bool INV>
__global__ __launch_bounds__(TILE_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void ntt_pass_kernel(PassArgs a)
{
    using G = Geo<RA, MODE>;
    __shared__ fe5 lds[G::LDS_ELEMS];
    __shared__ tw5 s_tw256[RA == 0 ? 1 : 256];
    const int t = threadIdx.x;
    if constexpr (RA > 0) s_tw256[t] = a.tw256[t];

    const uint64_t tile = blockIdx.x;
    fe raw[16];
    fe5 v[16];
    load_tile<RA, MODE, SMALL, INV>(raw, a, tile, t);
    unpack_tile<RA, MODE, TWIST, SMALL>(v, raw, a, tile, t);
    // RA = 0 (radix 16, first pass only): load_tile has already put the 16 points of column t into this
    // thread's registers, which is the layout round 2 starts from: no first round, no exchange
    if constexpr (RA > 0) {
        lds_barrier();          // s_tw256 staged
        round1_exchange<RA, MODE>(v, lds, s_tw256, t);
    }
    round2_store<RA, MODE, NEXT_LAST, SMALL, INV>(v, a, tile, t);
}

template <int RA, int MODE, bool TWIST, int NEXT_LAST, bool SMALL, bool INV>
static hipError_t launch_dir(const PassArgs& a, hipStream_t s)
{
    const uint64_t total = ((((uint64_t)1 << a.log_n) * a.n_batch) + TILE_ELEMS - 1) / TILE_ELEMS;
    if (total == 0) return hipSuccess;
    if (total > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((ntt_pass_kernel<RA, MODE, TWIST, NEXT_LAST, SMALL, INV>), dim3((uint32_t)total), dim3(TILE_THREADS), 0, s, a);
    return hipGetLastError();
}

template <int RA, int MODE, bool TWIST, int NEXT_LAST, bool SMALL>
static hipError_t launch_one(const PassArgs& a, hipStream_t s)
{
    // an inverse transform is the forward one with its outputs stored in reversed order: only its last
    // pass differs; a coset twist only ever precedes a forward transform
    if (a.inverse) {
        if constexpr (TWIST) return hipErrorInvalidValue;
        else if constexpr (MODE == PASS_LAST) return launch_dir<RA, MODE, TWIST, NEXT_LAST, SMALL, true>(a, s);
    }
    return launch_dir<RA, MODE, TWIST, NEXT_LAST, SMALL, false>(a, s);
}

// The first-pass instantiations of one radix: with or without the coset twist, and by where the twiddle
// of the next pass comes from (nl: 0 = hi table, the next pass is not the last; 1 = hi * lo; 2 = plane).
template <int RA, bool SMALL>
static hipError_t launch_first(const PassArgs& a, bool twist, int nl, hipStream_t s)
{
    if (twist) {
        if (nl == 3) return launch_one<RA, PASS_FIRST, true, 3, SMALL>(a, s);
        if (nl == 2) return launch_one<RA, PASS_FIRST, true, 2, SMALL>(a, s);
        if (nl == 1) return launch_one<RA, PASS_FIRST, true, 1, SMALL>(a, s);
        if constexpr (RA == 0) return hipErrorInvalidValue;    // a radix-16 pass is always followed by the last pass
        else return launch_one<RA, PASS_FIRST, true, 0, SMALL>(a, s);
    }
    if (nl == 3) return launch_one<RA, PASS_FIRST, false, 3, SMALL>(a, s);
    if (nl == 2) return launch_one<RA, PASS_FIRST, false, 2, SMALL>(a, s);
    if (nl == 1) return launch_one<RA, PASS_FIRST, false, 1, SMALL>(a, s);
    if constexpr (RA == 0) return hipErrorInvalidValue;
    else return launch_one<RA, PASS_FIRST, false, 0, SMALL>(a, s);
}

// Shapes that no plan uses are not built: a radix-16 pass (RA = 0) is only ever the first of two, a pass of
// 128 or 256 points over a shared tile only ever the single pass of a transform of that length, and
// transforms that share tiles take at most two passes.
template <int RA, bool SMALL>
static hipError_t launch_ra(const PassArgs& a, PassMode mode, bool twist, hipStream_t s)
{
    constexpr bool FIRST_ONLY = RA == 0;
    constexpr bool LAST_ONLY = SMALL && RA >= 3;
    const int nl = a.next_is_last ? (a.plane16 ? 3 : a.plane ? 2 : 1) : 0;
    if (mode == PASS_FIRST) {
        if constexpr (LAST_ONLY) return hipErrorInvalidValue;
        else return launch_first<RA, SMALL>(a, twist, nl, s);
    }
    if (mode == PASS_MIDDLE) {
        if constexpr (SMALL || FIRST_ONLY) return hipErrorInvalidValue;
        else return nl == 3 ? launch_one<RA, PASS_MIDDLE, false, 3, false>(a, s)
                    : nl == 2 ? launch_one<RA, PASS_MIDDLE, false, 2, false>(a, s)
                    : nl == 1 ? launch_one<RA, PASS_MIDDLE, false, 1, false>(a, s)
                              : launch_one<RA, PASS_MIDDLE, false, 0, false>(a, s);
    }
    if constexpr (FIRST_ONLY) return hipErrorInvalidValue;
    else return launch_one<RA, PASS_LAST, false, 0, SMALL>(a, s);
}

hipError_t launch_ntt_pass(const PassArgs& a_in, PassMode mode, bool twist, uint32_t n_batch, hipStream_t s)
{
    if (a_in.log_n < (uint32_t)SHORT_MIN_LOG || a_in.log_n > 30 || a_in.r > a_in.log_n) return hipErrorInvalidValue;
    // the kernels address a tile as a scalar base plus 32-bit lane offsets of up to 16 rows (tile_origin)
    if (mode != PASS_LAST && a_in.log_s > 24) return hipErrorInvalidValue;
    PassArgs a = a_in;
    a.n_batch = n_batch;
    if (a.log_n < 12) {                                        // several transforms share a 4096-element tile
        switch (a.r) {
        case 4: return launch_ra<0, true>(a, mode, twist, s);
        case 5: return launch_ra<1, true>(a, mode, twist, s);
        case 6: return launch_ra<2, true>(a, mode, twist, s);
        case 7: return launch_ra<3, true>(a, mode, twist, s);
        case 8: return launch_ra<4, true>(a, mode, twist, s);
        }
        return hipErrorInvalidValue;
    }
    switch (a.r) {
    case 5: return launch_ra<1, false>(a, mode, twist, s);
    case 6: return launch_ra<2, false>(a, mode, twist, s);
    case 7: return launch_ra<3, false>(a, mode, twist, s);
    case 8: return launch_ra<4, false>(a, mode, twist, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Coset or zero-padded transforms of fewer than 512 points: the whole transform lives in one
// workgroup's LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(TILE_THREADS) void ntt_small_kernel(SmallArgs a)
{
    __shared__ fe s[1 << SMALL_MAX_LOG];
    const uint32_t n = 1u << a.log_n;
    const fe* __restrict__ src = a.src + (uint64_t)blockIdx.x * a.src_batch_stride;
    fe* __restrict__ dst = a.dst + (uint64_t)blockIdx.x * a.dst_batch_stride;
    for (uint32_t i = threadIdx.x; i < n; i += TILE_THREADS) {
        fe x = fe_zero();
        if (i < a.n_src) {
            x = src[i];
            if (a.has_twist) x = mont_mul(mont_mul(x, a.twist_lo[i & 255]), a.twist_hi[i >> 8]);
        }
        const uint32_t rev = a.log_n ? (__brev(i) >> (32 - a.log_n)) : 0;
        s[rev] = x;
    }
    __syncthreads();
    for (uint32_t lg = 1; lg <= a.log_n; lg++) {
        const uint32_t half = 1u << (lg - 1);
        const uint32_t tstep = a.log_n - lg;                   // w_len^j = w_n^(j << tstep)
        for (uint32_t b = threadIdx.x; b < (n >> 1); b += TILE_THREADS) {
            const uint32_t j = b & (half - 1);
            const uint32_t i0 = ((b >> (lg - 1)) << lg) + j;
            fe u = s[i0];
            fe w = s[i0 + half];
            if (j) w = mont_mul(w, a.tw[j << tstep]);
            s[i0] = fe_add(u, w);
            s[i0 + half] = fe_sub(u, w);
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n; i += TILE_THREADS) {
        fe x = s[i];
        if (a.has_scale) x = mont_mul(x, a.scale_m);
        dst[i] = x;
    }
}

hipError_t launch_ntt_small(const SmallArgs& a, uint32_t n_batch, hipStream_t s)
{
    if (a.log_n > (uint32_t)SMALL_MAX_LOG) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ntt_small_kernel, dim3(n_batch), dim3(TILE_THREADS), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Plain transforms of at most 16 points: one THREAD per transform, the whole of it in registers on the
// lazy-limb core (the butterfly network of the pass kernels' second round, cut to the length).  The
// inverse's n^-1 (ntt.rs:36-39) is one product per output, which also brings the limbs back to a
// canonical value; the forward outputs are reduced without a product.
// ---------------------------------------------------------------------------
template <int LOGN, bool INV>
__global__ __launch_bounds__(TILE_THREADS) void ntt_tiny_kernel(TinyArgs a)
{
    constexpr int N = 1 << LOGN;
    const uint64_t b = (uint64_t)blockIdx.x * TILE_THREADS + threadIdx.x;
    if (b >= a.n_batch) return;
    const fe* __restrict__ src = a.src + b * a.src_batch_stride;
    fe* __restrict__ dst = a.dst + b * a.dst_batch_stride;
    fe raw[N];
#pragma unroll
    for (int i = 0; i < N; i++) raw[i] = src[i];
    fe5 v[16];
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = unpack5(raw[i]);
    ntt_dif_reg<LOGN, 0, INV>(v);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const fe5 x = v[brev_c(k, LOGN)];
        if constexpr (INV) dst[k] = pack_canonical(mont_mul5_s(x, a.scale5));
        else dst[k] = finish_lazy(x);
    }
}

template <int LOGN>
static hipError_t launch_tiny_n(const TinyArgs& a, bool inverse, hipStream_t s)
{
    const uint64_t blocks = (a.n_batch + TILE_THREADS - 1) / TILE_THREADS;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (inverse) hipLaunchKernelGGL((ntt_tiny_kernel<LOGN, true>), dim3((uint32_t)blocks), dim3(TILE_THREADS), 0, s, a);
    else hipLaunchKernelGGL((ntt_tiny_kernel<LOGN, false>), dim3((uint32_t)blocks), dim3(TILE_THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ntt_tiny(const TinyArgs& a, uint32_t log_n, bool inverse, hipStream_t s)
{
    switch (log_n) {
    case 0: return launch_tiny_n<0>(a, inverse, s);
    case 1: return launch_tiny_n<1>(a, inverse, s);
    case 2: return launch_tiny_n<2>(a, inverse, s);
    case 3: return launch_tiny_n<3>(a, inverse, s);
    case 4: return launch_tiny_n<4>(a, inverse, s);
    }
    return hipErrorInvalidValue;
}

// Pointwise canonical product (ntt.rs:125-127).  r2 = 2^256 mod p.
__global__ void vec_mul_kernel(const fe* __restrict__ x, const fe* __restrict__ y, fe* __restrict__ out,
                               size_t n, fe r2)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) out[i] = mont_mul(mont_mul(x[i], y[i]), r2);
}

// Twiddle step of a four-step transform split over devices: element (i, k) of a
// row-major block is multiplied by w^((row_start + i) * k).
__global__ void fourstep_twiddle_kernel(fe* __restrict__ data, uint64_t row_start, uint64_t n_rows, uint64_t n_cols,
                                        const fe* __restrict__ hi, const fe* __restrict__ lo)
{
    const uint64_t total = n_rows * n_cols;
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
    for (; idx < total; idx += step) {
        const uint64_t i = idx / n_cols, k = idx - i * n_cols;
        const uint64_t E = (row_start + i) * k;
        fe w = hi[E >> 8];
        const uint32_t el = (uint32_t)(E & 255);
        if (el) w = mont_mul(w, lo[el]);
        data[idx] = mont_mul(data[idx], w);
    }
}

hipError_t launch_fourstep_twiddle(fe* data, uint64_t row_start, uint64_t n_rows, uint64_t n_cols, const fe* hi,
                                   const fe* lo, hipStream_t s)
{
    const uint64_t total = n_rows * n_cols;
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fourstep_twiddle_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, data, row_start, n_rows, n_cols,
                       hi, lo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The exchanges of a four-step transform split over devices.  A device holds
// n_rows rows of an (all rows) x n_cols matrix; the all-to-all that transposes
// the matrix sends the columns [d * cpw, (d + 1) * cpw) of those rows to device
// d (cpw = n_cols / world), and what device d needs from them is their
// transpose.  pack writes the send buffer [d][column within d's share][row]
// through 32 x 32 tiles in LDS (both the reads and the writes are runs of 512
// contiguous bytes) and, on the exchange that follows the first set of local
// transforms, multiplies element (row, column) by w^((row_start + row) * column)
// on the way, so the twiddle step costs no pass of its own.  unpack puts the
// received blocks [source][row][run] side by side as rows [row][source][run].
// ---------------------------------------------------------------------------
constexpr int XT = 32;
template <bool TWIDDLE>
__global__ __launch_bounds__(256) void fourstep_pack_kernel(const fe* __restrict__ src, fe* __restrict__ dst,
                                                            uint64_t n_rows, uint64_t n_cols, uint64_t cpw,
                                                            uint64_t row_start, const fe* __restrict__ hi,
                                                            const fe* __restrict__ lo)
{
    __shared__ fe tile[XT][XT + 1];
    const uint32_t tx = threadIdx.x & (XT - 1), ty = threadIdx.x / XT;      // ty < 8
    const uint64_t d = blockIdx.z;
    const uint64_t c0 = (uint64_t)blockIdx.x * XT, r0 = (uint64_t)blockIdx.y * XT;
#pragma unroll
    for (int k = 0; k < XT / 8; k++) {
        const uint64_t r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < n_rows && c < cpw) {
            const uint64_t cg = d * cpw + c;
            fe v = src[r * n_cols + cg];
            if constexpr (TWIDDLE) {
                const uint64_t E = (row_start + r) * cg;
                fe w = hi[E >> 8];
                const uint32_t el = (uint32_t)(E & 255);
                if (el) w = mont_mul(w, lo[el]);
                v = mont_mul(v, w);
            }
            tile[ty + 8 * k][tx] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XT / 8; k++) {
        const uint64_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < n_rows && c < cpw) dst[(d * cpw + c) * n_rows + r] = tile[tx][ty + 8 * k];
    }
}

hipError_t launch_fourstep_pack(const fe* src, fe* dst, uint64_t n_rows, uint64_t n_cols, uint64_t world,
                                uint64_t row_start, const fe* hi, const fe* lo, hipStream_t s)
{
    if (n_rows == 0 || n_cols == 0) return hipSuccess;
    const uint64_t cpw = n_cols / world;
    const uint64_t gx = (cpw + XT - 1) / XT, gy = (n_rows + XT - 1) / XT;
    if (gx > 0x7fffffffull || gy > 65535 || world > 65535) return hipErrorInvalidValue;
    dim3 grid((uint32_t)gx, (uint32_t)gy, (uint32_t)world);
    if (hi) hipLaunchKernelGGL(fourstep_pack_kernel<true>, grid, dim3(256), 0, s, src, dst, n_rows, n_cols, cpw, row_start, hi, lo);
    else hipLaunchKernelGGL(fourstep_pack_kernel<false>, grid, dim3(256), 0, s, src, dst, n_rows, n_cols, cpw, row_start, hi, lo);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void fourstep_unpack_kernel(const fe* __restrict__ src, fe* __restrict__ dst,
                                                              uint64_t world, uint64_t n_rows, uint64_t run)
{
    const uint64_t total = world * n_rows * run;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * 256;
    for (; i < total; i += step) {
        const uint64_t a = i % run, q = i / run, sidx = q % world, b = q / world;
        dst[i] = src[(sidx * n_rows + b) * run + a];
    }
}

hipError_t launch_fourstep_unpack(const fe* src, fe* dst, uint64_t world, uint64_t n_rows, uint64_t run, hipStream_t s)
{
    const uint64_t total = world * n_rows * run;
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(fourstep_unpack_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, src, dst, world, n_rows, run);
    return hipGetLastError();
}

hipError_t launch_vec_mul(const fe* a, const fe* b, fe* out, size_t n, fe r2, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vec_mul_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a, b, out, n, r2);
    return hipGetLastError();
}

}  // namespace starkk
