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

#include <cstdlib>

#include "field5.hpp"
#include "internal.hpp"

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
// only the twiddle products reduce.  tw16[i] = w_16^i (Montgomery form, 2^156);
// the index is a compile-time constant after unrolling, so the loads are scalar.
template <int LOGM, int BASE>
__device__ __forceinline__ void ntt_dif_reg(fe5 (&v)[16], const tw5* __restrict__ tw16)
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
                v[BASE + s + j + half] = (e == 0) ? d : mont_mul5_s(d, tw16[e]);
            }
        }
    }
}

template <int LOGM, int G>
struct RoundOne {
    __device__ __forceinline__ static void run(fe5 (&v)[16], const tw5* __restrict__ tw16)
    {
        ntt_dif_reg<LOGM, (G - 1) * (1 << LOGM)>(v, tw16);
        RoundOne<LOGM, G - 1>::run(v, tw16);
    }
};
template <int LOGM>
struct RoundOne<LOGM, 0> {
    __device__ __forceinline__ static void run(fe5 (&)[16], const tw5* __restrict__) {}
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
__device__ __forceinline__ fe finish(const fe5& x)
{
    fe o = pack_fast(x);
    if (__builtin_expect(noncanonical5(x) != 0, 0)) o = reduce5_slow(x);
    return o;
}

// Issue the 16 global loads of one thread for one tile (round-1 layout).
// tile enumerates (batch, tile-in-transform); rows past n_src of a first pass
// are the zero padding and are not read.
template <int RA, int MODE>
__device__ __forceinline__ void load_tile(fe (&raw)[16], const PassArgs& a, uint64_t tile, int t)
{
    constexpr int M1 = 1 << RA;
    constexpr int G1 = 16 / M1;
    constexpr int LOGC = 8 - RA;
    constexpr int C = 1 << LOGC;
    const uint32_t ltiles = a.log_n - 12;
    const uint64_t batch = tile >> ltiles;
    const uint32_t tx = (uint32_t)(tile & (((uint64_t)1 << ltiles) - 1));
    const fe* __restrict__ src = a.src + batch * a.src_batch_stride;
#pragma unroll
    for (int g = 0; g < G1; g++) {
        const int q1 = t + TILE_THREADS * g;
        if constexpr (MODE == PASS_LAST) {
            const int c = q1 >> 4, jb = q1 & 15;
            const uint32_t b = tx * C + c;                         // low part of the output index
            const uint64_t row = (uint64_t)k_to_prefix(b, a) << a.r;
#pragma unroll
            for (int ja = 0; ja < M1; ja++) raw[g * M1 + ja] = src[row + ja * 16 + jb];
        } else {
            const int jb = q1 >> LOGC, c = q1 & (C - 1);
            const uint64_t f = (uint64_t)tx * C + c;               // flat column of this pass
            const uint64_t pre = f >> a.log_s;
            const uint64_t col = f & (((uint64_t)1 << a.log_s) - 1);
            const uint64_t base = (pre << (a.r + a.log_s)) + col;
#pragma unroll
            for (int ja = 0; ja < M1; ja++) {
                const uint64_t idx = base + ((uint64_t)(ja * 16 + jb) << a.log_s);
                if constexpr (MODE == PASS_FIRST) {
                    raw[g * M1 + ja] = fe_zero();
                    if (idx < a.n_src) raw[g * M1 + ja] = src[idx];
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
    static constexpr int LDS_ELEMS = R * PITCH; // >= 4096: also the staging area of the prefetch
};

// canonical loads -> lazy limbs, plus the coset twist of a first pass
template <int RA, int MODE, bool TWIST>
__device__ __forceinline__ void unpack_tile(fe5 (&v)[16], const fe (&raw)[16], const PassArgs& a, uint32_t tx, int t)
{
    using G = Geo<RA, MODE>;
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = unpack5(raw[i]);
    if constexpr (MODE == PASS_FIRST && TWIST) {
#pragma unroll
        for (int g = 0; g < G::G1; g++) {
            const int q1 = t + TILE_THREADS * g;
            const int jb = q1 >> G::LOGC, c = q1 & (G::C - 1);
            const uint64_t f = (uint64_t)tx * G::C + c;
            const uint64_t pre = f >> a.log_s;
            const uint64_t col = f & (((uint64_t)1 << a.log_s) - 1);
            const uint64_t base = (pre << (a.r + a.log_s)) + col;
#pragma unroll
            for (int ja = 0; ja < G::M1; ja++) {
                const uint64_t idx = base + ((uint64_t)(ja * 16 + jb) << a.log_s);
                if (idx < a.n_src)
                    v[g * G::M1 + ja] = mont_mul5_v(
                        v[g * G::M1 + ja], mont_mul5_v(*(const fe5*)&a.twist_hi[idx >> 8], a.twist_lo[idx & 255]));
            }
        }
    }
}

// round 1 (radix M1 over ja), inner twiddle w_R^(jb*ka), exchange through LDS;
// on return v holds the 16 round-2 inputs of this thread.  Ends with every
// thread's LDS reads issued and waited for, but NOT with a barrier.
template <int RA, int MODE>
__device__ __forceinline__ void round1_exchange(fe5 (&v)[16], fe* lds, const tw5* s_tw256, const tw5* __restrict__ tw16,
                                                int t)
{
    using G = Geo<RA, MODE>;
    RoundOne<RA, G::G1>::run(v, tw16);
#pragma unroll
    for (int g = 0; g < G::G1; g++) {
        const int q1 = t + TILE_THREADS * g;
        int jb, c;
        if constexpr (MODE == PASS_LAST) { c = q1 >> 4; jb = q1 & 15; }
        else { jb = q1 >> G::LOGC; c = q1 & (G::C - 1); }
#pragma unroll
        for (int ka = 0; ka < G::M1; ka++) {
            const int reg = g * G::M1 + brev_c(ka, RA);
            const fe5 x = mont_mul5_v(v[reg], s_tw256[(jb * ka) << G::TWSHIFT]);
            lds[(ka * 16 + jb) * G::PITCH + c] = finish(x);
        }
    }
    lds_barrier();
    const int ka2 = t >> G::LOGC, c2 = t & (G::C - 1);
#pragma unroll
    for (int jb = 0; jb < 16; jb++) v[jb] = unpack5(lds[(ka2 * 16 + jb) * G::PITCH + c2]);
}

// round 2 (radix 16 over jb), outgoing twiddle, store
template <int RA, int MODE, bool NEXT_LAST>
__device__ __forceinline__ void round2_store(fe5 (&v)[16], const PassArgs& a, fe* __restrict__ dst, uint32_t tx,
                                             const tw5* __restrict__ tw16, int t)
{
    using G = Geo<RA, MODE>;
    constexpr int M1 = G::M1;
    const int ka2 = t >> G::LOGC, c2 = t & (G::C - 1);
    ntt_dif_reg<4, 0>(v, tw16);
    if constexpr (MODE == PASS_LAST) {
        const uint64_t b = (uint64_t)tx * G::C + c2;
        const uint32_t sh = a.log_n - a.r;
        const tw5 fin = *a.fin;
#pragma unroll
        for (int kb = 0; kb < 16; kb++) {
            const uint64_t k = ka2 + M1 * kb;
            dst[(k << sh) + b] = finish(mont_mul5_s(v[brev_c(kb, 4)], fin));
        }
    } else {
        const uint64_t f = (uint64_t)tx * G::C + c2;
        const uint32_t pre = (uint32_t)(f >> a.log_s);
        const uint64_t col = f & (((uint64_t)1 << a.log_s) - 1);
        const uint64_t base = ((uint64_t)pre << (a.r + a.log_s)) + col;
        // twiddle of the next pass: w_n^(j_next * K * S_next), K = K_prev + k * 2^bits_prev
        const uint32_t j_next = (uint32_t)(col >> a.log_s_next);
        uint32_t bits_prev = 0;
        for (uint32_t u = 0; u < a.nprev; u++) bits_prev += a.rprev[u];
        const uint32_t k_prev = prefix_to_k(pre, a);
        if constexpr (NEXT_LAST) {
            // S_next = 1: exponent j_next * K, split over the hi and lo tables
#pragma clang loop unroll(full)
            for (int kb = 0; kb < 16; kb++) {
                const uint32_t k = ka2 + M1 * kb;
                const uint32_t E = j_next * (k_prev + (k << bits_prev));
                const fe5 w = mont_mul5_v(*(const fe5*)&a.hi[E >> a.h], a.lo[E & ((1u << a.h) - 1)]);
                dst[base + ((uint64_t)k << a.log_s)] = finish(mont_mul5_v(v[brev_c(kb, 4)], w));
            }
        } else {
            const uint32_t sn = a.log_s_next - a.h;                // S_next / 2^h
#pragma clang loop unroll(full)
            for (int kb = 0; kb < 16; kb++) {
                const uint32_t k = ka2 + M1 * kb;
                const uint32_t E = (j_next * (k_prev + (k << bits_prev))) << sn;
                dst[base + ((uint64_t)k << a.log_s)] = finish(mont_mul5_v(v[brev_c(kb, 4)], a.hi[E]));
            }
        }
    }
}

// RA = log2(R) - 4.  MODE: PASS_FIRST (strided columns, optional coset twist,
// zero fill), PASS_MIDDLE (strided columns), PASS_LAST (contiguous rows,
// transposed store to natural order).  The twiddle that the NEXT pass needs is
// applied to this pass's outputs, so every element leaves the pass through a
// multiplication, which is what brings the lazy limbs back to a canonical
// value; the last pass multiplies by `fin` (1, or n^-1 for the inverse).
// One workgroup per tile; the second resident workgroup of the CU hides this
// one's global-memory latency.
template <int RA, int MODE, bool TWIST, bool NEXT_LAST>
__global__ __launch_bounds__(TILE_THREADS) void ntt_pass_kernel(PassArgs a, uint64_t total_tiles)
{
    using G = Geo<RA, MODE>;
    __shared__ fe lds[G::LDS_ELEMS];
    __shared__ tw5 s_tw256[256];
    const int t = threadIdx.x;
    const tw5* __restrict__ tw16 = a.tw16;
    const uint32_t ltiles = a.log_n - 12;
    s_tw256[t] = a.tw256[t];
    (void)total_tiles;

    const uint64_t tile = blockIdx.x;
    const uint64_t batch = tile >> ltiles;
    const uint32_t tx = (uint32_t)(tile & (((uint64_t)1 << ltiles) - 1));
    fe raw[16];
    fe5 v[16];
    load_tile<RA, MODE>(raw, a, tile, t);
    unpack_tile<RA, MODE, TWIST>(v, raw, a, tx, t);
    lds_barrier();          // s_tw256 staged
    round1_exchange<RA, MODE>(v, lds, s_tw256, tw16, t);
    round2_store<RA, MODE, NEXT_LAST>(v, a, a.dst + batch * a.dst_batch_stride, tx, tw16, t);
}

// Asynchronous copy of one tile's 16 loads per thread straight into LDS
// (global_load_lds: no destination registers).  Slot i of thread t lands at
// stage[i*256 + t]; a wave's 64 lanes write 1 KiB contiguously, as the
// instruction requires.  Zero padding of a first pass reads a zero block.
template <int RA, int MODE>
__device__ __forceinline__ void prefetch_tile(fe* stage, const PassArgs& a, uint64_t tile, int t)
{
    using G = Geo<RA, MODE>;
    const uint32_t ltiles = a.log_n - 12;
    const uint64_t batch = tile >> ltiles;
    const uint32_t tx = (uint32_t)(tile & (((uint64_t)1 << ltiles) - 1));
    const fe* __restrict__ src = a.src + batch * a.src_batch_stride;
    const int wave_base = t & ~63;
#pragma unroll
    for (int g = 0; g < G::G1; g++) {
        const int q1 = t + TILE_THREADS * g;
        if constexpr (MODE == PASS_LAST) {
            const int c = q1 >> 4, jb = q1 & 15;
            const uint32_t b = tx * G::C + c;
            const uint64_t row = (uint64_t)k_to_prefix(b, a) << a.r;
#pragma unroll
            for (int ja = 0; ja < G::M1; ja++) {
                const fe* gp = src + row + ja * 16 + jb;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)&stage[(g * G::M1 + ja) * 256 + wave_base],
                                                 16, 0, 0);
            }
        } else {
            const int jb = q1 >> G::LOGC, c = q1 & (G::C - 1);
            const uint64_t f = (uint64_t)tx * G::C + c;
            const uint64_t pre = f >> a.log_s;
            const uint64_t col = f & (((uint64_t)1 << a.log_s) - 1);
            const uint64_t base = (pre << (a.r + a.log_s)) + col;
#pragma unroll
            for (int ja = 0; ja < G::M1; ja++) {
                const uint64_t idx = base + ((uint64_t)(ja * 16 + jb) << a.log_s);
                const fe* gp = src + idx;
                if constexpr (MODE == PASS_FIRST) {
                    if (idx >= a.n_src) gp = a.zero_block;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)&stage[(g * G::M1 + ja) * 256 + wave_base],
                                                 16, 0, 0);
            }
        }
    }
}

// Persistent form: a workgroup walks tiles blockIdx.x, +gridDim.x, ... and
// copies its NEXT tile into LDS asynchronously while it computes round 2 of
// the current one.  The copy reuses the exchange buffer, which is idle from
// the end of the round-2 reads until the next tile's round-1 writes, so
// neither LDS nor registers grow and two workgroups stay resident per CU.
template <int RA, int MODE, bool TWIST, bool NEXT_LAST>
__global__ __launch_bounds__(TILE_THREADS) void ntt_pass_kernel_persist(PassArgs a, uint64_t total_tiles)
{
    using G = Geo<RA, MODE>;
    __shared__ fe lds[G::LDS_ELEMS];
    __shared__ tw5 s_tw256[256];
    const int t = threadIdx.x;
    const tw5* __restrict__ tw16 = a.tw16;
    const uint32_t ltiles = a.log_n - 12;
    s_tw256[t] = a.tw256[t];

    uint64_t tile = blockIdx.x;
    if (tile < total_tiles) prefetch_tile<RA, MODE>(lds, a, tile, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; tile < total_tiles; tile += gridDim.x) {
        const uint64_t batch = tile >> ltiles;
        const uint32_t tx = (uint32_t)(tile & (((uint64_t)1 << ltiles) - 1));
        fe raw[16];
        fe5 v[16];
        // this thread's own 16 slots (written by its own wave's copy instructions)
#pragma unroll
        for (int i = 0; i < 16; i++) raw[i] = lds[i * 256 + t];
        unpack_tile<RA, MODE, TWIST>(v, raw, a, tx, t);
        lds_barrier();      // staging area fully read (and s_tw256 staged) before the exchange overwrites it
        round1_exchange<RA, MODE>(v, lds, s_tw256, tw16, t);
        lds_barrier();      // exchange buffer fully read before the next copy lands in it
        const bool more = tile + gridDim.x < total_tiles;
        if (more) prefetch_tile<RA, MODE>(lds, a, tile + gridDim.x, t);
        round2_store<RA, MODE, NEXT_LAST>(v, a, a.dst + batch * a.dst_batch_stride, tx, tw16, t);
        // The copy is older than this tile's 16 stores and every table load has been consumed,
        // so "all but the 16 youngest" means the copy has landed.
        if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
}

static uint32_t persistent_grid()
{
    static uint32_t cached = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return (uint32_t)(cus > 0 ? cus : 256) * 2;      // two resident workgroups per CU
    }();
    return cached;
}

static bool use_persistent()
{
    static const bool on = [] {
        const char* e = getenv("STARK_NTT_PERSIST");
        return e && e[0] == '1';
    }();
    return on;
}

template <int RA, int MODE, bool TWIST, bool NEXT_LAST>
static hipError_t launch_one(const PassArgs& a, uint32_t n_batch, hipStream_t s)
{
    const uint64_t total = (((uint64_t)1 << a.log_n) / TILE_ELEMS) * n_batch;
    if (total > 0x7fffffffull) return hipErrorInvalidValue;
    if (use_persistent() && total > persistent_grid()) {
        hipLaunchKernelGGL((ntt_pass_kernel_persist<RA, MODE, TWIST, NEXT_LAST>), dim3(persistent_grid()),
                           dim3(TILE_THREADS), 0, s, a, total);
    } else {
        hipLaunchKernelGGL((ntt_pass_kernel<RA, MODE, TWIST, NEXT_LAST>), dim3((uint32_t)total), dim3(TILE_THREADS), 0,
                           s, a, total);
    }
    return hipGetLastError();
}

template <int RA>
static hipError_t launch_ra(const PassArgs& a, PassMode mode, bool twist, uint32_t n_batch, hipStream_t s)
{
    const bool nl = a.next_is_last != 0;
    switch (mode) {
    case PASS_FIRST:
        if (twist) return nl ? launch_one<RA, PASS_FIRST, true, true>(a, n_batch, s)
                             : launch_one<RA, PASS_FIRST, true, false>(a, n_batch, s);
        return nl ? launch_one<RA, PASS_FIRST, false, true>(a, n_batch, s)
                  : launch_one<RA, PASS_FIRST, false, false>(a, n_batch, s);
    case PASS_MIDDLE:
        return nl ? launch_one<RA, PASS_MIDDLE, false, true>(a, n_batch, s)
                  : launch_one<RA, PASS_MIDDLE, false, false>(a, n_batch, s);
    case PASS_LAST: return launch_one<RA, PASS_LAST, false, false>(a, n_batch, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_ntt_pass(const PassArgs& a, PassMode mode, bool twist, uint32_t n_batch, hipStream_t s)
{
    if (a.log_n <= (uint32_t)SMALL_MAX_LOG || a.log_n > 30) return hipErrorInvalidValue;
    switch (a.r) {
    case 5: return launch_ra<1>(a, mode, twist, n_batch, s);
    case 6: return launch_ra<2>(a, mode, twist, n_batch, s);
    case 7: return launch_ra<3>(a, mode, twist, n_batch, s);
    case 8: return launch_ra<4>(a, mode, twist, n_batch, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// n <= 4096: the whole transform lives in one workgroup's LDS.
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

hipError_t launch_vec_mul(const fe* a, const fe* b, fe* out, size_t n, fe r2, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vec_mul_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a, b, out, n, r2);
    return hipGetLastError();
}

}  // namespace starkk
