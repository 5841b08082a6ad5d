// Merkle commitment and FRI fold kernels for gfx950.
// Reference semantics: src/modules/merkle.rs:17-51 (leaf = BLAKE2b-256(bytes),
// node = BLAKE2b-256(left || right), level-order pairing) and
// src/modules/fri.rs:133-138 (split and fold).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "internal.hpp"

namespace starkk {

using namespace starkf;
using namespace starkh;

// One thread per field-element leaf (fri.rs:119: leaf bytes are
// bincode::serialize(&FieldElement), built in registers by leaf_digest).
__global__ __launch_bounds__(256) void leaf_hash_kernel(const fe* __restrict__ elems, digest* __restrict__ out,
                                                        size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = elems[i];
    out[i] = leaf_digest(x.lo, x.hi);
}

// Leaves of a batch of columns: column c is the n elements at elems + c * col_stride
// (n a power of two); digests land column after column in level0.
__global__ __launch_bounds__(256) void leaf_hash_batch_kernel(const fe* __restrict__ elems, size_t col_stride,
                                                              digest* __restrict__ out, uint32_t log_n, size_t total)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t c = i >> log_n, j = i & (((size_t)1 << log_n) - 1);
    fe x = elems[c * col_stride + j];
    out[i] = leaf_digest(x.lo, x.hi);
}

hipError_t launch_leaf_hash_batch(const fe* elems, size_t col_stride, digest* level0, uint32_t log_n, size_t n_cols,
                                  hipStream_t s)
{
    const size_t total = n_cols << log_n;
    if ((total + 255) / 256 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(leaf_hash_batch_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, elems, col_stride,
                       level0, log_n, total);
    return hipGetLastError();
}

// Leaf digests of a vector that is also copied on the way (the first layer of a FRI commit keeps
// its own copy of the codeword, fri.rs:130): the kernel is bound by the hash, the store is free.
__global__ __launch_bounds__(256) void leaf_hash_copy_kernel(const fe* __restrict__ elems, fe* __restrict__ copy,
                                                             digest* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = elems[i];
    copy[i] = x;
    out[i] = leaf_digest(x.lo, x.hi);
}

hipError_t launch_leaf_hash_copy(const fe* elems, fe* copy, digest* level0, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(leaf_hash_copy_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, elems, copy, level0, n);
    return hipGetLastError();
}

hipError_t launch_leaf_hash(const fe* elems, digest* level0, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(leaf_hash_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, elems, level0, n);
    return hipGetLastError();
}

// Leaves and the two tree levels above them in one pass (merkle.rs:17-51, level-order pairing).  A wave
// owns 256 consecutive leaves of a column, as two blocks of 128: lane l hashes leaves 2l and 2l+1 of each
// block and their parent, so every store of the wave is one contiguous run.  The 128 parents then pair up across
// lanes: lane l fetches the parents held by lanes 2l and 2l+1 (mod 64) of block l / 32 and hashes one
// grandparent, so no lane idles.  Against one launch per level this leaves out the read-back of the two
// lowest levels, 48 of the 144 bytes per leaf a tree moves through HBM, and a 2^24-leaf commitment takes
// 1.67 ms instead of 1.78 (same box, alternating).  Column c is the
// n = 2^log_n >= 256 elements at elems + c * col_stride; level0 / level1 / level2 hold the levels of all
// columns, column after column.  The hashes run through one copy of the leaf code and two of the node code
// (the loops are kept as loops: unrolled, the kernel would be 75 KB of straight-line code).
template <bool COPY>
__global__ __launch_bounds__(256) void leaf_subtree_kernel(const fe* __restrict__ elems, size_t col_stride,
                                                           fe* __restrict__ copy, digest* __restrict__ level0,
                                                           digest* __restrict__ level1, digest* __restrict__ level2,
                                                           uint32_t log_n, size_t waves)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave >= waves) return;                                 // whole waves leave together: the shuffles below stay full
    const size_t first = wave * 256, c = first >> log_n, j = first & (((size_t)1 << log_n) - 1);
    const fe* __restrict__ src = elems + c * col_stride + j;
    digest pa{}, pb{};
#pragma clang loop unroll(disable)
    for (int k = 0; k < 2; k++) {
        digest da{}, db{};
#pragma clang loop unroll(disable)
        for (int h = 0; h < 2; h++) {
            const size_t in_wave = 128 * k + 2 * lane + h;
            const fe x = src[in_wave];
            if constexpr (COPY) copy[first + in_wave] = x;
            const digest d = leaf_digest(x.lo, x.hi);
            level0[first + in_wave] = d;
            if (h == 0) da = d;
            else db = d;
        }
        const digest p = node_digest(da, db);
        level1[first / 2 + 64 * k + lane] = p;
        if (k == 0) pa = p;
        else pb = p;
    }
    // grandparent `lane` of the wave: children 2 lane and 2 lane + 1 of the 128 parents, i.e. lanes
    // (2 lane) mod 64 and the next one of block lane / 32
    const int from = (int)((2 * lane) & 63);
    digest l, r;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint64_t la = __shfl(pa.w[w], from), lb = __shfl(pb.w[w], from);
        const uint64_t ra = __shfl(pa.w[w], from + 1), rb = __shfl(pb.w[w], from + 1);
        l.w[w] = lane < 32 ? la : lb;
        r.w[w] = lane < 32 ? ra : rb;
    }
    level2[first / 4 + lane] = node_digest(l, r);
}

// Leaf level and the two levels above it of n_cols trees of n = 2^log_n leaves each, laid out as
// launch_merkle_reduce expects them (level k of all trees at tree + (2 - 2^(1-k)) * total); *level_out and
// *m_out are where the reduction carries on.  Trees of fewer than 256 leaves get their leaf level only.
hipError_t launch_leaf_levels(const fe* elems, size_t col_stride, fe* copy, digest* tree, uint32_t log_n, size_t n_cols,
                              hipStream_t s, digest** level_out, size_t* m_out)
{
    const size_t total = n_cols << log_n;
    *level_out = tree;
    *m_out = total;
    if (total == 0) return hipSuccess;
    if (log_n < 8) {
        if (copy) {
            if (n_cols != 1) return hipErrorInvalidValue;
            return launch_leaf_hash_copy(elems, copy, tree, total, s);
        }
        return launch_leaf_hash_batch(elems, col_stride, tree, log_n, n_cols, s);
    }
    const size_t waves = total / 256, blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    digest *l1 = tree + total, *l2 = l1 + total / 2;
    if (copy)
        hipLaunchKernelGGL(leaf_subtree_kernel<true>, dim3((uint32_t)blocks), dim3(256), 0, s, elems, col_stride, copy,
                           tree, l1, l2, log_n, waves);
    else
        hipLaunchKernelGGL(leaf_subtree_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, s, elems, col_stride, copy,
                           tree, l1, l2, log_n, waves);
    *level_out = l2;
    *m_out = total / 4;
    return hipGetLastError();
}

// Generic byte-string leaves (merkle.rs:38-51 takes Vec<Vec<u8>>): streaming
// multi-block BLAKE2b, one thread per leaf.
__device__ void blake2b_compress_generic(uint64_t (&h)[8], const uint64_t (&m)[16], uint64_t t, bool last)
{
    const uint64_t iv[8] = {B2_IV0, B2_IV1, B2_IV2, B2_IV3, B2_IV4, B2_IV5, B2_IV6, B2_IV7};
    uint64_t v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = iv[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
    b2_round<16, 0>(v, m);
    b2_round<16, 1>(v, m);
    b2_round<16, 2>(v, m);
    b2_round<16, 3>(v, m);
    b2_round<16, 4>(v, m);
    b2_round<16, 5>(v, m);
    b2_round<16, 6>(v, m);
    b2_round<16, 7>(v, m);
    b2_round<16, 8>(v, m);
    b2_round<16, 9>(v, m);
    b2_round<16, 10>(v, m);
    b2_round<16, 11>(v, m);
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

__global__ __launch_bounds__(64) void leaf_hash_bytes_kernel(const uint8_t* __restrict__ flat,
                                                             const uint64_t* __restrict__ offsets,
                                                             digest* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = flat + offsets[i];
    uint64_t len = offsets[i + 1] - offsets[i];
    uint64_t h[8] = {B2_IV0 ^ 0x01010020ull, B2_IV1, B2_IV2, B2_IV3, B2_IV4, B2_IV5, B2_IV6, B2_IV7};
    uint64_t t = 0;
    uint64_t m[16];
    while (len > 128) {
        for (int w = 0; w < 16; w++) {
            uint64_t x = 0;
            for (int b = 0; b < 8; b++) x |= (uint64_t)p[8 * w + b] << (8 * b);
            m[w] = x;
        }
        t += 128;
        blake2b_compress_generic(h, m, t, false);
        p += 128;
        len -= 128;
    }
    for (int w = 0; w < 16; w++) {
        uint64_t x = 0;
        for (int b = 0; b < 8; b++) {
            uint64_t k = 8 * w + b;
            if (k < len) x |= (uint64_t)p[k] << (8 * b);
        }
        m[w] = x;
    }
    t += len;
    blake2b_compress_generic(h, m, t, true);
    digest d;
    d.w[0] = h[0]; d.w[1] = h[1]; d.w[2] = h[2]; d.w[3] = h[3];
    out[i] = d;
}

hipError_t launch_leaf_hash_bytes(const uint8_t* flat, const uint64_t* offsets, digest* level0, size_t n,
                                  hipStream_t s)
{
    hipLaunchKernelGGL(leaf_hash_bytes_kernel, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, s, flat, offsets,
                       level0, n);
    return hipGetLastError();
}

// One thread per parent of one level.
__global__ __launch_bounds__(256) void node_level_kernel(const digest* __restrict__ in, digest* __restrict__ out,
                                                         size_t m_out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m_out) return;
    digest l = in[2 * i], r = in[2 * i + 1];
    out[i] = node_digest(l, r);
}

// Two levels per launch: a wave owns 256 consecutive digests of a level, as two blocks of 128; lane l hashes
// the pair (2l, 2l+1) of each block, the 128 parents pair up across lanes as in leaf_subtree_kernel, and
// every lane hashes one grandparent.  m_in is a multiple of 256; out1 / out2 are the next two levels.
__global__ __launch_bounds__(256) void node_two_levels_kernel(const digest* __restrict__ in, digest* __restrict__ out1,
                                                              digest* __restrict__ out2, size_t m_in)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave * 256 >= m_in) return;                            // whole waves leave together
    const size_t first = wave * 256;
    digest pa{}, pb{};
#pragma clang loop unroll(disable)
    for (int k = 0; k < 2; k++) {
        const size_t at = first + 128 * k + 2 * lane;
        const digest p = node_digest(in[at], in[at + 1]);
        out1[first / 2 + 64 * k + lane] = p;
        if (k == 0) pa = p;
        else pb = p;
    }
    const int from = (int)((2 * lane) & 63);
    digest l, r;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint64_t la = __shfl(pa.w[w], from), lb = __shfl(pb.w[w], from);
        const uint64_t ra = __shfl(pa.w[w], from + 1), rb = __shfl(pb.w[w], from + 1);
        l.w[w] = lane < 32 ? la : lb;
        r.w[w] = lane < 32 ? ra : rb;
    }
    out2[first / 4 + lane] = node_digest(l, r);
}

// ---- four lanes per compression --------------------------------------------
// Below about 2^17 nodes a level no longer fills the machine and its cost is
// the latency of one compression.  There a compression is spread over the four
// lanes of a quad: lane q holds column q of the 4x4 state (a = v[q],
// b = v[4+q], c = v[8+q], d = v[12+q]), so the column step is one G per lane;
// rotating b, c, d by one, two and three lanes inside the quad (DPP quad_perm)
// lines the diagonals up for the second G of the round (RFC 7693 3.2).  The
// message sits in LDS as sixteen 64-bit words (the upper eight zero for a
// node) and each lane fetches the four words the schedule gives it in a round
// through a per-lane table of byte offsets.
constexpr uint8_t SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

// w[q][r]: byte offsets (8 * word index) of the four message words lane q uses
// in round r: column x, column y, diagonal x, diagonal y, one per byte.
struct QuadSchedule {
    uint32_t w[4][12];
};
constexpr QuadSchedule make_quad_schedule()
{
    QuadSchedule t{};
    for (int q = 0; q < 4; q++)
        for (int r = 0; r < 12; r++)
            t.w[q][r] = (uint32_t)(8 * SIGMA[r][2 * q]) | ((uint32_t)(8 * SIGMA[r][2 * q + 1]) << 8) |
                        ((uint32_t)(8 * SIGMA[r][8 + 2 * q]) << 16) | ((uint32_t)(8 * SIGMA[r][9 + 2 * q]) << 24);
    return t;
}
__constant__ QuadSchedule g_quad_schedule = make_quad_schedule();

template <int CTRL> __device__ __forceinline__ u64 quad_rotate(u64 x)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)x, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> 32), CTRL, 0xf, 0xf, false);
    return ((u64)hi << 32) | lo;
}
constexpr int QUAD_ROT2 = 0x4E;     // quad_perm:[2,3,0,1]: lane i reads lane i + 2 (mod 4)

// A lane of the quad does not move d to line a diagonal up (or to put the columns back): the xor that consumes
// it first reads it from the other lane itself, through its DPP operand (the 32-bit halves are written out so
// that the compiler can fold the lane move into the xor).  b and c are first consumed by 64-bit additions, which
// have no such operand, and are moved.  ROT = 0..3: the word comes from lane (i + ROT) mod 4.
template <int ROT> __device__ __forceinline__ constexpr int quad_ctrl()
{
    return ROT == 1 ? 0x39 : ROT == 2 ? 0x4E : 0x93;      // quad_perm:[1,2,3,0], [2,3,0,1], [3,0,1,2]
}
// x + (the word `far` of lane i + ROT)
template <int ROT> __device__ __forceinline__ u64 quad_add(u64 x, u64 far)
{
    if constexpr (ROT == 0) return x + far;
    else return x + quad_rotate<quad_ctrl<ROT>()>(far);
}
// x ^ (the word `far` of lane i + ROT)
template <int ROT> __device__ __forceinline__ u64 quad_xor(u64 x, u64 far)
{
    if constexpr (ROT == 0) return x ^ far;
    else {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)far, quad_ctrl<ROT>(), 0xf, 0xf, true) ^
                            (uint32_t)x;
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(far >> 32), quad_ctrl<ROT>(), 0xf, 0xf,
                                                                  true) ^ (uint32_t)(x >> 32);
        return ((u64)hi << 32) | lo;
    }
}

// G (RFC 7693 3.1) on this lane's a with the b, c, d that lanes i + RB, i + RC, i + RD hold
template <int RB, int RC, int RD>
__device__ __forceinline__ void quad_g(u64& a, u64& b, u64& c, u64& d, u64 x, u64 y)
{
    if constexpr (RB != 0) b = quad_rotate<quad_ctrl<RB>()>(b);
    a = a + b + x;
    d = rotr64(quad_xor<RD>(a, d), 32);
    c = quad_add<RC>(d, c);
    b = rotr64(b ^ c, 24);
    a = a + b + y; d = rotr64(d ^ a, 16);
    c = c + d;     b = rotr64(b ^ c, 63);
}

struct QuadState {
    u64 a0, b0, c0, d0;      // this lane's column of the initial state, d0 without the length
    uint32_t sched[12];
};

__device__ __forceinline__ QuadState quad_state_init(uint32_t q)
{
    QuadState st;
    const u64 iv_lo = q == 0 ? B2_IV0 : q == 1 ? B2_IV1 : q == 2 ? B2_IV2 : B2_IV3;
    const u64 iv_hi = q == 0 ? B2_IV4 : q == 1 ? B2_IV5 : q == 2 ? B2_IV6 : B2_IV7;
    st.a0 = iv_lo ^ (q == 0 ? 0x01010020ull : 0ull);     // digest_length=32, fanout=1, depth=1
    st.b0 = iv_hi;
    st.c0 = iv_lo;
    st.d0 = q == 2 ? ~iv_hi : iv_hi;                      // single, final block
#pragma clang loop unroll(full)
    for (int r = 0; r < 12; r++) st.sched[r] = g_quad_schedule.w[q][r];
    return st;
}

// One BLAKE2b-256 compression of the t-byte message at LDS address msg
// (sixteen words, zero padded), by the four lanes of a quad; lane q returns
// digest word q.  After a column step lane i holds (a_i, b_i, c_i, d_i); the diagonal step of lane i works on
// (a_i, b_(i+1), c_(i+2), d_(i+3)) and leaves its results where it computed them, so the next column step finds
// b_i, c_i, d_i in lanes i + 3, i + 2, i + 1.
__device__ __forceinline__ u64 quad_blake2b(const QuadState& st, uint32_t q, const uint8_t* msg, u64 t)
{
    u64 a = st.a0, b = st.b0, c = st.c0, d = st.d0 ^ (q == 0 ? t : 0ull);
#pragma clang loop unroll(full)
    for (int r = 0; r < 12; r++) {
        const uint32_t sc = st.sched[r];
        const u64 x0 = *(const u64*)(msg + (sc & 0xffu));
        const u64 y0 = *(const u64*)(msg + ((sc >> 8) & 0xffu));
        const u64 x1 = *(const u64*)(msg + ((sc >> 16) & 0xffu));
        const u64 y1 = *(const u64*)(msg + (sc >> 24));
        if (r == 0) quad_g<0, 0, 0>(a, b, c, d, x0, y0);
        else quad_g<3, 2, 1>(a, b, c, d, x0, y0);
        quad_g<1, 2, 3>(a, b, c, d, x1, y1);
    }
    return st.a0 ^ a ^ quad_rotate<QUAD_ROT2>(c);            // c_i sits in lane i + 2 after the last diagonal step
}

// Tail of a tree: every workgroup reduces c_blk <= TAIL_CHILDREN consecutive digests of
// one level (or the whole level if it is smaller) to a single node, writing
// every level it passes to the tree.  Levels alternate between two LDS buffers
// of 128-byte message slots whose upper halves stay zero; the buffers are sized
// by the launch for the children a workgroup takes (c_blk / 2 slots and c_blk / 4),
// and a workgroup has 2 c_blk threads (four lanes per node of its first level), at least a wave.
constexpr uint32_t TAIL_CHILDREN = 512;

struct TailBuffers {
    uint8_t* a;
    uint8_t* b;
};
static inline uint32_t tail_threads(size_t c_blk) { return c_blk < 32 ? 64u : (uint32_t)(2 * c_blk); }
static inline size_t tail_lds_bytes(size_t c_blk) { return (c_blk / 2 + c_blk / 4 + 2) * 128; }

__device__ __forceinline__ TailBuffers tail_buffers(uint8_t* lds, uint32_t c_blk)
{
    return TailBuffers{lds, lds + ((c_blk >> 1) + 1) * 128};
}

__device__ __forceinline__ void tail_zero_padding(const TailBuffers& lds, uint32_t t, uint32_t c_blk)
{
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (t < 2 * c_blk) *(uint4*)(lds.a + (t >> 2) * 128 + 64 + (t & 3) * 16) = z;
    if (t < c_blk) *(uint4*)(lds.b + (t >> 2) * 128 + 64 + (t & 3) * 16) = z;
}

// lds.a holds this workgroup's c_blk children (child k in slot k/2, half k%2).
// When the launch is a single workgroup its last node is the root, which is
// also stored to root_out if that is not null (host-mapped memory).
__device__ __forceinline__ void tail_reduce(const TailBuffers& lds, digest* level, uint32_t m, uint32_t c_blk,
                                            digest* root_out)
{
    const uint32_t t = threadIdx.x, q = t & 3, h = t >> 2;
    const QuadState st = quad_state_init(q);
    __syncthreads();
    uint8_t* in = lds.a;
    uint8_t* out = lds.b;
    size_t level_off = m;                       // digests from `level` to the level being written
    size_t m_level = m >> 1;                    // its size
    for (uint32_t p = c_blk >> 1; p >= 1; p >>= 1) {
        if (h < p) {
            const u64 w = quad_blake2b(st, q, in + h * 128, 64);
            *(u64*)(out + (h >> 1) * 128 + (h & 1) * 32 + q * 8) = w;
            level[level_off + (size_t)blockIdx.x * p + h].w[q] = w;
            if (root_out && m_level == 1) root_out->w[q] = w;
        }
        __syncthreads();
        uint8_t* sw = in; in = out; out = sw;
        level_off += m_level;
        m_level >>= 1;
    }
}

__global__ __launch_bounds__(2 * TAIL_CHILDREN) void merkle_tail_kernel(digest* level, uint32_t m, uint32_t c_blk,
                                                                        digest* root_out)
{
    extern __shared__ __align__(16) uint8_t tail_lds[];
    const TailBuffers lds = tail_buffers(tail_lds, c_blk);
    const uint32_t t = threadIdx.x;                                     // c_blk = children per workgroup
    tail_zero_padding(lds, t, c_blk);
    // thread t carries bytes 16t..16t+15 of this workgroup's children into slot t/4
    if (t < 2 * c_blk) {
        const uint4* src = (const uint4*)(level + (size_t)blockIdx.x * c_blk);
        *(uint4*)(lds.a + (t >> 2) * 128 + (t & 3) * 16) = src[t];
    }
    tail_reduce(lds, level, m, c_blk, root_out);
}

// Levels wider than this many children run one thread per parent (full
// throughput); narrower ones are latency bound and go to the tail kernel.
constexpr size_t TAIL_MAX_CHILDREN = (size_t)1 << 17;

// the levels one tail launch writes: sizes m, m/2, ..., 2*(m/c_blk)
static void tail_advance(digest*& level, size_t& m, size_t c_blk)
{
    const size_t m_out = m / c_blk;
    size_t sz = m;
    while (sz > m_out) { level += sz; sz >>= 1; }
    m = m_out;
}

// Reduces a level of m digests that holds n_trees trees side by side (m / n_trees digests each,
// a power of two) until every tree is down to its root; the roots end up next to one another and
// *roots_dev, if not null, receives their address.
hipError_t launch_merkle_reduce(digest* level, size_t m, hipStream_t s, digest* root_out, size_t n_trees,
                                digest** roots_dev)
{
    size_t m_tree = m / n_trees;
    // two levels per launch while both are still wide (and a tree's share of the level is whole waves)
    while (m_tree >= 256 && (m >> 1) > TAIL_MAX_CHILDREN && m % 1024 == 0 && m / 1024 <= 0x7fffffffull) {
        hipLaunchKernelGGL(node_two_levels_kernel, dim3((uint32_t)(m / 1024)), dim3(256), 0, s, level, level + m,
                           level + m + (m >> 1), m);
        level += m + (m >> 1);
        m >>= 2;
        m_tree >>= 2;
    }
    while (m_tree > 1 && m > TAIL_MAX_CHILDREN) {
        const size_t half = m >> 1;
        hipLaunchKernelGGL(node_level_kernel, dim3((uint32_t)((half + 255) / 256)), dim3(256), 0, s, level,
                           level + m, half);
        level += m;
        m = half;
        m_tree >>= 1;
    }
    while (m_tree > 1) {
        const size_t c_blk = m_tree < TAIL_CHILDREN ? m_tree : TAIL_CHILDREN;
        hipLaunchKernelGGL(merkle_tail_kernel, dim3((uint32_t)(m / c_blk)), dim3(tail_threads(c_blk)), tail_lds_bytes(c_blk),
                           s, level, (uint32_t)m, (uint32_t)c_blk, n_trees == 1 ? root_out : nullptr);
        tail_advance(level, m, c_blk);
        m_tree /= c_blk;
    }
    if (roots_dev) *roots_dev = level;
    return hipGetLastError();
}

// FRI split-and-fold (fri.rs:133-138):
//   out[i] = 2^-1 ((1 + alpha/x_i) cw[i] + (1 - alpha/x_i) cw[half+i]),  x_i = offset * omega^i
//          = (a+b)/2 + (2^-1 alpha offset^-1) omega^-i (a-b)
// omega^-i comes from the hi/lo power tables of the top-level root w
// (omega = w^(2^shift)); optionally the leaf digest of out[i] is produced in
// the same pass so the next layer's codeword is not re-read for hashing.
// With u = 2^-1 + c omega^-i the two weights are u and 1 - u, so out[i] = b + (a - b) u: the sum and
// the halving disappear, and on the lazy five-limb core the three products (table entries, constant,
// difference) cost half of what they do on 64-bit words; the one reduction is that of the result.
__device__ __forceinline__ fe fold_one(const FoldArgs& a, size_t i)
{
    const fe x = a.cw[i], y = a.cw[a.half + i];
    const uint64_t pos = (((uint64_t)i >> a.log_chunk) << a.log_period) + a.base + ((uint64_t)i & ((1ull << a.log_chunk) - 1));
    const uint64_t E = pos << a.shift;
    auto limbs = [](const tw5& t) {
        fe5 r;
#pragma unroll
        for (int l = 0; l < 5; l++) r.l[l] = t.l[l];
        return r;
    };
    fe5 w = limbs(a.hi[E >> 8]);
    if (a.shift < 8) w = mont_mul5(w, a.lo[E & 255]);          // (from layer 8 on the low exponent bits are zero)
    const fe5 u = add5(mont_mul5(w, a.c5), limbs(a.half5));
    const fe5 yb = unpack5(y);
    return finish_lazy(add5(yb, mont_mul5(sub5(unpack5(x), yb), u)));
}

template <bool LEAF>
__global__ __launch_bounds__(256) void fri_fold_kernel(FoldArgs a)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.half) return;
    fe o = fold_one(a, i);
    a.out[i] = o;
    if constexpr (LEAF) a.leaf_out[i] = leaf_digest(o.lo, o.hi);
}

hipError_t launch_fri_fold(const FoldArgs& a, hipStream_t s)
{
    if (a.half == 0) return hipSuccess;
    dim3 grid((uint32_t)((a.half + 255) / 256));
    if (a.leaf_out) hipLaunchKernelGGL(fri_fold_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(fri_fold_kernel<false>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// Fold, leaf digests and the two tree levels above them in one pass: leaf_subtree_kernel with the leaves
// produced by the fold instead of loaded (a wave owns 256 consecutive outputs; a.half is a multiple of 256).
__global__ __launch_bounds__(256) void fri_fold_subtree_kernel(FoldArgs a)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave * 256 >= a.half) return;                          // whole waves leave together
    const size_t first = wave * 256;
    digest* __restrict__ level1 = a.leaf_out + a.half;
    digest* __restrict__ level2 = level1 + a.half / 2;
    digest pa{}, pb{};
#pragma clang loop unroll(disable)
    for (int k = 0; k < 2; k++) {
        digest da{}, db{};
#pragma clang loop unroll(disable)
        for (int h = 0; h < 2; h++) {
            const size_t i = first + 128 * k + 2 * lane + h;
            const fe o = fold_one(a, i);
            a.out[i] = o;
            const digest d = leaf_digest(o.lo, o.hi);
            a.leaf_out[i] = d;
            if (h == 0) da = d;
            else db = d;
        }
        const digest p = node_digest(da, db);
        level1[first / 2 + 64 * k + lane] = p;
        if (k == 0) pa = p;
        else pb = p;
    }
    const int from = (int)((2 * lane) & 63);
    digest l, r;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint64_t la = __shfl(pa.w[w], from), lb = __shfl(pb.w[w], from);
        const uint64_t ra = __shfl(pa.w[w], from + 1), rb = __shfl(pb.w[w], from + 1);
        l.w[w] = lane < 32 ? la : lb;
        r.w[w] = lane < 32 ? ra : rb;
    }
    level2[first / 4 + lane] = node_digest(l, r);
}

// Small layers: fold, leaf digests and the first tree levels in one launch
// (a.leaf_out is level 0 of the new layer's tree, a.half its leaf count, c_blk the leaves a workgroup takes).
__global__ __launch_bounds__(2 * TAIL_CHILDREN) void fri_fold_tail_kernel(FoldArgs a, uint32_t c_blk, digest* root_out)
{
    extern __shared__ __align__(16) uint8_t tail_lds[];
    const TailBuffers lds = tail_buffers(tail_lds, c_blk);
    const uint32_t t = threadIdx.x;
    const uint32_t m = (uint32_t)a.half;
    tail_zero_padding(lds, t, c_blk);
    if (t < c_blk) {
        const size_t i = (size_t)blockIdx.x * c_blk + t;
        fe o = fold_one(a, i);
        a.out[i] = o;
        const digest d = leaf_digest(o.lo, o.hi);
        a.leaf_out[i] = d;
        *(digest*)(lds.a + (t >> 1) * 128 + (t & 1) * 32) = d;
        if (root_out && m == 1) *root_out = d;
    }
    tail_reduce(lds, a.leaf_out, m, c_blk, root_out);
}

// Leaves per workgroup of the fold-and-tree launch of a small layer.  A workgroup's first levels are bound by
// the one CU it runs on (four lanes per node: 256 nodes are sixteen waves, 4.9 us; 128 nodes 2.4 us; from 64
// nodes down a level costs the 1.4 us of one compression), so a layer is cut into more, smaller workgroups
// than the tree tail's 512 children while it has more than one of them, and the levels that moves to the
// following tail launch run there at the floor.
static size_t fold_tail_leaves(size_t half)
{
    static const size_t leaves = [] {
        const char* e = getenv("STARK_FRI_TAIL_LEAVES");
        size_t v = e ? (size_t)atol(e) : 256;
        if (v < 8 || v > TAIL_CHILDREN || (v & (v - 1))) v = 256;
        return v;
    }();
    return half <= 2 * leaves ? (half < TAIL_CHILDREN ? half : (size_t)TAIL_CHILDREN) : leaves;
}

// Fold layer r into layer r+1 and build layer r+1's whole tree.
hipError_t launch_fri_fold_commit(const FoldArgs& a, hipStream_t s, digest* root_out)
{
    if (a.half == 0 || !a.leaf_out) return hipErrorInvalidValue;
    if (a.half > TAIL_MAX_CHILDREN) {
        if (a.half % 1024 || a.half / 1024 > 0x7fffffffull) return hipErrorInvalidValue;     // (a power of two above 2^17)
        hipLaunchKernelGGL(fri_fold_subtree_kernel, dim3((uint32_t)(a.half / 1024)), dim3(256), 0, s, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_merkle_reduce(a.leaf_out + a.half + a.half / 2, a.half / 4, s, root_out);
    }
    const size_t c_blk = fold_tail_leaves(a.half);
    hipLaunchKernelGGL(fri_fold_tail_kernel, dim3((uint32_t)(a.half / c_blk)), dim3(tail_threads(c_blk)), tail_lds_bytes(c_blk), s,
                       a, (uint32_t)c_blk, root_out);
    digest* level = a.leaf_out;
    size_t m = a.half;
    tail_advance(level, m, c_blk);
    return launch_merkle_reduce(level, m, s, root_out);
}

__global__ void merkle_open_kernel(const digest* __restrict__ tree, size_t n, size_t index,
                                   digest* __restrict__ path)
{
    // thread k fetches the sibling on level k
    uint32_t k = threadIdx.x;
    size_t m = n;
    size_t off = 0;
    for (uint32_t l = 0; l < k; l++) { off += m; m >>= 1; }
    if (m > 1) path[k] = tree[off + ((index >> k) ^ 1)];
}

hipError_t launch_merkle_open(const digest* tree, size_t n, size_t index, digest* path_out, hipStream_t s,
                              uint32_t max_levels)
{
    uint32_t depth = 0;
    while (((size_t)1 << depth) < n) depth++;
    if (depth > max_levels) depth = max_levels;
    if (depth == 0) return hipSuccess;
    hipLaunchKernelGGL(merkle_open_kernel, dim3(1), dim3(depth), 0, s, tree, n, index, path_out);
    return hipGetLastError();
}

// Many openings of one tree at once: block b serves idx[b], thread k fetches the sibling on level k.
__global__ void merkle_open_many_kernel(const digest* __restrict__ tree, size_t n, const uint64_t* __restrict__ idx,
                                        uint32_t levels, digest* __restrict__ paths)
{
    const uint32_t k = threadIdx.x;
    if (k >= levels) return;
    const size_t index = idx[blockIdx.x];
    size_t m = n, off = 0;
    for (uint32_t l = 0; l < k; l++) { off += m; m >>= 1; }
    paths[(size_t)blockIdx.x * levels + k] = tree[off + ((index >> k) ^ 1)];
}

hipError_t launch_merkle_open_many(const digest* tree, size_t n, const uint64_t* idx_dev, size_t count, uint32_t levels,
                                   digest* paths_out, hipStream_t s)
{
    if (count == 0 || levels == 0) return hipSuccess;
    hipLaunchKernelGGL(merkle_open_many_kernel, dim3((uint32_t)count), dim3(64), 0, s, tree, n, idx_dev, levels, paths_out);
    return hipGetLastError();
}

// The whole query phase of Fri::prove (fri.rs:155-209,228-242) in one launch: block (s, r) answers opening s
// of round r -- s < count: position a of layer r, s < 2 count: position b, else position c of layer r + 1 --
// thread 0 fetches the element, thread k the sibling on level k of that layer's tree.
__global__ __launch_bounds__(64) void fri_query_all_kernel(const QueryRound* __restrict__ rounds,
                                                           const uint64_t* __restrict__ idx, uint32_t count,
                                                           fe* __restrict__ values, digest* __restrict__ paths)
{
    const uint32_t s = blockIdx.x, r = blockIdx.y, k = threadIdx.x;
    const QueryRound q = rounds[r];
    const bool next = s >= 2 * count;
    const size_t index = idx[(size_t)r * 3 * count + s];
    const size_t n = next ? q.n_next : q.n;
    const uint32_t levels = next ? q.depth_next : q.depth;
    if (k == 0) values[(size_t)r * 3 * count + s] = (next ? q.cw_next : q.cw)[index];
    if (k >= levels) return;
    const digest* __restrict__ tree = next ? q.tree_next : q.tree;
    const size_t at = next ? q.path_at + 2 * (size_t)count * q.depth + (size_t)(s - 2 * count) * q.depth_next
                           : q.path_at + (size_t)s * q.depth;
    paths[at + k] = tree[2 * n - ((2 * n) >> k) + ((index >> k) ^ 1)];       // level k starts at 2n - (2n >> k)
}

hipError_t launch_fri_query_all(const QueryRound* rounds_dev, size_t n_rounds, const uint64_t* idx_dev, size_t count,
                                fe* values_out, digest* paths_out, hipStream_t s)
{
    if (n_rounds == 0 || count == 0) return hipSuccess;
    if (3 * count > 0x7fffffffull || n_rounds > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fri_query_all_kernel, dim3((uint32_t)(3 * count), (uint32_t)n_rounds), dim3(64), 0, s, rounds_dev,
                       idx_dev, (uint32_t)count, values_out, paths_out);
    return hipGetLastError();
}

__global__ void gather_fe_kernel(const fe* __restrict__ src, const uint64_t* __restrict__ idx, fe* __restrict__ out,
                                 size_t count)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = src[idx[i]];
}

hipError_t launch_gather_fe(const fe* src, const uint64_t* idx, fe* out, size_t count, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_fe_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, src, idx, out, count);
    return hipGetLastError();
}

}  // namespace starkk
