// Merkle commitment and FRI fold kernels for gfx950.
// Reference semantics: src/modules/merkle.rs:17-51 (leaf = BLAKE2b-256(bytes),
// node = BLAKE2b-256(left || right), level-order pairing) and
// src/modules/fri.rs:133-138 (split and fold).
#include <hip/hip_runtime.h>

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

hipError_t launch_leaf_hash(const fe* elems, digest* level0, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(leaf_hash_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, elems, level0, n);
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

// Single workgroup: reduces m <= 2048 digests to the root, passing levels
// through LDS and writing each level to the tree.
constexpr int TOP_MAX = 2048;
__global__ __launch_bounds__(1024) void merkle_top_kernel(digest* level, uint32_t m)
{
    __shared__ digest s[TOP_MAX];
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) s[i] = level[i];
    __syncthreads();
    digest* out = level + m;
    while (m > 1) {
        const uint32_t half = m >> 1;
        digest d;
        const bool act = threadIdx.x < half;
        if (act) {
            d = node_digest(s[2 * threadIdx.x], s[2 * threadIdx.x + 1]);
            out[threadIdx.x] = d;
        }
        __syncthreads();
        if (act) s[threadIdx.x] = d;
        __syncthreads();
        out += half;
        m = half;
    }
}

hipError_t launch_merkle_reduce(digest* level, size_t m, hipStream_t s)
{
    while (m > (size_t)TOP_MAX) {
        const size_t half = m >> 1;
        hipLaunchKernelGGL(node_level_kernel, dim3((uint32_t)((half + 255) / 256)), dim3(256), 0, s, level,
                           level + m, half);
        level += m;
        m = half;
    }
    if (m > 1) hipLaunchKernelGGL(merkle_top_kernel, dim3(1), dim3(1024), 0, s, level, (uint32_t)m);
    return hipGetLastError();
}

// FRI split-and-fold (fri.rs:133-138):
//   out[i] = 2^-1 ((1 + alpha/x_i) cw[i] + (1 - alpha/x_i) cw[half+i]),  x_i = offset * omega^i
//          = (a+b)/2 + (2^-1 alpha offset^-1) omega^-i (a-b)
// omega^-i comes from the hi/lo power tables of the top-level root w
// (omega = w^(2^shift)); optionally the leaf digest of out[i] is produced in
// the same pass so the next layer's codeword is not re-read for hashing.
template <bool LEAF>
__global__ __launch_bounds__(256) void fri_fold_kernel(FoldArgs a)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.half) return;
    fe x = a.cw[i], y = a.cw[a.half + i];
    fe sum = fe_half(fe_add(x, y));
    fe dif = fe_sub(x, y);
    const uint64_t E = (uint64_t)i << a.shift;
    fe w = a.hi[E >> 8];
    const uint32_t el = (uint32_t)(E & 255);
    if (el) w = mont_mul(w, a.lo[el]);
    w = mont_mul(w, a.c_m);
    fe o = fe_add(sum, mont_mul(dif, w));
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

hipError_t launch_merkle_open(const digest* tree, size_t n, size_t index, digest* path_out, hipStream_t s)
{
    uint32_t depth = 0;
    while (((size_t)1 << depth) < n) depth++;
    if (depth == 0) return hipSuccess;
    hipLaunchKernelGGL(merkle_open_kernel, dim3(1), dim3(depth), 0, s, tree, n, index, path_out);
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
