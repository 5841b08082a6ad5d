// Compute-only ceiling of the BLAKE2b leaf and node compressions on gfx950:
// chained hashes in registers, no memory traffic, 8 waves per SIMD.
// Standalone: hipcc --offload-arch=gfx950 -O3 hash_bench.hip -o hash_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../stark_amd/csrc/blake2b.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

using namespace starkh;
constexpr int CHAIN = 32;

// Variant: rotations by 24/16/63 as two v_alignbit_b32 on the 32-bit halves.
__device__ __forceinline__ u64 rot_ab(u64 x, int r)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (r == 32) return ((u64)lo << 32) | hi;
    uint32_t nlo, nhi;
    if (r < 32) { nlo = __builtin_amdgcn_alignbit(hi, lo, r); nhi = __builtin_amdgcn_alignbit(lo, hi, r); }
    else { nlo = __builtin_amdgcn_alignbit(lo, hi, r - 32); nhi = __builtin_amdgcn_alignbit(hi, lo, r - 32); }
    return ((u64)nhi << 32) | nlo;
}
#define G2(a, b, c, d, x, y)                                 \
    v[a] = v[a] + v[b] + (x); v[d] = rot_ab(v[d] ^ v[a], 32);  \
    v[c] = v[c] + v[d];       v[b] = rot_ab(v[b] ^ v[c], 24);  \
    v[a] = v[a] + v[b] + (y); v[d] = rot_ab(v[d] ^ v[a], 16);  \
    v[c] = v[c] + v[d];       v[b] = rot_ab(v[b] ^ v[c], 63);
template <int NW, int R> __device__ __forceinline__ void round2(u64 (&v)[16], const u64 (&m)[NW])
{
    using S = sigma<R>;
    G2(0, 4, 8, 12, (mword<NW, S::s[0]>(m)), (mword<NW, S::s[1]>(m)));
    G2(1, 5, 9, 13, (mword<NW, S::s[2]>(m)), (mword<NW, S::s[3]>(m)));
    G2(2, 6, 10, 14, (mword<NW, S::s[4]>(m)), (mword<NW, S::s[5]>(m)));
    G2(3, 7, 11, 15, (mword<NW, S::s[6]>(m)), (mword<NW, S::s[7]>(m)));
    G2(0, 5, 10, 15, (mword<NW, S::s[8]>(m)), (mword<NW, S::s[9]>(m)));
    G2(1, 6, 11, 12, (mword<NW, S::s[10]>(m)), (mword<NW, S::s[11]>(m)));
    G2(2, 7, 8, 13, (mword<NW, S::s[12]>(m)), (mword<NW, S::s[13]>(m)));
    G2(3, 4, 9, 14, (mword<NW, S::s[14]>(m)), (mword<NW, S::s[15]>(m)));
}
template <int NW> __device__ __forceinline__ digest hash2(const u64 (&m)[NW], u64 t)
{
    const u64 h0 = B2_IV0 ^ 0x01010020ull;
    u64 v[16] = {h0, B2_IV1, B2_IV2, B2_IV3, B2_IV4, B2_IV5, B2_IV6, B2_IV7,
                 B2_IV0, B2_IV1, B2_IV2, B2_IV3, B2_IV4 ^ t, B2_IV5, ~B2_IV6, B2_IV7};
    round2<NW, 0>(v, m); round2<NW, 1>(v, m); round2<NW, 2>(v, m); round2<NW, 3>(v, m);
    round2<NW, 4>(v, m); round2<NW, 5>(v, m); round2<NW, 6>(v, m); round2<NW, 7>(v, m);
    round2<NW, 8>(v, m); round2<NW, 9>(v, m); round2<NW, 10>(v, m); round2<NW, 11>(v, m);
    digest d;
    d.w[0] = h0 ^ v[0] ^ v[8]; d.w[1] = B2_IV1 ^ v[1] ^ v[9]; d.w[2] = B2_IV2 ^ v[2] ^ v[10]; d.w[3] = B2_IV3 ^ v[3] ^ v[11];
    return d;
}
__device__ __forceinline__ digest node2(const digest& l, const digest& r)
{
    u64 m[8] = {l.w[0], l.w[1], l.w[2], l.w[3], r.w[0], r.w[1], r.w[2], r.w[3]};
    return hash2<8>(m, 64);
}

// clk[0..1]: shader-clock and 100 MHz wall-clock ticks spent by one wave in the middle of the grid
template <int KIND> __global__ __launch_bounds__(256) void chain_kernel(digest* out, uint64_t seed, uint64_t* clk = nullptr)
{
    uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool probe = clk && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
    uint64_t c0 = 0, w0 = 0;
    if (probe) { c0 = clock64(); w0 = wall_clock64(); }
    digest d;
    d.w[0] = seed + gid; d.w[1] = seed ^ (gid << 20); d.w[2] = gid * 0x9E3779B97F4A7C15ull; d.w[3] = ~gid;
#pragma clang loop unroll(disable)
    for (int it = 0; it < CHAIN; it++) {
        if constexpr (KIND == 0) d = leaf_digest(d.w[0], d.w[1] | (1ull << 63));
        else if constexpr (KIND == 1) d = node_digest(d, d);
        else d = node2(d, d);
    }
    if (d.w[0] == 0x1234567) out[0] = d;
    if (probe) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

// node level with memory traffic: K parents per thread, grid-stride so a wave's loads stay coalesced
template <int K, int BS> __global__ __launch_bounds__(BS) void level_kernel(const digest* __restrict__ in, digest* __restrict__ out, size_t m_out)
{
    const size_t stride = (size_t)gridDim.x * BS;
    size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    digest l[K], r[K];
#pragma clang loop unroll(full)
    for (int k = 0; k < K; k++) { size_t j = i + k * stride; if (j < m_out) { l[k] = in[2 * j]; r[k] = in[2 * j + 1]; } }
#pragma clang loop unroll(disable)
    for (int k = 0; k < K; k++) { size_t j = i + k * stride; if (j < m_out) out[j] = node_digest(l[k], r[k]); }
}
template <int K, int BS> int run_level(const char* name, const digest* in, digest* out, size_t m_out)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((m_out + (size_t)K * BS - 1) / ((size_t)K * BS));
    hipLaunchKernelGGL((level_kernel<K, BS>), dim3(grid), dim3(BS), 0, 0, in, out, m_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((level_kernel<K, BS>), dim3(grid), dim3(BS), 0, 0, in, out, m_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-24s %8.3f ms/launch  %7.2f Ghash/s\n", name, ms / 5, 5.0 * m_out / (ms * 1e-3) / 1e9);
    return 0;
}

template <int KIND> int run(const char* name, digest* buf)
{
    const int blocks = 256 * 8 * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, 1ull);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(chain_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, 2ull + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double hashes = 3.0 * blocks * 256 * CHAIN;
    uint64_t* clk; CHECK(hipMalloc(&clk, 16));
    hipLaunchKernelGGL(chain_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, 9ull, clk);
    uint64_t h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("%-24s %8.3f ms  %7.2f Ghash/s   shader clock under this load %.2f GHz\n", name, ms,
           hashes / (ms * 1e-3) / 1e9, (double)h[0] / ((double)h[1] * 10.0));
    CHECK(hipFree(clk));
    return 0;
}

int main()
{
    digest* buf; CHECK(hipMalloc(&buf, 4096));
    run<0>("leaf (compute only)", buf);
    run<1>("node (compute only)", buf);
    run<2>("node, alignbit rotations", buf);
    const size_t m_out = (size_t)1 << 23;
    digest *in, *out;
    CHECK(hipMalloc(&in, 2 * m_out * sizeof(digest)));
    CHECK(hipMalloc(&out, m_out * sizeof(digest)));
    CHECK(hipMemset(in, 0x5a, 2 * m_out * sizeof(digest)));
    run_level<1, 256>("level K=1 BS=256", in, out, m_out);
    run_level<1, 64>("level K=1 BS=64", in, out, m_out);
    run_level<1, 1024>("level K=1 BS=1024", in, out, m_out);
    run_level<2, 256>("level K=2 BS=256", in, out, m_out);
    run_level<4, 256>("level K=4 BS=256", in, out, m_out);
    run_level<8, 256>("level K=8 BS=256", in, out, m_out);
    return 0;
}
