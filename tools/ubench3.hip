// Cost of a lazy-limb product by a wave-uniform constant on gfx950: the general form
// (mont_mul5_s: 31 multiply-adds, six Montgomery digit steps, ten digit extractions) against the
// tabulated form (mont_mul5_c: 27 multiply-adds, two digit steps, six extractions).  Each thread runs
// the 16-point in-register butterfly network of the transform passes (17 constant products and 64
// limb additions per network) over and over; the two forms must leave the same values.
// Standalone: hipcc --offload-arch=gfx950 -O3 -I stark_amd/csrc tools/ubench3.hip -o tools/_build/ubench3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "field5.hpp"
#include "field_host.hpp"
#include "tw16_tables.hpp"

using namespace starkf;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <bool TAB>
__device__ __forceinline__ void network(fe5 (&v)[16], const tw5* __restrict__ tw16, const tw25* __restrict__ tc16)
{
#pragma unroll
    for (int lg = 4; lg >= 1; lg--) {
        const int len = 1 << lg, half = len >> 1;
#pragma unroll
        for (int s = 0; s < 16; s += len) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const fe5 a = v[s + j], b = v[s + j + half];
                v[s + j] = add5(a, b);
                const fe5 d = sub5(a, b);
                const int e = j * (16 / len);
                if (e == 0) v[s + j + half] = d;
                else if constexpr (TAB) v[s + j + half] = mont_mul5_c(d, TW16C[0][e]);
                else v[s + j + half] = mont_mul5_s(d, tw16[e]);
            }
        }
    }
}

template <bool TAB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void probe(const fe* __restrict__ in, fe* __restrict__ out,
                                                                                         const tw5* __restrict__ tw16,
                                                                                         const tw25* __restrict__ tc16, int iters)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    fe5 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = unpack5(in[g * 16 + i]);
    for (int it = 0; it < iters; it++) {
        network<TAB>(v, tw16, tc16);
        // every element through a product, as between the rounds of a pass: brings the limbs back to 26 bits
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if constexpr (TAB) v[i] = mont_mul5_c(v[i], TW16C[0][1 + (i & 3)]);
            else v[i] = mont_mul5_s(v[i], tw16[1 + (i & 3)]);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[g * 16 + i] = pack_canonical(v[i]);
}

static tw5 host_tw5(const HostField& F, fe x)
{
    fe r{1, 0};
    for (int i = 0; i < 156; i++) r = fe_add(r, r);
    fe5 l = unpack5(F.mul(x, r));
    tw5 t{};
    for (int i = 0; i < 5; i++) t.l[i] = l.l[i];
    return t;
}

static tw25 host_tw25(const HostField& F, fe x)
{
    tw25 t{};
    fe cur = x;
    for (int i = 0; i < 52; i++) cur = fe_add(cur, cur);
    for (int i = 0; i < 5; i++) {
        fe5 l = unpack5(cur);
        for (int k = 0; k < 5; k++) t.l[k][i] = l.l[k];
        for (int b = 0; b < 26; b++) cur = fe_add(cur, cur);
    }
    return t;
}

int main()
{
    HostField F;
    const fe w16 = F.primitive_nth_root(4);
    std::vector<tw5> h5(8);
    std::vector<tw25> h25(8);
    fe cur{1, 0};
    for (int i = 0; i < 8; i++) {
        h5[i] = host_tw5(F, cur);
        h25[i] = host_tw25(F, cur);
        cur = F.mul(cur, w16);
    }
    const int blocks = 256 * 12, iters = 64;
    const size_t n = (size_t)blocks * 256 * 16;
    std::vector<fe> hin(n);
    uint64_t s = 0x5741524B;
    for (auto& x : hin) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        x.lo = s;
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        x.hi = s >> 2;      // below p
    }
    fe *din, *d0, *d1;
    tw5* dt5;
    tw25* dt25;
    CHECK(hipMalloc(&din, n * sizeof(fe)));
    CHECK(hipMalloc(&d0, n * sizeof(fe)));
    CHECK(hipMalloc(&d1, n * sizeof(fe)));
    CHECK(hipMalloc(&dt5, 8 * sizeof(tw5)));
    CHECK(hipMalloc(&dt25, 8 * sizeof(tw25)));
    CHECK(hipMemcpy(din, hin.data(), n * sizeof(fe), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dt5, h5.data(), 8 * sizeof(tw5), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dt25, h25.data(), 8 * sizeof(tw25), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) {
        for (int k = 0; k < 2; k++) {
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 4; r++) {
                if (k == 0) hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(256), 0, 0, din, d0, dt5, dt25, iters);
                else hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(256), 0, 0, din, d1, dt5, dt25, iters);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms[k], e0, e1));
        }
        const double products = 4.0 * blocks * 256 * iters * 33;
        printf("general   %8.3f ms  %7.1f Gproduct/s    tabulated %8.3f ms  %7.1f Gproduct/s   ratio %.3f\n", ms[0],
               products / ms[0] / 1e6, ms[1], products / ms[1] / 1e6, ms[1] / ms[0]);
    }
    std::vector<fe> h0(n), h1(n);
    CHECK(hipMemcpy(h0.data(), d0, n * sizeof(fe), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h1.data(), d1, n * sizeof(fe), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += (h0[i].lo != h1[i].lo || h0[i].hi != h1[i].hi);
    printf("%zu elements, %zu differ between the two forms\n", n, bad);
    return bad ? 2 : 0;
}
