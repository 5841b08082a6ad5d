// Instruction-throughput probes for the integer paths used by the 128-bit
// field core on gfx950.  Standalone: hipcc --offload-arch=gfx950 -O3 ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../stark_amd/csrc/field5.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

template <int KIND> __global__ __launch_bounds__(256) void probe(uint64_t* out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 7;
    uint64_t x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    uint32_t y0 = a, y1 = a + 1, y2 = a + 2, y3 = a + 3, y4 = a + 4, y5 = a + 5, y6 = a + 6, y7 = a + 7;
    double d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3, d4 = a + 4, d5 = a + 5, d6 = a + 6, d7 = a + 7, dm = 1.0000001;
    for (int it = 0; it < ITERS; it++) {
        if constexpr (KIND == 0) {      // v_mad_u64_u32, 8 independent chains
#define M(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
            M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
#undef M
        } else if constexpr (KIND == 1) {   // v_mul_lo_u32
#define M(y) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 2) {   // v_mul_hi_u32
#define M(y) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 3) {   // v_lshl_add_u64
#define M(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(x7));
            M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x0)
#undef M
        } else if constexpr (KIND == 4) {   // v_add_u32 (full-rate reference)
#define M(y) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 5) {   // add_co + addc_co dependent pair (64-bit add via flags), 4 pairs
#define M(lo, hi) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(lo), "+v"(hi) : "v"(b) : "vcc");
            M(y0, y1) M(y2, y3) M(y4, y5) M(y6, y7)
#undef M
        } else if constexpr (KIND == 6) {   // v_mul_u32_u24
#define M(y) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 7) {   // v_fma_f64
#define M(d) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "v"(dm));
            M(d0) M(d1) M(d2) M(d3) M(d4) M(d5) M(d6) M(d7)
#undef M
        } else if constexpr (KIND == 8) {   // v_mad_u32_u24
#define M(y) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 9) {   // add_co with SGPR-pair carries, interleaved (no vcc reuse)
            uint64_t c0, c1, c2, c3;
            asm volatile("v_add_co_u32 %0, %4, %0, %8\n\tv_add_co_u32 %1, %5, %1, %8\n\tv_add_co_u32 %2, %6, %2, %8\n\tv_add_co_u32 %3, %7, %3, %8\n\t"
                         "v_addc_co_u32 %9, %4, %9, %8, %4\n\tv_addc_co_u32 %10, %5, %10, %8, %5\n\tv_addc_co_u32 %11, %6, %11, %8, %6\n\tv_addc_co_u32 %12, %7, %12, %8, %7"
                         : "+v"(y0), "+v"(y2), "+v"(y4), "+v"(y6), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                         : "v"(b), "v"(y1), "v"(y3), "v"(y5), "v"(y7));
        } else if constexpr (KIND == 10) {  // v_mad_u64_u32 single dependent chain (latency)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(b) : "vcc");
        } else if constexpr (KIND == 11) {  // v_alignbit_b32
#define M(y) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 12) {  // v_xor3 / v_xor_b32
#define M(y) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 13) {  // v_add3_u32
#define M(y) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(y) : "v"(b));
            M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
        } else if constexpr (KIND == 14) {  // v_lshrrev_b64
#define M(x) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(x));
            M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
#undef M
        }
    }
    uint64_t r = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7;
    r ^= (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (r == 0x1234567) out[0] = r;
}

// mont_mul throughput: 4 independent chains per thread
__global__ __launch_bounds__(256) void probe_montmul(starkf::fe* out, starkf::fe seed)
{
    using namespace starkf;
    fe x0 = seed, x1 = seed, x2 = seed, x3 = seed;
    x0.lo += threadIdx.x; x1.lo += threadIdx.x * 3; x2.lo ^= threadIdx.x; x3.hi ^= threadIdx.x;
    fe w = fe{seed.hi ^ 0x55, seed.lo >> 2};
    for (int it = 0; it < 256; it++) {
        x0 = mont_mul(x0, w); x1 = mont_mul(x1, w); x2 = mont_mul(x2, w); x3 = mont_mul(x3, w);
    }
    fe r = fe_add(fe_add(x0, x1), fe_add(x2, x3));
    if (r.lo == 0x1234567 && r.hi == 77) out[0] = r;
}

// fe_add + fe_sub throughput
__global__ __launch_bounds__(256) void probe_addsub(starkf::fe* out, starkf::fe seed)
{
    using namespace starkf;
    fe x0 = seed, x1 = seed, x2 = seed, x3 = seed;
    x0.lo += threadIdx.x; x1.lo += threadIdx.x * 3; x2.lo ^= threadIdx.x; x3.lo ^= threadIdx.x * 5;
    for (int it = 0; it < 1024; it++) {
        fe a = fe_add(x0, x1), b = fe_sub(x0, x1), c = fe_add(x2, x3), d = fe_sub(x2, x3);
        x0 = a; x1 = c; x2 = b; x3 = d;
    }
    fe r = fe_add(fe_add(x0, x1), fe_add(x2, x3));
    if (r.lo == 0x1234567 && r.hi == 77) out[0] = r;
}

// mont_mul5 throughput: KIND 0 = C++ form, 1 = asm with vector twiddle, 2 = asm with scalar twiddle
template <int KIND> __global__ __launch_bounds__(256) void probe_montmul5(int32_t* out, starkf::tw5 wu, const starkf::tw5* wtab)
{
    using namespace starkf;
    fe5 x0, x1, x2, x3;
    for (int i = 0; i < 5; i++) {
        x0.l[i] = (int32_t)(threadIdx.x * 2654435761u + i * 40503u + blockIdx.x);
        x1.l[i] = x0.l[i] ^ 0x1234567; x2.l[i] = x0.l[i] >> 3; x3.l[i] = -x0.l[i];
    }
    tw5 wv = wtab[threadIdx.x & 15];
    for (int it = 0; it < 256; it++) {
        if constexpr (KIND == 0) { x0 = mont_mul5(x0, wv); x1 = mont_mul5(x1, wv); x2 = mont_mul5(x2, wv); x3 = mont_mul5(x3, wv); }
        if constexpr (KIND == 1) { x0 = mont_mul5_v(x0, wv); x1 = mont_mul5_v(x1, wv); x2 = mont_mul5_v(x2, wv); x3 = mont_mul5_v(x3, wv); }
        if constexpr (KIND == 2) { x0 = mont_mul5_s(x0, wu); x1 = mont_mul5_s(x1, wu); x2 = mont_mul5_s(x2, wu); x3 = mont_mul5_s(x3, wu); }
        if constexpr (KIND == 3) { x0 = mont_mul5(x0, wu); x1 = mont_mul5(x1, wu); x2 = mont_mul5(x2, wu); x3 = mont_mul5(x3, wu); }
    }
    fe5 r = add5(add5(x0, x1), add5(x2, x3));
    int32_t* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 5;
    for (int i = 0; i < 5; i++) o[i] = r.l[i];
}

template <int KIND> int run(const char* name, int instr_per_iter, uint64_t* d)
{
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 12345u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double waves = 5.0 * blocks * threads / 64;
    double winstr = waves * ITERS * instr_per_iter;
    double per_s = winstr / (ms * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
    double cyc = 2.4e9 * 1024 / per_s;
    printf("%-28s %8.3f ms  %9.2f Gwave-instr/s  ~%5.2f cyc/wave-instr/SIMD @2.4GHz\n", name, ms, per_s / 1e9, cyc);
    return 0;
}

int main()
{
    uint64_t* d; CHECK(hipMalloc(&d, 1024));
    run<4>("v_add_u32", 8, d);
    run<0>("v_mad_u64_u32 (8 indep)", 8, d);
    run<10>("v_mad_u64_u32 (1 dep chain)", 1, d);
    run<1>("v_mul_lo_u32", 8, d);
    run<2>("v_mul_hi_u32", 8, d);
    run<6>("v_mul_u32_u24", 8, d);
    run<8>("v_mad_u32_u24", 8, d);
    run<3>("v_lshl_add_u64", 8, d);
    run<14>("v_lshrrev_b64", 8, d);
    run<5>("add_co+nop1+addc (x4)", 8, d);
    run<9>("add_co/addc sgpr carries", 8, d);
    run<7>("v_fma_f64", 8, d);
    run<11>("v_alignbit_b32", 8, d);
    run<12>("v_xor_b32", 8, d);
    run<13>("v_add3_u32", 8, d);
    {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        starkf::fe seed{0x123456789abcdefull, 0x0123456789abcdefull};
        hipLaunchKernelGGL(probe_montmul, dim3(2048), dim3(256), 0, 0, (starkf::fe*)d, seed);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(probe_montmul, dim3(2048), dim3(256), 0, 0, (starkf::fe*)d, seed);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double muls = 5.0 * 2048 * 256 * 256 * 4;
        printf("mont_mul (u128 C code)       %8.3f ms  %9.2f Gmul/s  ~%6.1f cyc/wave-mul/SIMD\n", ms, muls / (ms * 1e-3) / 1e9,
               2.4e9 * 1024 / (muls / 64 / (ms * 1e-3)));
        hipLaunchKernelGGL(probe_addsub, dim3(2048), dim3(256), 0, 0, (starkf::fe*)d, seed);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(probe_addsub, dim3(2048), dim3(256), 0, 0, (starkf::fe*)d, seed);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double ops = 5.0 * 2048 * 256 * 1024 * 4;
        printf("fe_add/fe_sub                %8.3f ms  %9.2f Gop/s   ~%6.1f cyc/wave-op/SIMD\n", ms, ops / (ms * 1e-3) / 1e9,
               2.4e9 * 1024 / (ops / 64 / (ms * 1e-3)));
    }
    {
        using namespace starkf;
        for (int blocks : {2048, 1024, 512, 256}) {
        printf("-- mont_mul5 with %d blocks of 256 threads (%d waves/SIMD)\n", blocks, blocks / 256);
        int32_t* o[4]; tw5* wt;
        for (int k = 0; k < 4; k++) CHECK(hipMalloc(&o[k], (size_t)blocks * 256 * 5 * 4));
        std::vector<tw5> hw(16);
        for (int i = 0; i < 16; i++) for (int j = 0; j < 5; j++) hw[i].l[j] = (int32_t)((1103515245u * (i * 5 + j + 1) + 12345u) & (j == 4 ? 0xffffff : 0x3ffffff));
        CHECK(hipMalloc(&wt, sizeof(tw5) * 16));
        CHECK(hipMemcpy(wt, hw.data(), sizeof(tw5) * 16, hipMemcpyHostToDevice));
        tw5 wu = hw[3];
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const char* names[4] = {"mont_mul5 C++ (v twiddle)", "mont_mul5 asm (v twiddle)", "mont_mul5 asm (s twiddle)", "mont_mul5 C++ (s twiddle)"};
        for (int k = 0; k < 4; k++) {
            auto launch = [&](int kk) {
                if (kk == 0) hipLaunchKernelGGL(probe_montmul5<0>, dim3(blocks), dim3(256), 0, 0, o[0], wu, wt);
                if (kk == 1) hipLaunchKernelGGL(probe_montmul5<1>, dim3(blocks), dim3(256), 0, 0, o[1], wu, wt);
                if (kk == 2) hipLaunchKernelGGL(probe_montmul5<2>, dim3(blocks), dim3(256), 0, 0, o[2], wu, wt);
                if (kk == 3) hipLaunchKernelGGL(probe_montmul5<3>, dim3(blocks), dim3(256), 0, 0, o[3], wu, wt);
            };
            launch(k);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) launch(k);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double muls = 5.0 * blocks * 256 * 256 * 4;
            printf("%-28s %8.3f ms  %9.2f Gmul/s  ~%6.1f cyc/wave-mul/SIMD\n", names[k], ms, muls / (ms * 1e-3) / 1e9,
                   2.4e9 * 1024 / (muls / 64 / (ms * 1e-3)));
        }
        size_t cnt = (size_t)blocks * 256 * 5;
        std::vector<int32_t> h0(cnt), h1(cnt), h2(cnt), h3(cnt);
        CHECK(hipMemcpy(h0.data(), o[0], cnt * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h1.data(), o[1], cnt * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h2.data(), o[2], cnt * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h3.data(), o[3], cnt * 4, hipMemcpyDeviceToHost));
        size_t bad1 = 0, bad2 = 0;
        for (size_t i = 0; i < cnt; i++) { bad1 += h0[i] != h1[i]; bad2 += h3[i] != h2[i]; }
        printf("asm(v) vs C++: %zu mismatches; asm(s) vs C++: %zu mismatches\n", bad1, bad2);
        for (int k = 0; k < 4; k++) CHECK(hipFree(o[k]));
        CHECK(hipFree(wt));
        }
    }
    return 0;
}
