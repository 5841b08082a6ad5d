// Issue-rate probes for the 32-bit logic/shift/permute instructions that a
// BLAKE2b compression is made of, on gfx950.  32 independent instructions per
// loop iteration (8 chains x 4), 8 waves per SIMD.
// Standalone: hipcc --offload-arch=gfx950 -O3 ubench2.hip -o ubench2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 1024;

#define REP8(M) M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#define REP32(M) REP8(M) REP8(M) REP8(M) REP8(M)

template <int KIND> __global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 7 + threadIdx.x, c = 0x07060504u;
    uint32_t y0 = a, y1 = a + 1, y2 = a + 2, y3 = a + 3, y4 = a + 4, y5 = a + 5, y6 = a + 6, y7 = a + 7;
    for (int it = 0; it < ITERS; it++) {
        if constexpr (KIND == 0) {
#define M(y) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 1) {
#define M(y) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 2) {
#define M(y) asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 3) {
#define M(y) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 4) {
#define M(y) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 5) {
#define M(y) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 6) {
#define M(y) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 7) {
#define M(y) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 8) {
#define M(y) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 9) {
#define M(y) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 10) {
#define M(y) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 11) {
#define M(y) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 12) {
#define M(y) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(y) : "v"(b), "v"(c));
            REP32(M)
#undef M
        } else if constexpr (KIND == 13) {
#define M(y) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(y));
            REP32(M)
#undef M
        } else if constexpr (KIND == 14) {
#define M(y) asm volatile("v_xor_b32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 15) {   // 64-bit add as add_co/addc pairs (16 pairs)
#define M2(lo, hi) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(b), "v"(c) : "vcc");
#define R4 M2(y0, y1) M2(y2, y3) M2(y4, y5) M2(y6, y7)
            R4 R4 R4 R4
#undef R4
#undef M2
        } else if constexpr (KIND == 16) {
#define M(y) asm volatile("v_lshrrev_b32_e32 %0, 7, %0" : "+v"(y));
            REP32(M)
#undef M
        } else if constexpr (KIND == 17) {
#define M(y) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 18) {   // v_pk_add_u16 (VOP3P)
#define M(y) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 19) {   // xor with a 32-bit literal (8-byte VOP2)
#define M(y) asm volatile("v_xor_b32_e32 %0, 0x12345678, %0" : "+v"(y));
            REP32(M)
#undef M
        } else if constexpr (KIND == 20) {   // xor with an SGPR operand
#define M(y) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(y) : "s"(seed));
            REP32(M)
#undef M
        } else if constexpr (KIND == 22) {   // xor of selected words, written to one word of the destination (SDWA)
#define M(y) asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 23) {   // SDWA xor writing the whole destination
#define M(y) asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:BYTE_2" : "+v"(y) : "v"(b));
            REP32(M)
#undef M
        } else if constexpr (KIND == 24) {   // v_mad_i64_i32 with a vector multiplier
            uint64_t z0 = y0, z1 = y1, z2 = y2, z3 = y3, z4 = y4, z5 = y5, z6 = y6, z7 = y7;
#define M(z) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(z) : "v"(b), "v"(c) : "vcc");
#define R4 M(z0) M(z1) M(z2) M(z3) M(z4) M(z5) M(z6) M(z7)
            R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4
            y4 ^= (uint32_t)z4; y5 ^= (uint32_t)z5; y6 ^= (uint32_t)z6; y7 ^= (uint32_t)z7;
#undef R4
#undef M
            y0 ^= (uint32_t)z0; y1 ^= (uint32_t)z1; y2 ^= (uint32_t)z2; y3 ^= (uint32_t)z3;
        } else if constexpr (KIND == 25) {   // v_mad_i64_i32 with a scalar multiplier
            uint64_t z0 = y0, z1 = y1, z2 = y2, z3 = y3, z4 = y4, z5 = y5, z6 = y6, z7 = y7;
#define M(z) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(z) : "v"(b), "s"(seed) : "vcc");
#define R4 M(z0) M(z1) M(z2) M(z3) M(z4) M(z5) M(z6) M(z7)
            R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4
            y4 ^= (uint32_t)z4; y5 ^= (uint32_t)z5; y6 ^= (uint32_t)z6; y7 ^= (uint32_t)z7;
#undef R4
#undef M
            y0 ^= (uint32_t)z0; y1 ^= (uint32_t)z1; y2 ^= (uint32_t)z2; y3 ^= (uint32_t)z3;
        } else if constexpr (KIND == 26) {   // v_ashrrev_i64
            uint64_t z0 = y0, z1 = y1, z2 = y2, z3 = y3, z4 = y4, z5 = y5, z6 = y6, z7 = y7;
#define M(z) asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(z));
#define R4 M(z0) M(z1) M(z2) M(z3) M(z4) M(z5) M(z6) M(z7)
            R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4
            y4 ^= (uint32_t)z4; y5 ^= (uint32_t)z5; y6 ^= (uint32_t)z6; y7 ^= (uint32_t)z7;
#undef R4
#undef M
            y0 ^= (uint32_t)z0; y1 ^= (uint32_t)z1; y2 ^= (uint32_t)z2; y3 ^= (uint32_t)z3;
        } else if constexpr (KIND == 27) {   // v_lshl_add_u64
            uint64_t z0 = y0, z1 = y1, z2 = y2, z3 = y3, z4 = y4, z5 = y5, z6 = y6, z7 = y7, w = ((uint64_t)c << 32) | b;
#define M(z) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(z) : "v"(w));
#define R4 M(z0) M(z1) M(z2) M(z3) M(z4) M(z5) M(z6) M(z7)
            R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4 R4
            y4 ^= (uint32_t)z4; y5 ^= (uint32_t)z5; y6 ^= (uint32_t)z6; y7 ^= (uint32_t)z7;
#undef R4
#undef M
            y0 ^= (uint32_t)z0; y1 ^= (uint32_t)z1; y2 ^= (uint32_t)z2; y3 ^= (uint32_t)z3;
        } else if constexpr (KIND == 21) {   // v_mov_b32
#define M(y) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(y) : "v"(b));
            REP32(M)
#undef M
        }
    }
    uint32_t r = y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7;
    if (r == 0x1234567) out[0] = r;
}

// Straight-line code far larger than a loop body: does instruction fetch limit a
// fully unrolled hash?  KIND 0: 4096 x 4-byte xor, KIND 1: 4096 x 8-byte xor.
#define REP256(M) REP32(M) REP32(M) REP32(M) REP32(M) REP32(M) REP32(M) REP32(M) REP32(M)
#define REP4096(M) REP256(M) REP256(M) REP256(M) REP256(M) REP256(M) REP256(M) REP256(M) REP256(M) \
                   REP256(M) REP256(M) REP256(M) REP256(M) REP256(M) REP256(M) REP256(M) REP256(M)
template <int KIND> __global__ __launch_bounds__(256) void probe_long(uint32_t* out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 7 + threadIdx.x;
    uint32_t y0 = a, y1 = a + 1, y2 = a + 2, y3 = a + 3, y4 = a + 4, y5 = a + 5, y6 = a + 6, y7 = a + 7;
    for (int it = 0; it < ITERS / 128; it++) {
        if constexpr (KIND == 0) {
#define M(y) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(y) : "v"(b));
            REP4096(M)
#undef M
        } else {
#define M(y) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(y) : "v"(b));
            REP4096(M)
#undef M
        }
    }
    uint32_t r = y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7;
    if (r == 0x1234567) out[0] = r;
}

template <int KIND> int run_long(const char* name, uint32_t* d, int blocks)
{
    const int threads = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe_long<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(probe_long<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 12345u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double waves = 5.0 * blocks * threads / 64;
    double winstr = waves * ITERS * 32;
    double per_s = winstr / (ms * 1e-3);
    printf("%-28s blocks %5d %8.3f ms  %9.2f Gwave-instr/s  ~%5.2f cyc/wave-instr/SIMD @2.4GHz\n", name, blocks, ms, per_s / 1e9,
           2.4e9 * 1024 / per_s);
    return 0;
}

// shader clock vs the 100 MHz constant counter while the VALUs are busy
__global__ __launch_bounds__(256) void clock_probe(uint64_t* out, uint32_t seed)
{
    uint32_t b = seed + threadIdx.x;
    uint32_t y0 = b, y1 = b + 1, y2 = b + 2, y3 = b + 3, y4 = b + 4, y5 = b + 5, y6 = b + 6, y7 = b + 7;
    uint64_t c0 = __builtin_readcyclecounter();
    uint64_t r0 = wall_clock64();
    for (int it = 0; it < 4 * ITERS; it++) {
#define M(y) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(y) : "v"(b));
        REP32(M)
#undef M
    }
    uint64_t c1 = __builtin_readcyclecounter();
    uint64_t r1 = wall_clock64();
    uint32_t r = y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = r; }
}

template <int KIND> int run(const char* name, uint32_t* d)
{
    constexpr int PER_ITER = (KIND >= 24 && KIND <= 27) ? 256 : 32;     // instructions of the kind per loop iteration
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
    double winstr = waves * ITERS * PER_ITER;
    double per_s = winstr / (ms * 1e-3);
    double cyc = 2.4e9 * 1024 / per_s;
    printf("%-28s %8.3f ms  %9.2f Gwave-instr/s  ~%5.2f cyc/wave-instr/SIMD @2.4GHz\n", name, ms, per_s / 1e9, cyc);
    return 0;
}

int main()
{
    uint32_t* d; CHECK(hipMalloc(&d, 1024));
    run<17>("v_add_u32_e32", d);
    run<0>("v_xor_b32_e32", d);
    run<1>("v_xor_b32_e64", d);
    run<19>("v_xor_b32 literal", d);
    run<20>("v_xor_b32 sgpr", d);
    run<21>("v_mov_b32", d);
    run<16>("v_lshrrev_b32", d);
    run<2>("v_alignbit_b32", d);
    run<3>("v_perm_b32", d);
    run<4>("v_alignbyte_b32", d);
    run<5>("v_bfi_b32", d);
    run<6>("v_lshl_or_b32", d);
    run<7>("v_and_or_b32", d);
    run<8>("v_or3_b32", d);
    run<9>("v_fma_f32", d);
    run<10>("v_fmac_f32", d);
    run<11>("v_add3_u32", d);
    run<12>("v_xad_u32", d);
    run<13>("v_mov_b32_dpp quad_perm", d);
    run<14>("v_xor_b32_dpp quad_perm", d);
    run<15>("add_co+addc pair (per instr)", d);
    run<18>("v_pk_add_u16", d);
    run<22>("v_xor_b32_sdwa word->word", d);
    run<23>("v_xor_b32_sdwa dword dest", d);
    run<24>("v_mad_i64_i32 vgpr", d);
    run<25>("v_mad_i64_i32 sgpr", d);
    run<26>("v_ashrrev_i64", d);
    run<27>("v_lshl_add_u64", d);
    for (int blocks : {256, 512, 1024, 2048}) {
        run_long<0>("straight-line 16 KB xor e32", d, blocks);
        run_long<1>("straight-line 32 KB xor e64", d, blocks);
    }
    uint64_t* c; CHECK(hipMalloc(&c, 64));
    hipLaunchKernelGGL(clock_probe, dim3(2048), dim3(256), 0, 0, c, 99u);
    CHECK(hipDeviceSynchronize());
    uint64_t h[3]; CHECK(hipMemcpy(h, c, 24, hipMemcpyDeviceToHost));
    printf("shader clock under VALU load: %llu cycles in %llu ticks of 100 MHz => %.1f MHz\n",
           (unsigned long long)h[0], (unsigned long long)h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0);
    return 0;
}
