// Pointwise polynomial work of the prover around the transforms (gfx950): the
// pieces of Stark::prove (src/modules/stark.rs:223-471) that the reference does
// with symbolic O(n^2) algebra and that become passes over vectors once the
// polynomials are held as values on a coset or as coefficient vectors.  All of
// them are bound by HBM or by the 128-bit Montgomery product, none is a
// contraction.  Field values in memory are canonical; constants travel in
// Montgomery form (x * 2^128 mod p) so that one mont_mul gives a canonical
// product.
#include <hip/hip_runtime.h>

#include "field5.hpp"
#include "internal.hpp"

namespace starkk {

using namespace starkf;

// coefficient j of every column times s^j (Polynomial::compose with the linear
// polynomial s*x, univariate_poly.rs:203-221 as used at stark.rs:319-325);
// hi[e] = s^(256 e), lo[e] = s^e in Montgomery form
__global__ __launch_bounds__(256) void compose_linear_kernel(fe* __restrict__ data, uint64_t n, uint64_t stride,
                                                             const fe* __restrict__ hi, const fe* __restrict__ lo)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    fe* p = data + (uint64_t)blockIdx.y * stride + j;
    fe w = hi[j >> 8];                                   // Montgomery form
    const uint32_t el = (uint32_t)(j & 255);
    if (el) w = mont_mul(w, lo[el]);
    *p = mont_mul(*p, w);
}

hipError_t launch_compose_linear(fe* data, uint64_t n, uint64_t n_cols, uint64_t stride, const fe* hi, const fe* lo,
                                 hipStream_t s)
{
    if (n == 0 || n_cols == 0) return hipSuccess;
    if (n_cols > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(compose_linear_kernel, dim3((uint32_t)((n + 255) / 256), (uint32_t)n_cols), dim3(256), 0, s, data,
                       n, stride, hi, lo);
    return hipGetLastError();
}

// x^(p-2) for x in Montgomery form (FieldElement::inverse, field.rs:67-93, by Fermat).
// p - 2 = 406 * 2^119 + (2^119 - 1): the nine bits of 406, then 119 one bits, which are taken seven at a
// time with x^127: 133 squarings and 25 multiplications instead of the 128 + 120 of bit-by-bit.
__device__ __forceinline__ fe mont_inverse(fe xm)
{
    // x^(2^k - 1) for k = 2, 3, 6, 7
    fe a = mont_mul(mont_mul(xm, xm), xm);                       // x^3
    a = mont_mul(mont_mul(a, a), xm);                            // x^7
    fe b = a;
    for (int i = 0; i < 3; i++) b = mont_mul(b, b);
    b = mont_mul(b, a);                                          // x^63
    const fe x127 = mont_mul(mont_mul(b, b), xm);
    // x^406, 406 = 0b110010110
    fe r = xm;
    for (int bit = 7; bit >= 0; bit--) {
        r = mont_mul(r, r);
        if ((406 >> bit) & 1) r = mont_mul(r, xm);
    }
    for (int w = 0; w < 17; w++) {                               // 119 = 17 * 7 one bits
        for (int i = 0; i < 7; i++) r = mont_mul(r, r);
        r = mont_mul(r, x127);
    }
    return r;
}

// out[i] = 1 / a[i], 0 for 0 (the reference's inverse(0) is 0).  Each thread
// inverts INV_K elements with one field inversion (Montgomery's trick): prefix
// products, one Fermat power (158 products shared by the INV_K elements), and three
// multiplications per element on the way back, for which the element is read a second time rather
// than kept in registers.  Element k of thread t in block b is b*INV_K*256 + k*256 + t, so every
// access of a wave is contiguous.
constexpr int INV_K = 16;
// (a and out may be the same vector: a thread reads an element for the second time before it writes it)
__global__ __launch_bounds__(256) void vec_inv_kernel(const fe* a, fe* out, uint64_t n, fe r2, fe one_m)
{
    const uint64_t base = (uint64_t)blockIdx.x * (INV_K * 256) + threadIdx.x;
    fe pre[INV_K];
    fe acc = one_m;
#pragma unroll
    for (int k = 0; k < INV_K; k++) {
        const uint64_t i = base + (uint64_t)k * 256;
        fe v = fe_zero();
        if (i < n) v = a[i];
        pre[k] = acc;
        if (!fe_is_zero(v)) acc = mont_mul(acc, mont_mul(v, r2));      // zeros (and the tail) ride along as 1
    }
    fe inv = mont_inverse(acc);
#pragma unroll
    for (int k = INV_K - 1; k >= 0; k--) {
        const uint64_t i = base + (uint64_t)k * 256;
        fe v = fe_zero();
        if (i < n) v = a[i];
        const bool zero = fe_is_zero(v);
        const fe r = mont_mul(inv, pre[k]);                  // (1/x_k) in Montgomery form
        if (!zero) inv = mont_mul(inv, mont_mul(v, r2));
        if (i < n) out[i] = zero ? fe_zero() : mont_mul(r, fe{1, 0});
    }
}

hipError_t launch_vec_inv(const fe* a, fe* out, uint64_t n, fe r2, fe one_m, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint64_t per_block = (uint64_t)INV_K * 256;
    hipLaunchKernelGGL(vec_inv_kernel, dim3((uint32_t)((n + per_block - 1) / per_block)), dim3(256), 0, s, a, out, n, r2,
                       one_m);
    return hipGetLastError();
}

// out[j] = sum_k w_k * src_k[j - shift_k]: a weighted sum of polynomials times
// powers of x in coefficient form (stark.rs:385-404: terms, shifts by
// x^(max_degree - bound), weights).  Weights in Montgomery form.
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs a)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n_out) return;
    fe acc = fe_zero();
    for (uint32_t k = 0; k < a.n_terms; k++) {
        const uint64_t sh = a.shift[k];
        if (j >= sh && j - sh < a.len[k]) acc = fe_add(acc, mont_mul(a.src[k][j - sh], a.weight_m[k]));
    }
    a.out[j] = acc;
}

hipError_t launch_lincomb(const LincombArgs& a, hipStream_t s)
{
    if (a.n_out == 0) return hipSuccess;
    if (a.n_terms > LINCOMB_MAX_TERMS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lincomb_kernel, dim3((uint32_t)((a.n_out + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// MPolynomial::eval (multivariate_poly.rs:42-72) at n points at once: the
// coordinates of point i are cols[v][(i + rot[v]) mod n].  A thread keeps its
// point's coordinates in LDS in Montgomery form (the term table is indexed at
// run time, which registers cannot be) and walks the term table, which is the
// same for every thread and is read through the scalar path.  If `mul` is
// given the value is multiplied by mul[i] on the way out (the division by a
// zeroifier whose inverse values are at hand, stark.rs:336-341).
constexpr int MPOLY_THREADS = 128;
constexpr int MPOLY_LAZY_TERMS = 16;       // lazy sums of this many products keep their limbs below 2^31
__global__ __launch_bounds__(MPOLY_THREADS) void mpoly_eval_kernel(MPolyArgs a)
{
    // The arithmetic is the transform passes' lazy five-limb core (219 cycles a product against 391 on two
    // 64-bit words).  Values live scaled by 2^156, the factor a five-limb product divides by, so products of
    // scaled values are scaled values; the last product, by mul[i] or by 1 unscaled, takes the factor out.
    // pt[v * MPOLY_THREADS + t]: sized by the launch for the variables there are, so that a polynomial in
    // nine variables leaves room for six workgroups on a CU instead of the three that sixteen would
    extern __shared__ fe5 pt[];
    const uint32_t t = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * MPOLY_THREADS + t;
    if (i >= a.n) return;
    for (uint32_t v = 0; v < a.n_vars; v++) {
        uint64_t idx = i + a.rot[v];
        if (idx >= a.n) idx -= a.n;
        pt[v * MPOLY_THREADS + t] = mont_mul5_s(unpack5(a.cols[v][idx]), a.scale_in);
    }
    fe5 acc{{0, 0, 0, 0, 0}};
    uint32_t pending = 0;
    // A term's sixteen exponent bytes are ONE 16-byte row, the same for every thread: it is fetched whole
    // (a scalar load) one term ahead and taken apart with scalar shifts.  Read a byte at a time the
    // exponents cost a vector-memory round trip per variable and term, several times the arithmetic.
    static_assert(MPOLY_MAX_VARS == 16, "one exponent row is one uint4");
    const uint4* __restrict__ rows = (const uint4*)a.exps;
    uint4 row_next = a.n_terms ? rows[0] : uint4{0, 0, 0, 0};
    for (uint32_t k = 0; k < a.n_terms; k++) {
        bool started = false;
        fe5 prod{};
        const uint4 row = row_next;
        if (k + 1 < a.n_terms) row_next = rows[k + 1];
        const tw5 c = a.coef5[k];                  // scalar too; asked for here, needed after the powers
        const uint64_t ex_lo = (uint64_t)row.x | ((uint64_t)row.y << 32), ex_hi = (uint64_t)row.z | ((uint64_t)row.w << 32);
        for (uint32_t v = 0; v < a.n_vars; v++) {
            const uint32_t e = (uint32_t)((v < 8 ? ex_lo >> (8 * v) : ex_hi >> (8 * (v - 8))) & 0xff);
            if (e == 0) continue;
            const fe5 b = pt[v * MPOLY_THREADS + t];
            // left-to-right over the bits of e: e is small (the AIR's degree)
            fe5 p = b;
            int top = 31 - __builtin_clz(e);
            for (int bit = top - 1; bit >= 0; bit--) {
                p = mont_mul5_v(p, p);
                if ((e >> bit) & 1) p = mont_mul5_v(p, b);
            }
            prod = started ? mont_mul5_v(prod, p) : p;
            started = true;
        }
        // the coefficient last
        prod = started ? mont_mul5_s(prod, c) : fe5{{c.l[0], c.l[1], c.l[2], c.l[3], c.l[4]}};
        acc = add5(acc, prod);
        if (++pending == MPOLY_LAZY_TERMS) {
            acc = mont_mul5_s(acc, a.scaled_one);
            pending = 1;
        }
    }
    fe5 w{{1, 0, 0, 0, 0}};
    if (a.mul) w = unpack5(a.mul[i]);
    a.out[i] = pack_canonical(mont_mul5_v(acc, w));
}

hipError_t launch_mpoly_eval(const MPolyArgs& a, hipStream_t s)
{
    if (a.n == 0) return hipSuccess;
    if (a.n_vars > MPOLY_MAX_VARS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mpoly_eval_kernel, dim3((uint32_t)((a.n + MPOLY_THREADS - 1) / MPOLY_THREADS)), dim3(MPOLY_THREADS),
                       (size_t)a.n_vars * MPOLY_THREADS * sizeof(fe5), s, a);
    return hipGetLastError();
}

// Polynomial::eval at a few points.  A thread evaluates its strip of 16 coefficients by Horner for every
// point (the coefficients are loaded once), the strips of a workgroup are then combined pairwise through
// LDS with the powers x^(16 * 2^k); what leaves the workgroup is the value of its 4096-coefficient block
// as a polynomial of its own, and the caller combines the blocks with powers of x^4096.
__global__ __launch_bounds__(256) void poly_eval_points_kernel(EvalArgs a)
{
    __shared__ fe s[256];
    const uint32_t t = threadIdx.x;
    const fe* __restrict__ c = a.coeffs + (uint64_t)blockIdx.y * a.stride;
    const uint64_t base = (uint64_t)blockIdx.x * EVAL_BLOCK_COEFFS + 16ull * t;
    fe strip[16];
#pragma unroll
    for (int j = 0; j < 16; j++) strip[j] = base + j < a.n ? c[base + j] : fe_zero();
    for (uint32_t p = 0; p < a.n_points; p++) {
        const fe x = a.x_m[p];
        fe acc = strip[15];
#pragma unroll
        for (int j = 14; j >= 0; j--) acc = fe_add(mont_mul(acc, x), strip[j]);
        s[t] = acc;
        __syncthreads();
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t step = 1u << k;
            if ((t & (2 * step - 1)) == 0) s[t] = fe_add(s[t], mont_mul(s[t + step], a.x_pow_m[p][k]));
            __syncthreads();
        }
        if (t == 0) a.partial[((uint64_t)blockIdx.y * a.n_blocks + blockIdx.x) * a.n_points + p] = s[0];
        __syncthreads();
    }
}

hipError_t launch_poly_eval_points(const EvalArgs& a, uint32_t n_polys, hipStream_t s)
{
    if (a.n == 0 || n_polys == 0 || a.n_points == 0) return hipSuccess;
    if (a.n_points > EVAL_MAX_POINTS || n_polys > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(poly_eval_points_kernel, dim3(a.n_blocks, n_polys), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Polynomial::degree (univariate_poly.rs:69-85) of a coefficient vector, lowest
// degree first: *result = 1 + the largest index holding a nonzero coefficient,
// 0 when there is none.  *result must be zero before the launch.
__global__ __launch_bounds__(256) void poly_degree_kernel(const fe* __restrict__ c, uint64_t n,
                                                          unsigned long long* __restrict__ result)
{
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t step = (uint64_t)gridDim.x * 256;
    unsigned long long best = 0;
    for (; i < n; i += step)
        if (!fe_is_zero(c[i])) best = i + 1;
    // one atomic per workgroup at most: the largest candidate of each wave, then of the four waves
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    __shared__ unsigned long long wave_best[4];
    if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) best = wave_best[w] > best ? wave_best[w] : best;
        if (best) atomicMax(result, best);
    }
}

hipError_t launch_poly_degree(const fe* c, uint64_t n, unsigned long long* result_dev, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(poly_degree_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, c, n, result_dev);
    return hipGetLastError();
}

// FieldElement::sample (field.rs:110-116) of n byte strings of `width` bytes each:
// the big-endian integer of the bytes modulo p.  The reference reduces after every byte; the value is
// the same taken fifteen bytes at a time (120 bits, below p): acc = acc * 2^120 + chunk, one product
// per chunk instead of one per byte (a leading chunk of width mod 15 bytes first).  The randomizer
// coefficients of stark.rs:345-351 from their raw random bytes.  c120_m = 2^120 in Montgomery form.
// With `reversed` string i lands at out[n-1-i]: the reference fills Polynomial::coeffs, which is
// highest degree first (univariate_poly.rs:27), in drawing order.
__global__ __launch_bounds__(256) void sample_bytes_kernel(const uint8_t* __restrict__ bytes, uint32_t width,
                                                           fe* __restrict__ out, uint64_t n, fe c120_m, int reversed)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = bytes + i * width;
    uint32_t take = width % 15;
    if (take == 0) take = 15;
    fe acc = fe_zero();
    for (uint32_t k = 0; k < width; k += take, take = 15) {
        uint64_t hi = 0, lo = 0;                               // the chunk's integer, below 2^120
        for (uint32_t b = 0; b < take; b++) {
            hi = (hi << 8) | (lo >> 56);
            lo = (lo << 8) | p[k + b];
        }
        acc = fe_add(mont_mul(acc, c120_m), fe{lo, hi});
    }
    out[reversed ? n - 1 - i : i] = acc;
}

hipError_t launch_sample_bytes(const uint8_t* bytes, uint32_t width, fe* out, uint64_t n, fe c120_m, bool reversed,
                               hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (width == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sample_bytes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, bytes, width, out, n,
                       c120_m, reversed ? 1 : 0);
    return hipGetLastError();
}

}  // namespace starkk
