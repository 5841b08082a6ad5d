// Internal launcher interface between the C-ABI layer (api.cpp) and the kernel
// translation units.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "blake2b.hpp"
#include "field.hpp"
#include "field5.hpp"

namespace starkk {

using starkf::fe;
using starkf::tw5;
using starkh::digest;

constexpr int TILE_ELEMS = 4096;      // elements per workgroup tile
constexpr int TILE_THREADS = 256;
constexpr int SMALL_MAX_LOG = 8;      // the single-workgroup kernel (ntt_small_kernel) serves coset or padded n <= 256
constexpr int SHORT_MIN_LOG = 5;      // 32 .. 256 points: one pass (a tile spans 4096 / n transforms); 512: radix 16 then 32
constexpr int MAX_PASSES = 4;

// One global pass of the multi-pass NTT.  n = R_0 * R_1 * ... * R_{P-1}; pass q
// transforms digit q (radix R_q = 2^r, stride S_q = prod_{u>q} R_u).
struct PassArgs {
    const fe* src;
    fe* dst;
    const tw5* tw256;      // omega_256^i, i<256 (times n^-1 in the last pass of an inverse transform)
    const tw5* hi;         // omega_n^(i << h), i < n >> h
    const tw5* lo;         // omega_n^i, i < 2^h
    const tw5* twist_hi;   // s^(256 i), first pass of a coset transform
    const tw5* twist_lo;   // s^i, i<256
    uint64_t src_batch_stride;   // elements between consecutive transforms of a batch
    uint64_t dst_batch_stride;
    uint64_t n_src;        // valid source elements (first pass zero-fills the rest)
    uint32_t log_n;
    uint32_t log_s;        // log2 S_q
    uint32_t log_s_next;   // log2 S_{q+1} (unused by the last pass)
    uint32_t next_is_last; // 1 when pass q+1 is the last pass
    uint32_t r;            // log2 R_q (5..8)
    uint32_t h;            // log2 of the last radix
    uint32_t nprev;        // q
    uint32_t rprev[MAX_PASSES];   // log2 R_u, u < q
    uint32_t n_batch;             // transforms in the launch (set by launch_ntt_pass)
    uint32_t inverse;             // 1: inverse transform (ntt.rs:19): the last pass stores output k at (n - k) mod n
    // Pass before the last only: its outgoing twiddle omega_n^(j_next * K) has all of n as exponent
    // range.  If set, plane[5 i .. 5 i + 4] holds the five limbs of the twiddle of output element i
    // (same index as dst); otherwise the kernel multiplies hi and lo entries together per element.
    const int32_t* plane;
    // The same twiddles as canonical 16-byte values (plane16[i] = the twiddle of output element i times 2^156 mod p):
    // aligned records and a fifth less traffic for nine more instructions per element to cut them into limbs.
    const fe* plane16;
};
// Fills the plane of the pass described by `a` (n = 2^log_n entries of five limbs).
hipError_t launch_ntt_plane(const PassArgs& a, int32_t* plane, hipStream_t s);
hipError_t launch_ntt_plane16(const PassArgs& a, fe* plane16, hipStream_t s);
// All tw5 tables hold Montgomery forms w * 2^156 mod p as five 26-bit limbs.

enum PassMode { PASS_FIRST = 0, PASS_MIDDLE = 1, PASS_LAST = 2 };

hipError_t launch_ntt_pass(const PassArgs& a, PassMode mode, bool twist, uint32_t n_batch, hipStream_t s);

struct SmallArgs {
    const fe* src;
    fe* dst;
    const fe* tw;          // omega_n^i, i < n/2 (Montgomery form)
    const fe* twist_hi;
    const fe* twist_lo;
    fe scale_m;            // n^-1 in Montgomery form (inverse only)
    uint64_t src_batch_stride, dst_batch_stride, n_src;
    uint32_t log_n;
    uint32_t has_scale, has_twist;
};
hipError_t launch_ntt_small(const SmallArgs& a, uint32_t n_batch, hipStream_t s);

// Plain transforms of at most 16 points, one thread each.
struct TinyArgs {
    const fe* src;
    fe* dst;
    uint64_t src_batch_stride, dst_batch_stride;
    uint64_t n_batch;
    tw5 scale5;            // n^-1 as a five-limb Montgomery form (inverse only)
};
constexpr int TINY_MAX_LOG = 4;
hipError_t launch_ntt_tiny(const TinyArgs& a, uint32_t log_n, bool inverse, hipStream_t s);

// data[i][k] *= w^((row_start + i) * k) for an n_rows x n_cols row-major block;
// hi/lo are the 16-byte power tables of w (lo[e] = w^e, e<256; hi[e] = w^(256 e)).
hipError_t launch_fourstep_twiddle(fe* data, uint64_t row_start, uint64_t n_rows, uint64_t n_cols, const fe* hi,
                                   const fe* lo, hipStream_t s);

// dst[d][c][r] = src[r][d*cpw + c] (times w^((row_start + r)(d*cpw + c)) when hi/lo are given), cpw = n_cols/world
hipError_t launch_fourstep_pack(const fe* src, fe* dst, uint64_t n_rows, uint64_t n_cols, uint64_t world,
                                uint64_t row_start, const fe* hi, const fe* lo, hipStream_t s);
// dst[b][s][a] = src[s][b][a], s < world, b < n_rows, a < run
hipError_t launch_fourstep_unpack(const fe* src, fe* dst, uint64_t world, uint64_t n_rows, uint64_t run, hipStream_t s);

hipError_t launch_vec_mul(const fe* a, const fe* b, fe* out, size_t n, fe r2, hipStream_t s);

// --- Merkle / FRI -----------------------------------------------------------
// Tree storage: levels concatenated, leaf digests first; level l holds n>>l
// digests at digest offset 2n - (2n >> l); the root is digest 2n-2.
hipError_t launch_leaf_hash(const fe* elems, digest* level0, size_t n, hipStream_t s);
hipError_t launch_leaf_hash_copy(const fe* elems, fe* copy, digest* level0, size_t n, hipStream_t s);
hipError_t launch_leaf_hash_bytes(const uint8_t* flat, const uint64_t* offsets, digest* level0, size_t n,
                                  hipStream_t s);
// Reduce a level of m digests (power of two) to the root, writing every level
// contiguously after it.
hipError_t launch_leaf_levels(const fe* elems, size_t col_stride, fe* copy, digest* tree, uint32_t log_n, size_t n_cols,
                              hipStream_t s, digest** level_out, size_t* m_out);
hipError_t launch_leaf_hash_batch(const fe* elems, size_t col_stride, digest* level0, uint32_t log_n, size_t n_cols,
                                  hipStream_t s);
// `level` holds n_trees trees side by side; root_out, if not null (single tree only), also receives
// the root (device-accessible memory); *roots_dev, if not null, the address of the adjacent roots
hipError_t launch_merkle_reduce(digest* level, size_t m, hipStream_t s, digest* root_out = nullptr,
                                size_t n_trees = 1, digest** roots_dev = nullptr);

struct FoldArgs {
    const fe* cw;          // N elements
    fe* out;               // N/2 elements
    digest* leaf_out;      // optional: leaf digests of out (level 0 of its tree)
    const tw5* hi;         // w^-(i << 8), five-limb Montgomery form
    const tw5* lo;         // w^-i, i<256, five-limb Montgomery form
    tw5 c5;                // 2^-1 * alpha * offset^-1, five-limb Montgomery form
    tw5 half5;             // 2^-1, five-limb Montgomery form
    uint64_t half;         // N/2
    uint32_t shift;        // omega of this layer = w^(2^shift)
    // Position of local element l in the whole layer: ((l >> log_chunk) << log_period) + base +
    // (l & (2^log_chunk - 1)).  A device that holds one chunk out of every period of a layer
    // folds its share with these; the defaults (63, 0, 0) make the position equal to l.
    uint32_t log_chunk = 63;
    uint32_t log_period = 0;
    uint64_t base = 0;
};
hipError_t launch_fri_fold(const FoldArgs& a, hipStream_t s);
hipError_t launch_fri_fold_commit(const FoldArgs& a, hipStream_t s, digest* root_out);

// One round of the query phase as the device sees it (launch_fri_query_all): layer r and layer r + 1 with their
// trees, and where round r's paths start in the output (2 count paths of `depth` digests, then count of `depth_next`).
struct QueryRound {
    const fe* cw;
    const fe* cw_next;
    const digest* tree;
    const digest* tree_next;
    uint64_t n, n_next;
    uint64_t path_at;
    uint32_t depth, depth_next;
};
hipError_t launch_fri_query_all(const QueryRound* rounds_dev, size_t n_rounds, const uint64_t* idx_dev, size_t count,
                                fe* values_out, digest* paths_out, hipStream_t s);

// Authentication path of `index` (merkle.rs:54-93): depth digests, sibling leaf first.
// (only the lowest max_levels siblings if that is less than the depth)
hipError_t launch_merkle_open(const digest* tree, size_t n, size_t index, digest* path_out, hipStream_t s,
                              uint32_t max_levels = 64);

// count openings of one tree: paths_out[b*levels + k] = sibling of idx[b] on level k (levels <= 64)
hipError_t launch_merkle_open_many(const digest* tree, size_t n, const uint64_t* idx_dev, size_t count, uint32_t levels,
                                   digest* paths_out, hipStream_t s);

hipError_t launch_gather_fe(const fe* src, const uint64_t* idx, fe* out, size_t count, hipStream_t s);

// --- pointwise polynomial work (poly_kernels.hip) ------------------------------
// data[c*stride + j] *= x^j, j < n; hi[e] = x^(256 e), lo[e] = x^e (Montgomery form)
hipError_t launch_compose_linear(fe* data, uint64_t n, uint64_t n_cols, uint64_t stride, const fe* hi, const fe* lo,
                                 hipStream_t s);
// out[i] = 1 / a[i] (0 for 0); r2 = 2^256 mod p, one_m = 2^128 mod p
hipError_t launch_vec_inv(const fe* a, fe* out, uint64_t n, fe r2, fe one_m, hipStream_t s);

constexpr uint32_t LINCOMB_MAX_TERMS = 16;
struct LincombArgs {
    const fe* src[LINCOMB_MAX_TERMS];
    uint64_t len[LINCOMB_MAX_TERMS];
    uint64_t shift[LINCOMB_MAX_TERMS];
    fe weight_m[LINCOMB_MAX_TERMS];      // Montgomery form
    fe* out;
    uint64_t n_out;
    uint32_t n_terms;
};
hipError_t launch_lincomb(const LincombArgs& a, hipStream_t s);

// Polynomial::eval (univariate_poly.rs:87-96) of several polynomials at a few points: workgroup b of
// polynomial q reduces its 4096 coefficients to sum_j c[4096 b + j] x^j for every point x.
constexpr uint32_t EVAL_MAX_POINTS = 16;
constexpr uint32_t EVAL_BLOCK_COEFFS = 4096;
struct EvalArgs {
    const fe* coeffs;                    // polynomial q at coeffs + q * stride, lowest degree first
    uint64_t n, stride;
    fe* partial;                         // [q][block][point]
    uint32_t n_points, n_blocks;
    fe x_m[EVAL_MAX_POINTS];             // the points, Montgomery form
    fe x_pow_m[EVAL_MAX_POINTS][8];      // x^(16 * 2^k), k < 8, Montgomery form
};
hipError_t launch_poly_eval_points(const EvalArgs& a, uint32_t n_polys, hipStream_t s);

constexpr uint32_t MPOLY_MAX_VARS = 16;
struct MPolyArgs {
    const fe* cols[MPOLY_MAX_VARS];      // variable v at point i: cols[v][(i + rot[v]) mod n]
    uint64_t rot[MPOLY_MAX_VARS];
    const tw5* coef5;                    // n_terms coefficients as five-limb forms c * 2^156 (device memory)
    const uint8_t* exps;                 // n_terms x MPOLY_MAX_VARS exponents (device memory)
    const fe* mul;                       // optional pointwise factor
    fe* out;
    uint64_t n;
    uint32_t n_vars, n_terms;
    tw5 scale_in;                        // five-limb form of 2^156: brings a loaded value into the scaled domain
    tw5 scaled_one;                      // five-limb form of 1: a product by it only renormalises the limbs
};
hipError_t launch_mpoly_eval(const MPolyArgs& a, hipStream_t s);
// *result_dev (zeroed by the caller) = 1 + largest index of a nonzero coefficient, 0 if none
hipError_t launch_poly_degree(const fe* c, uint64_t n, unsigned long long* result_dev, hipStream_t s);
// out[i] (out[n-1-i] if reversed) = big-endian integer of bytes[i*width .. (i+1)*width) modulo p
hipError_t launch_sample_bytes(const uint8_t* bytes, uint32_t width, fe* out, uint64_t n, fe c120_m, bool reversed,
                               hipStream_t s);

}  // namespace starkk
