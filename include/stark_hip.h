/*
 * stark_hip.h -- C ABI of the MI355X (gfx950) implementation of the STARK
 * prover hot path: NTT / coset low-degree extension / FRI split-and-fold /
 * BLAKE2b Merkle commitment over p = 407 * 2^119 + 1.
 *
 * These are the entry points an FFI binding of the reference crate
 * (`zk_stark`, the files under src/modules) binds; each one names the reference function
 * it stands in for.  Conventions:
 *   - a field element is 16 bytes: two little-endian uint64_t {lo, hi}, a
 *     canonical integer in [0, p) (the packed image of FieldElement.value,
 *     field.rs:11-14);
 *   - every function returns 0 on success or a STARK_E* code; the message of
 *     the last failure on the calling thread is stark_last_error().  The
 *     reference panics instead (ntt.rs:15-16,26,32; merkle.rs:18,55-56;
 *     fri.rs:34,116); a binding re-raises the code as the same panic text;
 *   - functions without a suffix take HOST pointers and copy in/out;
 *     functions ending in _dev take DEVICE pointers and enqueue on the
 *     context's stream without a host sync (except where a result is returned
 *     to the host, e.g. a root);
 *   - one context per process per GPU.  Threads may share it: every call
 *     takes the context's lock (so calls on one context run one at a time,
 *     a transcript callback may call back in from its own thread) and makes
 *     the context's device current on the calling thread.  Destroy a context
 *     only when no call on it, its trees or its layers is in flight.
 */
#ifndef STARK_HIP_H
#define STARK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct stark_ctx stark_ctx;
typedef struct stark_tree stark_tree;             /* retained Merkle tree (device) */
typedef struct stark_fri_layers stark_fri_layers; /* codewords + trees of one Fri::commit */

enum {
    STARK_OK = 0,
    STARK_ENOTPOW2 = 1,   /* "NTT size must be power of 2" / "length must be power of two" */
    STARK_ETOOLARGE = 2,  /* "NTT size too large for practical use" (ntt.rs:16); input beyond a stated maximum */
    STARK_ELENGTH = 3,    /* "Input size must match NTT size" and other length mismatches */
    STARK_EINDEX = 4,     /* "cannot open invalid index" (merkle.rs:56) */
    STARK_EROUNDS = 5,    /* "Cannot do FRI with less than 1 round" (fri.rs:34) */
    STARK_EORDER = 6,     /* "error in commit: omega does not have the right order!" (fri.rs:116) */
    STARK_EARG = 7,       /* null pointer / bad argument */
    STARK_EHIP = 8,       /* HIP runtime error */
    STARK_ENOMEM = 9,     /* device or host allocation failed */
    STARK_ECALLBACK = 10  /* transcript callback returned non-zero */
};

/* ---- context ------------------------------------------------------------ */
int32_t stark_ctx_create(int32_t device, stark_ctx** out);
void stark_ctx_destroy(stark_ctx* ctx);
/* A new context launches on a stream of its own.  stark_ctx_set_stream makes it
 * launch on the caller's hipStream_t instead (NULL = HIP's default stream);
 * stark_ctx_reset_stream goes back to the context's own stream.  Leaving a
 * stream waits for the work the context queued on it, unless the stream left
 * or the stream taken is being captured into a graph (no wait is allowed
 * then): to capture, make one call of every size on the capturing stream
 * before the capture begins (tables and scratch are built then), and do not
 * leave work on another stream that the captured calls depend on. */
int32_t stark_ctx_set_stream(stark_ctx* ctx, void* hip_stream);
int32_t stark_ctx_reset_stream(stark_ctx* ctx);
int32_t stark_ctx_synchronize(stark_ctx* ctx);
/* Releases the device memory a context keeps between calls for speed (cached
 * scratch blocks of up to 8 GiB in total -- when that is reached the blocks
 * released longest ago make room for the one just released --, the transform's
 * ping buffer, twiddle tables and planes); everything is rebuilt on next use.  When an allocation
 * fails the library gives the cached scratch blocks back by itself and tries
 * once more before it reports STARK_ENOMEM; the ping buffer, tables and planes
 * are only released by this call. */
int32_t stark_ctx_trim(stark_ctx* ctx);
const char* stark_last_error(void);

/* ---- device memory helpers (for callers that keep vectors resident) ------
 * Freed blocks stay in the context's cache for the next stark_dev_alloc of the
 * same size (stark_ctx_trim gives them back to the runtime).  stark_dev_free
 * waits for the context's stream only, and the next allocation (the caller's or
 * the library's own scratch) may hand the block out at once: a caller who used
 * the block on any other stream (a framework's stream, a second context) must
 * wait for that work itself before freeing it. */
int32_t stark_dev_alloc(stark_ctx* ctx, size_t bytes, void** dptr);
int32_t stark_dev_free(stark_ctx* ctx, void* dptr);
int32_t stark_copy_h2d(stark_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int32_t stark_copy_d2h(stark_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Device to device on the context's stream, not synchronised (Vec::clone of a resident vector, fri.rs:130). */
int32_t stark_copy_d2d(stark_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);

/* ---- NTT ------------------------------------------------------------------
 * NTT::forward / NTT::inverse (ntt.rs:25-40): in place, natural order in and
 * out, omega = primitive_nth_root(n) (field.rs:96-107); the inverse uses
 * omega^-1 and scales by n^-1.  n must be a power of two <= 2^30. */
int32_t stark_ntt(stark_ctx* ctx, uint64_t* data, size_t n, int32_t inverse);
int32_t stark_ntt_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n, int32_t inverse);
/* n_cols independent transforms; column c starts at data + 2*c*col_stride (col_stride in elements, >= n). */
int32_t stark_ntt_batch(stark_ctx* ctx, uint64_t* data, size_t n, size_t n_cols, size_t col_stride,
                        int32_t inverse);
int32_t stark_ntt_batch_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n, size_t n_cols, size_t col_stride,
                            int32_t inverse);

/* Coset low-degree extension, the fast path behind Polynomial::eval_domain
 * (univariate_poly.rs:44-54) for a domain offset * omega_n^i, i < n_out:
 *   out[c][i] = sum_j coeffs[c][j] * (offset * omega^i)^j,
 * coefficients lowest degree first, n_coeffs <= n_out (zero padded), n_out a
 * power of two.  With offset = 1 this is NTT::evaluate (ntt.rs:101-107).
 * Strides are in elements. */
int32_t stark_coset_lde_batch(stark_ctx* ctx, const uint64_t* coeffs, size_t n_coeffs, size_t coeff_stride,
                              size_t n_cols, size_t n_out, const uint64_t offset[2], uint64_t* out,
                              size_t out_stride);
int32_t stark_coset_lde_batch_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n_coeffs,
                                  size_t coeff_stride, size_t n_cols, size_t n_out, const uint64_t offset[2],
                                  uint64_t* out_dev, size_t out_stride);

/* Pointwise product out[i] = a[i]*b[i] (the middle step of NTT::multiply, ntt.rs:125-127). */
int32_t stark_vec_mul(stark_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int32_t stark_vec_mul_dev(stark_ctx* ctx, const uint64_t* a_dev, const uint64_t* b_dev, uint64_t* out_dev,
                          size_t n);

/* `count` elements of a resident vector of n elements, vec_dev[indices[k]], into host memory in one round trip: the
 * leaves Stark::prove reveals at the opened indices (stark.rs:433,455).  indices and out are host arrays.
 * Synchronises. */
int32_t stark_vec_read_dev(stark_ctx* ctx, const uint64_t* vec_dev, size_t n, const uint64_t* indices, size_t count,
                           uint64_t* out);

/* ---- polynomials as vectors ------------------------------------------------
 * The algebra of Stark::prove (stark.rs:223-471) on polynomials too long for the
 * reference's O(n^2) forms: a polynomial is a coefficient vector (lowest degree
 * first) or its values on a coset, and every operation is a pass over device
 * vectors between transforms.  None of these synchronises unless it says so.
 *
 * Polynomial::compose with the linear polynomial s*x (univariate_poly.rs:203-221,
 * used as tp(omicron*x) at stark.rs:319-325): coefficient j of every column times s^j. */
int32_t stark_poly_compose_linear_dev(stark_ctx* ctx, uint64_t* coeffs_dev, size_t n, size_t n_cols,
                                      size_t col_stride, const uint64_t s[2]);
/* The inverse of stark_coset_lde_batch_dev with n_coeffs = n_out = n: values on
 * offset * omega_n^i -> the n coefficients, in place.  Polynomial::lagrange over
 * that domain (univariate_poly.rs:127-177; with offset = 1, lagrange_ntt). */
int32_t stark_coset_interpolate_batch_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n, size_t n_cols,
                                          size_t col_stride, const uint64_t offset[2]);
/* out[i] = FieldElement::inverse(a[i]) (field.rs:67-93; 0 for 0, as there): the
 * pointwise form of Polynomial Div (univariate_poly.rs:437-484) for a division
 * known to be exact, on values that avoid the divisor's zeros.  May run in place. */
int32_t stark_vec_inv_dev(stark_ctx* ctx, const uint64_t* a_dev, uint64_t* out_dev, size_t n);
/* out[j] = sum_k weights[k] * srcs[k][j - shifts[k]], j < n_out (terms outside
 * [0, lens[k]) count as zero; lens[k] + shifts[k] <= n_out; at most 16 terms):
 * Polynomial Add, scalar Mul and Mul by x^shift in one pass, the weighted sum of
 * stark.rs:385-404.  srcs, lens, shifts, weights (2 words each) are host arrays
 * of device pointers / numbers. */
int32_t stark_poly_lincomb_dev(stark_ctx* ctx, size_t n_terms, const uint64_t* const* srcs_dev, const size_t* lens,
                               const size_t* shifts, const uint64_t* weights, uint64_t* out_dev, size_t n_out);
/* MPolynomial::eval (multivariate_poly.rs:42-72) at n points at once, which on
 * the values of the point polynomials is MPolynomial::eval_symbolic (:75-88,
 * stark.rs:328-333): out[i] = sum_k coeffs[k] * prod_v cols[v][(i + rotations[v]) mod n] ^ exps[k*n_vars + v].
 * cols_dev is a host array of n_vars <= 16 device pointers, rotations (may be
 * NULL) turns a column of values on offset*omega^i into the values of the
 * polynomial composed with omega^rot * x; coeffs (2 words each) and exps (one
 * byte each) are host arrays (copied before the call returns).  If mul_dev is
 * not NULL the result is multiplied by mul_dev[i] (a zeroifier's inverse values,
 * stark.rs:336-341).  Not synchronised while the term table fits 16 KiB (about
 * 450 terms); a longer table drains the stream. */
int32_t stark_mpoly_eval_dev(stark_ctx* ctx, size_t n_vars, const uint64_t* const* cols_dev, const size_t* rotations,
                             size_t n_terms, const uint64_t* coeffs, const uint8_t* exps, const uint64_t* mul_dev,
                             uint64_t* out_dev, size_t n);
/* Polynomial::degree (univariate_poly.rs:69-85) of n coefficients, lowest first:
 * the largest index with a nonzero coefficient, 0 for the zero polynomial
 * (*is_zero_out, if not NULL, tells the two apart).  Synchronises. */
int32_t stark_poly_degree_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n, size_t* degree_out,
                              int32_t* is_zero_out);
/* The same without the wait, for a prover that checks its degree assertions
 * (stark.rs:379-380) once at the end: *result_dev (8 bytes of device memory)
 * receives 0 for the zero polynomial, else degree + 1.  Not synchronised. */
int32_t stark_poly_degree_post_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n, uint64_t* result_dev);
/* Polynomial::eval (univariate_poly.rs:87-96) of n_polys polynomials of n coefficients each (lowest degree
 * first, polynomial q at coeffs_dev + 2*q*stride words) at n_points <= 16 points given on the host:
 * out[q * n_points + p] = polynomial q at points[p] (host memory).  The verifier's evaluations of long
 * public polynomials at the opened points.  Synchronises. */
int32_t stark_poly_eval_points_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n, size_t n_polys, size_t stride,
                                   const uint64_t* points, size_t n_points, uint64_t* out);
/* FieldElement::sample (field.rs:110-116) of n byte strings of `width` bytes
 * lying back to back in device memory (the randomizer coefficients of
 * stark.rs:345-351 from their raw random bytes).  String i lands at out[i], or
 * at out[n-1-i] if reversed is non-zero: the reference fills Polynomial::coeffs,
 * highest degree first (univariate_poly.rs:27), in drawing order, and the
 * transforms here take coefficients lowest first. */
int32_t stark_sample_bytes_dev(stark_ctx* ctx, const uint8_t* bytes_dev, size_t width, size_t n, int32_t reversed,
                               uint64_t* out_dev);

/* Twiddle step of a four-step transform of n_total points split over devices
 * (no counterpart in the single-threaded reference; the composition of the
 * steps equals NTT::forward/inverse, ntt.rs:25-40): the row-major block
 * data[i][k], i < n_rows, k < n_cols, is multiplied in place by
 * w^((row_start + i) * k), w = primitive_nth_root(n_total) or its inverse. */
int32_t stark_fourstep_twiddle_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n_total, size_t row_start,
                                   size_t n_rows, size_t n_cols, int32_t inverse);

/* The two local steps around the all-to-all of a four-step transform (same
 * contract: the composition equals NTT::forward/inverse, ntt.rs:25-40).  A device
 * holds n_rows rows of a matrix with n_cols columns, row-major; pack writes what
 * it sends: dst[d][c][r] = src[r][d*cpw + c], cpw = n_cols / world, so that the
 * block for device d is the transpose of that device's share of the columns.  If
 * twiddle is non-zero, element (r, col) is multiplied on the way by
 * w^((row_start + r) * col), w = primitive_nth_root(n_total) (its inverse for
 * twiddle < 0): stark_fourstep_twiddle_dev without a pass of its own.  unpack
 * puts the blocks a device received, src[s][b][a] (s < world sources, b < n_rows,
 * a < run), side by side as rows dst[b][s][a].  Neither runs in place. */
int32_t stark_fourstep_pack_dev(stark_ctx* ctx, const uint64_t* src_dev, size_t n_rows, size_t n_cols, size_t world,
                                size_t n_total, size_t row_start, int32_t twiddle, uint64_t* dst_dev);
int32_t stark_fourstep_unpack_dev(stark_ctx* ctx, const uint64_t* src_dev, size_t world, size_t n_rows, size_t run,
                                  uint64_t* dst_dev);

/* ---- Merkle ------------------------------------------------------------------
 * Merkle::commit (merkle.rs:38-51) over the leaves
 * bincode::serialize(&FieldElement) of n field elements (fri.rs:119,
 * stark.rs:302): leaf = BLAKE2b-256(bytes), node = BLAKE2b-256(left||right).
 * n must be a power of two.  If tree_out is non-NULL the whole tree stays on
 * the device for stark_merkle_open (free with stark_tree_free). */
int32_t stark_merkle_commit_fe(stark_ctx* ctx, const uint64_t* elems, size_t n, uint8_t root[32],
                               stark_tree** tree_out);
int32_t stark_merkle_commit_fe_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t n, uint8_t root[32],
                                   stark_tree** tree_out);
/* One Merkle::commit per column of a batch, as Stark::prove commits every
 * boundary-quotient codeword (stark.rs:293-306): column c is the n elements at
 * elems_dev + c*col_stride (elements), roots_out receives n_cols x 32 bytes
 * (host).  All columns go through the same launches; trees are not retained. */
int32_t stark_merkle_roots_fe_batch_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t n, size_t n_cols,
                                        size_t col_stride, uint8_t* roots_out);
/* No counterpart in the reference.  For callers that cannot allocate (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the
 * root that stark_merkle_commit_fe_dev computes, with the tree built in a static
 * device array inside the library (16 MiB per GPU, held until the process ends)
 * instead of allocated memory.  n must be a power of two and at most 2^18
 * (STARK_ETOOLARGE beyond); root_dev_out is 32 bytes of device-accessible memory.
 * The call allocates and frees nothing and is not synchronised.  Every call a
 * process makes on one GPU uses that GPU's one array, which keeps the last
 * call's tree: calls that overlap on a GPU (two streams, two contexts) race. */
int32_t stark_merkle_root_fe_static_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t n, uint8_t* root_dev_out);
/* A Merkle tree cut across devices.  Merkle::commit pairs nodes (2i, 2i+1)
 * level by level (merkle.rs:38-51), so every aligned run of chunk_len leaves
 * (a power of two) is a subtree of its own: a device that holds n_chunks such
 * runs back to back at elems_dev reduces each to its subtree root here
 * (roots_dev_out: n_chunks x 32 bytes of device memory, not synchronised), and
 * stark_merkle_root_of_digests_dev finishes the tree from all m subtree roots
 * in leaf order (device memory, m a power of two; root is host memory).  The
 * root equals Merkle::commit over all the leaves.
 * Both can keep what they built for Merkle::open (merkle.rs:54-93): *forest_out
 * (n_chunks then a power of two) answers stark_merkle_open(forest, local leaf
 * index) with the log2(chunk_len) siblings inside that leaf's subtree, and
 * *tree_out answers stark_merkle_open(tree, chunk index) with the siblings from
 * the subtree root up; one after the other they are the path of the whole tree.
 * Pass NULL to keep nothing. */
int32_t stark_merkle_subtree_roots_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t chunk_len,
                                       size_t n_chunks, uint8_t* roots_dev_out, stark_tree** forest_out);
int32_t stark_merkle_root_of_digests_dev(stark_ctx* ctx, const uint8_t* digests_dev, size_t m, uint8_t root[32],
                                         stark_tree** tree_out);
/* Merkle::commit(&Vec<Vec<u8>>) for arbitrary byte leaves: leaf i is
 * flat[offsets[i] .. offsets[i+1]) (host pointers). */
int32_t stark_merkle_commit_bytes(stark_ctx* ctx, const uint8_t* flat, const uint64_t* offsets, size_t n,
                                  uint8_t root[32], stark_tree** tree_out);
/* Merkle::open (merkle.rs:54-93): path[0] = sibling leaf digest ... last =
 * sibling of the root's child; writes log2(n) digests of 32 bytes (for a
 * forest handle: log2(chunk_len), the siblings below the subtree root). */
int32_t stark_merkle_open(stark_tree* tree, size_t index, uint8_t* path_out, size_t* depth_out);
/* The same for `count` indices in one launch and one copy (a query round of
 * Fri::query opens 3 x num_colinearity_tests leaves, fri.rs:182-204):
 * paths_out receives count x depth digests, index by index. */
int32_t stark_merkle_open_many(stark_tree* tree, const uint64_t* indices, size_t count, uint8_t* paths_out,
                               size_t* depth_out);
size_t stark_tree_num_leaves(const stark_tree* tree);
/* All 2n-1 digests, leaf level first, each level followed by its parent level. */
int32_t stark_tree_download(stark_tree* tree, uint8_t* out);
void stark_tree_free(stark_tree* tree);

/* ---- FRI ---------------------------------------------------------------------
 * One split-and-fold step (fri.rs:133-138), N even:
 *   out[i] = 2^-1((1 + alpha/(offset*omega^i)) cw[i] + (1 - alpha/(offset*omega^i)) cw[N/2+i]). */
int32_t stark_fri_fold(stark_ctx* ctx, const uint64_t* codeword, size_t N, const uint64_t alpha[2],
                       const uint64_t offset[2], const uint64_t omega[2], uint64_t* out);
int32_t stark_fri_fold_dev(stark_ctx* ctx, const uint64_t* codeword_dev, size_t N, const uint64_t alpha[2],
                           const uint64_t offset[2], const uint64_t omega[2], uint64_t* out_dev);

/* The same fold (fri.rs:133-138) on one device's share of a layer of length N
 * that is dealt out in chunks: the share is the chunks [q*period + base,
 * q*period + base + chunk_len), q = 0 .. N/period - 1, back to back
 * (local_len = N/period * chunk_len elements; chunk_len, period powers of two,
 * period <= N/2 so that positions i and N/2+i lie in the same share).  out_dev
 * receives the share of the folded layer in the same arrangement (local_len/2
 * elements); omega and offset are those of the layer being folded.  If
 * chunk_roots_dev_out is not NULL it receives the subtree root of every output
 * chunk (local_len/2/chunk_len x 32 bytes, device memory), as
 * stark_merkle_subtree_roots_dev would compute them, and *forest_out, if not
 * NULL, keeps those subtrees for openings in the same way.  Not synchronised. */
int32_t stark_fri_fold_shard_dev(stark_ctx* ctx, const uint64_t* local_dev, size_t local_len,
                                 const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2],
                                 size_t chunk_len, size_t period, size_t base, uint64_t* out_dev,
                                 uint8_t* chunk_roots_dev_out, stark_tree** forest_out);

/* Fri::num_rounds (fri.rs:39-51). */
size_t stark_fri_num_rounds(size_t domain_length, size_t expansion_factor, size_t num_colinearity_tests);

/* The host owns the Fiat-Shamir transcript (ProofStream, proof_stream.rs).
 * Called once per round r < num_rounds with that layer's Merkle root (the
 * binding pushes hex::encode(root), fri.rs:120); for every round but the last
 * it must return alpha = FieldElement::sample(prover_fiat_shamir(32))
 * (fri.rs:126-127) in alpha_out.  `last` is 1 on the final round (alpha_out is
 * ignored).  Return non-zero to abort. */
typedef int32_t (*stark_transcript_cb)(void* user, uint32_t round, int32_t last, const uint8_t root[32],
                                       uint64_t alpha_out[2]);

/* Fri::commit (fri.rs:100-152) with the fold loop, per-layer leaf hashing and
 * trees on the device.  The codewords (all num_rounds of them, as the
 * reference returns) and their trees stay resident in *layers_out. */
int32_t stark_fri_commit(stark_ctx* ctx, const uint64_t* codeword, size_t N, const uint64_t offset[2],
                         const uint64_t omega[2], size_t expansion_factor, size_t num_colinearity_tests,
                         stark_transcript_cb cb, void* user, stark_fri_layers** layers_out);
int32_t stark_fri_commit_dev(stark_ctx* ctx, const uint64_t* codeword_dev, size_t N, const uint64_t offset[2],
                             const uint64_t omega[2], size_t expansion_factor, size_t num_colinearity_tests,
                             stark_transcript_cb cb, void* user, stark_fri_layers** layers_out);
size_t stark_fri_layers_count(const stark_fri_layers* l);
size_t stark_fri_layer_len(const stark_fri_layers* l, size_t layer);
int32_t stark_fri_layer_root(const stark_fri_layers* l, size_t layer, uint8_t root[32]);
int32_t stark_fri_layer_download(stark_fri_layers* l, size_t layer, uint64_t* out);
/* Gather `count` elements of a layer (the reads of Fri::query, fri.rs:166-179). */
int32_t stark_fri_layer_read(stark_fri_layers* l, size_t layer, const uint64_t* indices, size_t count,
                             uint64_t* out);
/* Merkle::open on a layer's retained tree (fri.rs:187,194,201). */
int32_t stark_fri_layer_open(stark_fri_layers* l, size_t layer, size_t index, uint8_t* path_out,
                             size_t* depth_out);
int32_t stark_fri_layer_open_many(stark_fri_layers* l, size_t layer, const uint64_t* indices, size_t count,
                                  uint8_t* paths_out, size_t* depth_out);
/* Everything one round of Fri::query reveals (fri.rs:155-209) in one round
 * trip: for each c in c_indices (count of them, each < len(layer)/2) the
 * values cw[layer][c], cw[layer][c + len/2], cw[layer+1][c] and their three
 * authentication paths.  values_out: 3*count elements ordered [all a | all b |
 * all c]; paths_out: [count a-paths | count b-paths] of *depth_out digests
 * each, then count c-paths of *next_depth_out digests each. */
int32_t stark_fri_layer_query(stark_fri_layers* l, size_t layer, const uint64_t* c_indices, size_t count,
                              uint64_t* values_out, uint8_t* paths_out, size_t* depth_out, size_t* next_depth_out);

/* The whole query phase of Fri::prove (fri.rs:229-238: one Fri::query per round) in one round trip.  The
 * indices of every round follow from the top-level ones (round r asks for top[s] mod len(r)/2), so nothing
 * has to come back before the next round's reads can be issued.  For each round r < layers-1, in order:
 * values_out holds 3*count elements [all a | all b | all c] as stark_fri_layer_query gives them for
 * c_indices[s] = top_indices[s] mod len(r)/2, and paths_out holds [count a-paths | count b-paths] of
 * depth(r) digests each followed by count c-paths of depth(r+1) digests each, depth(r) = log2 len(r).
 * values_out: 3*count*(layers-1) elements; paths_out: the sum over r of count*(2*depth(r)+depth(r+1))
 * digests of 32 bytes. */
int32_t stark_fri_query_all(stark_fri_layers* l, const uint64_t* top_indices, size_t count, uint64_t* values_out,
                            uint8_t* paths_out);
void stark_fri_layers_free(stark_fri_layers* l);

#ifdef __cplusplus
}
#endif
#endif /* STARK_HIP_H */
