// C-ABI layer: contexts, twiddle/power tables, scratch, and the entry points
// declared in include/stark_hip.h.
#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/stark_hip.h"
#include "field_host.hpp"
#include "internal.hpp"
#include "tw16_tables.hpp"

using namespace starkf;
using starkh::digest;
using namespace starkk;

static thread_local std::string g_last_error;

static int32_t fail(int32_t code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? STARK_ENOMEM : STARK_EHIP,                     \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

#define ST_TRY(expr)                  \
    do {                              \
        int32_t rc_ = (expr);         \
        if (rc_ != STARK_OK) return rc_; \
    } while (0)

namespace {

struct NttTables {
    uint32_t log_n = 0, h = 0;
    std::vector<uint32_t> radices;   // empty for the single-workgroup path
    tw5* tw256 = nullptr;            // multi-pass path: five-limb Montgomery forms (2^156)
    tw5* hi = nullptr;
    tw5* lo = nullptr;               // omega^i, i < 2^h; times n^-1 for the inverse
    int32_t* plane = nullptr;        // outgoing twiddles of the pass before the last, 5 limbs per element (built on first use)
    fe* plane16 = nullptr;           // the same twiddles as canonical 16-byte values (see PassArgs::plane16)
    bool plane_refused = false;      // the allocation failed once: do not try again
    fe* small = nullptr;             // omega_n^i, i < n/2 (n <= 4096), Montgomery form (2^128)
    fe scale_m{0, 0};                // n^-1, Montgomery form (2^128)
    tw5 scale5{};                    // n^-1, five-limb Montgomery form (2^156)
};

// powers of one element x: lo[i] = x^i (i<256), hi[i] = x^(256 i); either as
// 16-byte Montgomery forms (2^128) or as five-limb forms (2^156)
// Handed out as shared pointers: a commit keeps its tables alive across the transcript callback
// (which may call back into the same context and push other tables through the bounded cache), and
// the device memory goes when the last holder lets go.
struct PowTables {
    void* hi = nullptr;
    void* lo = nullptr;
    size_t hi_len = 0;
    uint64_t stamp = 0;                                // last use, for eviction
    PowTables() = default;
    PowTables(const PowTables&) = delete;
    PowTables& operator=(const PowTables&) = delete;
    ~PowTables()
    {
        if (hi) (void)hipFree(hi);
        if (lo) (void)hipFree(lo);
    }
};
typedef std::shared_ptr<PowTables> PowRef;

typedef std::tuple<uint64_t, uint64_t, uint64_t, int> PowKey;   // (x.lo, x.hi, hi_len, five-limb?)

}  // namespace

// Scratch of stark_merkle_root_fe_static_dev: the whole tree of up to STATIC_TREE_MAX_LEAVES leaves
// (2n - 1 digests of 32 bytes), 16 MiB that every GPU this library runs on holds until the process
// ends.  Host code never uses this name as an address: stark_ctx_create asks the runtime for the
// array's address on the context's device and keeps it in stark_ctx::static_scratch.
constexpr size_t STATIC_TREE_MAX_LEAVES = (size_t)1 << 18;
static __device__ __align__(256) unsigned char scratch[2 * STATIC_TREE_MAX_LEAVES * sizeof(digest)];

struct stark_ctx {
    int device = 0;
    // Sizes from 2^this on keep the twiddle plane of the pass before the last as canonical 16-byte values instead
    // of five limbs (20 bytes): measured 0.5 % faster at 2^24 and 1.6 % at 2^26, where that pass is at the
    // memory system's ceiling, no different at 2^22 and 1 % slower below (tools/plane16_ab.py), and a fifth
    // less table memory.  STARK_NTT_PLANE16_FROM overrides (a value above 30 turns it off).
    uint32_t plane16_min_log = 24;
    void* static_scratch = nullptr;                    // device address of `scratch` on this context's device
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    HostField F;
    std::map<uint32_t, NttTables> ntt_tables;          // key = log_n*2 + inverse
    std::map<PowKey, PowRef> pow_tables;
    uint64_t pow_clock = 0;
    fe* scratch = nullptr;
    size_t scratch_elems = 0;
    struct PoolBlock {
        void* p;
        uint64_t stamp;                                // when it was released (pool_clock)
    };
    std::multimap<size_t, PoolBlock> pool;             // cached device blocks by size
    size_t pool_large_bytes = 0;                       // bytes of the cached blocks of at least 1 MiB
    uint64_t pool_clock = 0;
    std::map<void*, size_t> dev_blocks;                // blocks handed out by stark_dev_alloc, by address
    void* pinned = nullptr;                            // small pinned staging buffer (4 KiB)
    void* pin_io = nullptr;                            // 2 x 64 KiB host-mapped: small transforms run on it in place
    std::recursive_mutex mu;
    // Lanes for large transfers between pageable host memory and the device: each has a stream
    // of its own and two pinned chunks (created on first use).
    struct Lane {
        hipStream_t stream = nullptr;
        void* pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
    };
    std::vector<Lane> lanes;
    hipEvent_t lane_gate = nullptr;
    // Small tables that go up with a launch (the term table of a constraint): a ring of slots, each a pinned
    // host block, its device twin and an event recorded behind the kernel that reads the twin.  A call fills the
    // next slot and waits only if that slot's previous kernel has not finished, so the call itself never
    // drains the stream.
    struct SmallSlot {
        void* pin = nullptr;
        void* dev = nullptr;
        hipEvent_t done = nullptr;
        bool pending = false;
    };
    static constexpr size_t SMALL_SLOT_BYTES = 16384;
    static constexpr size_t SMALL_SLOTS = 8;
    SmallSlot small[SMALL_SLOTS];
    size_t small_next = 0;
};

struct stark_tree {
    stark_ctx* ctx;
    digest* d;        // 2n-1 digests
    size_t n;
    bool owns;
    // levels above the leaves that hold digests: log2(n) for a whole tree; log2(chunk) for a forest
    // of equal subtrees side by side (same layout, the levels above the subtree roots never computed)
    uint32_t levels;
};

// Held for the duration of every entry point: a context is shared by the threads of a
// process (cargo runs the reference's tests on several), so calls on it are serialised, and the
// calling thread is pointed at the context's device (the current device is per thread in HIP).
// Recursive because entry points call one another and a transcript callback may call back in.
struct CtxGuard {
    std::unique_lock<std::recursive_mutex> lock;
    explicit CtxGuard(stark_ctx* c);
};

struct stark_fri_layers {
    stark_ctx* ctx;
    fe* cw_arena = nullptr;
    digest* tree_arena = nullptr;
    size_t cw_bytes = 0, tree_bytes = 0;
    std::vector<size_t> len;
    std::vector<fe*> cw;
    std::vector<digest*> tree;
    std::vector<std::array<uint8_t, 32>> roots;
};

CtxGuard::CtxGuard(stark_ctx* c)
{
    if (!c) return;
    lock = std::unique_lock<std::recursive_mutex>(c->mu);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) (void)hipSetDevice(c->device);
}

namespace {

static int log2_exact(size_t n)
{
    if (n == 0 || (n & (n - 1))) return -1;
    int l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

// Device scratch is recycled through a small cache keyed by block size.  Requests below 1 MiB are
// rounded up to a power of two (at least 256 B) so that the many short-lived buffers of a query
// phase (index lists, paths, a few elements) hit the cache whatever their exact length: a
// hipMalloc/hipFree pair costs far more than the kernels they serve, and hipFree drains the device.
static size_t pool_bucket(size_t bytes)
{
    if (bytes >= ((size_t)1 << 20)) return bytes;
    size_t b = 256;
    while (b < bytes) b <<= 1;
    return b;
}

// Blocks of at least 1 MiB that the cache may hold: a count and a byte total (a 2^24 FRI commit
// parks a 0.5 GiB and a 2 GiB arena, a 2^26 one 2 + 8 GiB).
constexpr size_t POOL_LARGE = (size_t)1 << 20;
constexpr size_t POOL_MAX_LARGE_BLOCKS = 64;
constexpr size_t POOL_MAX_LARGE_BYTES = (size_t)8 << 30;

static void pool_drop(stark_ctx* ctx)
{
    for (auto& kv : ctx->pool) (void)hipFree(kv.second.p);
    ctx->pool.clear();
    ctx->pool_large_bytes = 0;
}

// hipMalloc that gives the cached blocks back to the runtime and retries once when memory is short
static hipError_t malloc_retry(stark_ctx* ctx, void** out, size_t bytes)
{
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory && !ctx->pool.empty()) {
        (void)hipGetLastError();
        pool_drop(ctx);
        e = hipMalloc(out, bytes);
    }
    return e;
}

static int32_t pool_alloc(stark_ctx* ctx, size_t bytes, void** out)
{
    bytes = pool_bucket(bytes);
    auto it = ctx->pool.find(bytes);
    if (it != ctx->pool.end()) {
        *out = it->second.p;
        ctx->pool.erase(it);
        if (bytes >= POOL_LARGE) ctx->pool_large_bytes -= bytes;
        return STARK_OK;
    }
    hipError_t e = malloc_retry(ctx, out, bytes);
    if (e != hipSuccess) return fail(STARK_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return STARK_OK;
}

static void pool_free(stark_ctx* ctx, void* p, size_t bytes)
{
    if (!p) return;
    bytes = pool_bucket(bytes);
    // keep at most a handful of large blocks, bounded in bytes too, and a few hundred small ones
    bool keep = ctx->pool.size() < 512;
    if (keep && bytes >= POOL_LARGE) {
        // The block just released is the likeliest to be asked for again (a loop of calls of one size), so
        // when the cache is full it is the blocks released longest ago that go, not this one: a cache that
        // refuses the new block instead makes every later call of that size pay a hipMalloc and a hipFree
        // (6 ms per call at 64 MiB once a 2^24 FRI commit's arenas had filled it).
        keep = bytes <= POOL_MAX_LARGE_BYTES;
        while (keep) {
            size_t large = 0;
            auto oldest = ctx->pool.end();
            for (auto it = ctx->pool.begin(); it != ctx->pool.end(); ++it) {
                if (it->first < POOL_LARGE) continue;
                large++;
                if (oldest == ctx->pool.end() || it->second.stamp < oldest->second.stamp) oldest = it;
            }
            if (large < POOL_MAX_LARGE_BLOCKS && ctx->pool_large_bytes + bytes <= POOL_MAX_LARGE_BYTES) break;
            if (oldest == ctx->pool.end()) break;
            (void)hipFree(oldest->second.p);
            ctx->pool_large_bytes -= oldest->first;
            ctx->pool.erase(oldest);
        }
    }
    if (!keep) {
        (void)hipFree(p);
        return;
    }
    ctx->pool.emplace(bytes, stark_ctx::PoolBlock{p, ++ctx->pool_clock});
    if (bytes >= POOL_LARGE) ctx->pool_large_bytes += bytes;
}

template <class T>
static int32_t upload(stark_ctx* ctx, const std::vector<T>& v, T** out)
{
    *out = nullptr;
    HIP_TRY(malloc_retry(ctx, (void**)out, v.size() * sizeof(T)));
    hipError_t e = hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(*out);
        *out = nullptr;
        HIP_TRY(e);
    }
    return STARK_OK;
}

// powers x^0, x^1, ... (canonical x) in Montgomery form
static std::vector<fe> powers_m(const HostField& F, fe x, size_t count)
{
    std::vector<fe> v(count);
    fe xm = F.to_mont(x);
    fe cur = F.one_m;
    for (size_t i = 0; i < count; i++) {
        v[i] = cur;
        cur = mont_mul(cur, xm);
    }
    return v;
}

// canonical value -> five-limb Montgomery form (x * 2^156 mod p)
static tw5 to_tw5(const HostField& F, fe x)
{
    static const fe r156 = [] {
        fe r{1, 0};
        for (int i = 0; i < 156; i++) r = fe_add(r, r);
        return r;
    }();
    fe5 l = unpack5(F.mul(x, r156));
    tw5 t{};
    for (int i = 0; i < 5; i++) t.l[i] = l.l[i];
    return t;
}

// powers x^0, x^1, ... as five-limb Montgomery forms
static std::vector<tw5> powers_5(const HostField& F, fe x, size_t count, fe first = fe{1, 0})
{
    std::vector<tw5> v(count);
    fe xm = F.to_mont(x);
    fe cur = first;
    for (size_t i = 0; i < count; i++) {
        v[i] = to_tw5(F, cur);
        cur = mont_mul(cur, xm);       // canonical * Montgomery form -> canonical
    }
    return v;
}

// Does row [inverse] of the literal table TW16C hold the tabulated forms (limb k of w16^e * 2^(26 i + 52))
// of the powers of w16?
static bool tw16_literals_match(const HostField& F, fe w16, bool inverse)
{
    fe cur{1, 0};
    for (int e = 0; e < 8; e++) {
        fe sh = cur;
        for (int b = 0; b < 52; b++) sh = fe_add(sh, sh);
        for (int i = 0; i < 5; i++) {
            const fe5 l = unpack5(sh);
            for (int k = 0; k < 5; k++)
                if (TW16C[inverse ? 1 : 0][e].l[k][i] != l.l[k]) return false;
            for (int b = 0; b < 26; b++) sh = fe_add(sh, sh);
        }
        cur = F.mul(cur, w16);
    }
    return true;
}

// Sizes up to this keep a twiddle plane (20 bytes per element, shared by both directions: 1.3 GB at 2^26).
constexpr uint32_t PLANE_MAX_LOG = 26;

static std::vector<uint32_t> plan_radices(uint32_t log_n)
{
    std::vector<uint32_t> r;
    if (log_n < (uint32_t)SHORT_MIN_LOG) return r;
    if (log_n <= 8) return {log_n};                           // one pass; a tile spans 4096 / n transforms
    if (log_n == 9) return {4, 5};                            // radix 16 (no exchange), then 32
    uint32_t P = (log_n + 7) / 8;
    uint32_t base = log_n / P, rem = log_n % P;
    for (uint32_t q = 0; q < P; q++) r.push_back(base + (q < rem ? 1 : 0));
    return r;
}

static void free_ntt_tables(NttTables& t)
{
    (void)hipFree(t.tw256); (void)hipFree(t.hi); (void)hipFree(t.lo);
    (void)hipFree(t.small); (void)hipFree(t.plane); (void)hipFree(t.plane16);
    t.tw256 = t.hi = t.lo = nullptr;
    t.small = nullptr;
    t.plane = nullptr;
    t.plane16 = nullptr;
}

static int32_t get_ntt_tables(stark_ctx* ctx, uint32_t log_n, bool inverse, const NttTables** out)
{
    uint32_t key = log_n * 2 + (inverse ? 1 : 0);
    auto it = ctx->ntt_tables.find(key);
    if (it != ctx->ntt_tables.end()) {
        *out = &it->second;
        return STARK_OK;
    }
    const HostField& F = ctx->F;
    NttTables t;
    t.log_n = log_n;
    fe omega = F.primitive_nth_root(log_n);                   // ntt.rs:18
    if (inverse) omega = F.inverse(omega);                    // ntt.rs:19
    fe n_inv = F.inverse(fe{(u64)1 << log_n, 0});             // ntt.rs:36
    t.scale_m = F.to_mont(n_inv);
    t.scale5 = to_tw5(F, n_inv);
    int32_t rc = STARK_OK;
    // the lazy-limb kernels carry the powers of the 16th root as literals (tw16_tables.hpp); they must be
    // the powers of THIS field's root (omega^(n/16) where n allows)
    {
        fe w16 = F.primitive_nth_root(4);
        if (inverse) w16 = F.inverse(w16);
        if (!tw16_literals_match(F, w16, inverse))
            return fail(STARK_EARG, "the compiled-in 16th-root table does not match primitive_nth_root");
    }
    if (log_n <= 8) {                                         // the single-workgroup kernel: coset and padded transforms
        size_t half = log_n ? ((size_t)1 << (log_n - 1)) : 1;
        rc = upload(ctx, powers_m(F, omega, half), &t.small);
    }
    if (rc == STARK_OK && log_n >= (uint32_t)SHORT_MIN_LOG) {
        // The tiled passes run an inverse transform as the forward one with its outputs stored in reversed
        // order (sum_j x[j] w^(-jk) is the forward sum at index -k), so both directions share the forward
        // root's hi/lo tables and twiddle plane; the inverse's entry only holds the exchange twiddles of its
        // last pass, scaled by n^-1 (ntt.rs:36-39: every element meets exactly one of them).
        t.radices = plan_radices(log_n);
        t.h = t.radices.back();
        const fe w256 = F.primitive_nth_root(8);              // omega^(n/256) where n allows
        rc = upload(ctx, powers_5(F, w256, 256, inverse ? n_inv : fe{1, 0}), &t.tw256);
        if (!inverse) {
            const fe w_hi = F.pow(omega, (u128)1 << t.h);
            if (rc == STARK_OK) rc = upload(ctx, powers_5(F, w_hi, (size_t)1 << (log_n - t.h)), &t.hi);
            if (rc == STARK_OK) rc = upload(ctx, powers_5(F, omega, (size_t)1 << t.h), &t.lo);
        }
    }
    if (rc != STARK_OK) {
        free_ntt_tables(t);                                   // nothing half-built stays behind
        return rc;
    }
    auto ins = ctx->ntt_tables.emplace(key, t);
    *out = &ins.first->second;
    return STARK_OK;
}

static int32_t get_pow_tables(stark_ctx* ctx, fe x, size_t count, bool five, PowRef* out)
{
    size_t hi_len = (count + 255) / 256;
    if (hi_len == 0) hi_len = 1;
    PowKey key(x.lo, x.hi, hi_len, five ? 1 : 0);
    auto it = ctx->pow_tables.find(key);
    if (it != ctx->pow_tables.end()) {
        it->second->stamp = ++ctx->pow_clock;
        *out = it->second;
        return STARK_OK;
    }
    while (ctx->pow_tables.size() >= 32) {                    // bounded cache: drop the least recently used entry;
        auto victim = ctx->pow_tables.begin();                // whoever still holds it keeps it alive
        for (auto jt = ctx->pow_tables.begin(); jt != ctx->pow_tables.end(); ++jt)
            if (jt->second->stamp < victim->second->stamp) victim = jt;
        ctx->pow_tables.erase(victim);
    }
    const HostField& F = ctx->F;
    PowRef t = std::make_shared<PowTables>();                 // a failed upload below frees what was uploaded
    t->hi_len = hi_len;
    t->stamp = ++ctx->pow_clock;
    if (five) {
        ST_TRY(upload(ctx, powers_5(F, x, 256), (tw5**)&t->lo));
        ST_TRY(upload(ctx, powers_5(F, F.pow(x, 256), hi_len), (tw5**)&t->hi));
    } else {
        ST_TRY(upload(ctx, powers_m(F, x, 256), (fe**)&t->lo));
        ST_TRY(upload(ctx, powers_m(F, F.pow(x, 256), hi_len), (fe**)&t->hi));
    }
    ctx->pow_tables.emplace(key, t);
    *out = t;
    return STARK_OK;
}

static int32_t ensure_scratch(stark_ctx* ctx, size_t elems)
{
    if (ctx->scratch_elems >= elems) return STARK_OK;
    if (ctx->scratch) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_elems = 0;
    }
    HIP_TRY(malloc_retry(ctx, (void**)&ctx->scratch, elems * sizeof(fe)));
    ctx->scratch_elems = elems;
    return STARK_OK;
}

// Transform n_cols columns: column c reads n_src elements at src + c*src_stride
// (zero padded to n), writes n elements at dst + c*dst_stride.  All device
// pointers.  twist == nullptr means offset 1.
static int32_t run_ntt(stark_ctx* ctx, const fe* src, size_t src_stride, size_t n_src, fe* dst,
                       size_t dst_stride, uint32_t log_n, bool inverse, const fe* twist_x, size_t n_cols)
{
    if (n_cols == 0) return STARK_OK;
    const NttTables* t;
    ST_TRY(get_ntt_tables(ctx, log_n, inverse, &t));
    const NttTables* tf = t;                                  // the forward root's tables (shared by the tiled passes)
    if (inverse && log_n >= (uint32_t)SHORT_MIN_LOG) ST_TRY(get_ntt_tables(ctx, log_n, false, &tf));
    const size_t n = (size_t)1 << log_n;
    // the tiled passes take everything from 512 points up, and plain transforms (no coset shift, no zero
    // padding) from 32 points; plain transforms of up to 16 points run one per thread; the rest (coset
    // and padded transforms below 512 points) runs in one workgroup each
    const bool tiled = log_n >= 9 || (log_n >= (uint32_t)SHORT_MIN_LOG && !twist_x && n_src == n);
    PowRef twist;
    if (twist_x) ST_TRY(get_pow_tables(ctx, *twist_x, n_src, tiled, &twist));
    if (log_n <= (uint32_t)TINY_MAX_LOG && !twist_x && n_src == n) {       // one thread per transform
        TinyArgs a{};
        a.src = src;
        a.dst = dst;
        a.src_batch_stride = src_stride;
        a.dst_batch_stride = dst_stride;
        a.n_batch = n_cols;
        a.scale5 = t->scale5;
        HIP_TRY(launch_ntt_tiny(a, log_n, inverse, ctx->stream));
        return STARK_OK;
    }
    if (!tiled) {
        SmallArgs a{};
        a.tw = t->small;
        a.twist_hi = twist ? (const fe*)twist->hi : nullptr;
        a.twist_lo = twist ? (const fe*)twist->lo : nullptr;
        a.scale_m = t->scale_m;
        a.src_batch_stride = src_stride;
        a.dst_batch_stride = dst_stride;
        a.n_src = n_src;
        a.log_n = log_n;
        a.has_scale = inverse ? 1 : 0;
        a.has_twist = twist ? 1 : 0;
        for (size_t c0 = 0; c0 < n_cols; c0 += 65535) {
            size_t nb = n_cols - c0 < 65535 ? n_cols - c0 : 65535;
            a.src = src + c0 * src_stride;
            a.dst = dst + c0 * dst_stride;
            HIP_TRY(launch_ntt_small(a, (uint32_t)nb, ctx->stream));
        }
        return STARK_OK;
    }
    // scratch-bounded chunks of columns
    size_t chunk = ((size_t)1 << 26) / n;
    if (chunk == 0) chunk = 1;
    if (chunk > n_cols) chunk = n_cols;
    if (chunk > 65535) chunk = 65535;
    ST_TRY(ensure_scratch(ctx, chunk * n));
    const size_t P = t->radices.size();
    for (size_t c0 = 0; c0 < n_cols; c0 += chunk) {
        const size_t nb = n_cols - c0 < chunk ? n_cols - c0 : chunk;
        uint32_t log_s = log_n;
        for (size_t q = 0; q < P; q++) {
            log_s -= t->radices[q];
            PassArgs a{};
            a.inverse = inverse ? 1 : 0;
            a.tw256 = (q == P - 1) ? t->tw256 : tf->tw256;    // the inverse's n^-1 rides on its last pass's exchange
            a.hi = tf->hi;
            a.lo = tf->lo;
            a.log_n = log_n;
            a.log_s = log_s;
            a.log_s_next = (q + 1 < P) ? log_s - t->radices[q + 1] : 0;
            a.next_is_last = (q + 2 == P) ? 1 : 0;
            a.r = t->radices[q];
            a.h = t->h;
            a.nprev = (uint32_t)q;
            for (size_t u = 0; u < q; u++) a.rprev[u] = t->radices[u];
            a.n_src = n;
            PassMode mode;
            if (q == 0) {
                mode = PASS_FIRST;
                a.src = src + c0 * src_stride;
                a.src_batch_stride = src_stride;
                a.n_src = n_src;
                a.twist_hi = twist ? (const tw5*)twist->hi : nullptr;
                a.twist_lo = twist ? (const tw5*)twist->lo : nullptr;
            } else {
                mode = (q == P - 1) ? PASS_LAST : PASS_MIDDLE;
                a.src = ctx->scratch;
                a.src_batch_stride = n;
            }
            if (P == 1) mode = PASS_LAST;                     // rows in, natural order out: the one pass is a last pass
            if (q == P - 1) {
                a.dst = dst + c0 * dst_stride;
                a.dst_batch_stride = dst_stride;
            } else {
                a.dst = ctx->scratch;
                a.dst_batch_stride = n;
            }
            if (a.next_is_last && log_n <= PLANE_MAX_LOG) {
                // the twiddle plane of this pass: 20 bytes per element, built once per size
                NttTables* tm = const_cast<NttTables*>(tf);
                if (ctx->plane16_min_log && log_n >= ctx->plane16_min_log) {
                    if (!tm->plane16 && !tm->plane_refused) {
                        if (hipMalloc((void**)&tm->plane16, (size_t)16 << log_n) != hipSuccess) {
                            (void)hipGetLastError();
                            tm->plane16 = nullptr;
                            tm->plane_refused = true;
                        } else if (launch_ntt_plane16(a, tm->plane16, ctx->stream) != hipSuccess) {
                            (void)hipGetLastError();
                            (void)hipFree(tm->plane16);
                            tm->plane16 = nullptr;
                            tm->plane_refused = true;
                        }
                    }
                    a.plane16 = tm->plane16;
                } else if (!tm->plane && !tm->plane_refused) {
                    if (hipMalloc((void**)&tm->plane, (size_t)20 << log_n) != hipSuccess) {
                        (void)hipGetLastError();          // no room: this size keeps forming hi*lo per element
                        tm->plane = nullptr;
                        tm->plane_refused = true;
                    } else if (launch_ntt_plane(a, tm->plane, ctx->stream) != hipSuccess) {
                        (void)hipGetLastError();          // an unfilled plane must never be read
                        (void)hipFree(tm->plane);
                        tm->plane = nullptr;
                        tm->plane_refused = true;
                    }
                }
                a.plane = tm->plane;
            }
            HIP_TRY(launch_ntt_pass(a, mode, q == 0 && twist != nullptr, (uint32_t)nb, ctx->stream));
        }
    }
    return STARK_OK;
}

static bool is_one(fe x) { return x.lo == 1 && x.hi == 0; }
static bool is_canonical(fe x) { return !ge_p(x.lo, x.hi); }

// RAII staging buffer for the host-pointer entry points
// ---------------------------------------------------------------------------
// Host <-> device transfers of pageable memory.  The runtime stages such copies
// through pinned memory on one thread (about 20 GB/s here); for large vectors
// several lanes do the same staging side by side, each on its own thread and
// stream, which brings the rate close to what the link gives for pinned memory.
// ---------------------------------------------------------------------------
// A host-pointer transform whose data fits this many bytes (up to 4096 elements) runs on host-mapped
// pinned memory directly: the kernels read and write it over the link, so a call is launch +
// synchronisation instead of copy + launch + copy.
constexpr size_t SMALL_IO_BYTES = (size_t)64 << 10;
constexpr size_t LANE_CHUNK = (size_t)4 << 20;
constexpr size_t LANE_COUNT = 4;
constexpr size_t LANE_THRESHOLD = (size_t)8 << 20;     // smaller transfers go through the runtime

static hipError_t lanes_init(stark_ctx* ctx)
{
    if (!ctx->lanes.empty()) return hipSuccess;
    std::vector<stark_ctx::Lane> lanes(LANE_COUNT);
    hipError_t e = hipEventCreateWithFlags(&ctx->lane_gate, hipEventDisableTiming);
    for (auto& l : lanes) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
        for (int b = 0; b < 2 && e == hipSuccess; b++) {
            e = hipHostMalloc(&l.pin[b], LANE_CHUNK, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev[b], hipEventDisableTiming);
        }
    }
    if (e != hipSuccess) return e;          // partially built lanes are released with the context's process
    ctx->lanes = std::move(lanes);
    return hipSuccess;
}

static void lanes_destroy(stark_ctx* ctx)
{
    for (auto& l : ctx->lanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        for (int b = 0; b < 2; b++) {
            if (l.pin[b]) (void)hipHostFree(l.pin[b]);
            if (l.ev[b]) (void)hipEventDestroy(l.ev[b]);
        }
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    ctx->lanes.clear();
    if (ctx->lane_gate) (void)hipEventDestroy(ctx->lane_gate);
    ctx->lane_gate = nullptr;
}

// dst_dev <- src_host, ordered after earlier work on the context's stream; later work on that
// stream is ordered after the copy.  The host buffer is consumed before the call returns.
static hipError_t copy_in(stark_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
    if (bytes < LANE_THRESHOLD)
        return hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    hipError_t e = lanes_init(ctx);
    if (e == hipSuccess) e = hipEventRecord(ctx->lane_gate, ctx->stream);
    if (e != hipSuccess) return e;
    const size_t chunks = (bytes + LANE_CHUNK - 1) / LANE_CHUNK;
    std::atomic<int> err{(int)hipSuccess};
    auto work = [&](size_t t) {
        stark_ctx::Lane& l = ctx->lanes[t];
        hipError_t r = hipSetDevice(ctx->device);
        if (r == hipSuccess) r = hipStreamWaitEvent(l.stream, ctx->lane_gate, 0);
        size_t turn = 0;
        for (size_t k = t; k < chunks && r == hipSuccess; k += LANE_COUNT, turn++) {
            const int b = (int)(turn & 1);
            const size_t off = k * LANE_CHUNK, len = bytes - off < LANE_CHUNK ? bytes - off : LANE_CHUNK;
            if (turn >= 2) r = hipEventSynchronize(l.ev[b]);                 // the chunk's previous DMA has read it
            if (r != hipSuccess) break;
            memcpy(l.pin[b], (const uint8_t*)src_host + off, len);
            r = hipMemcpyAsync((uint8_t*)dst_dev + off, l.pin[b], len, hipMemcpyHostToDevice, l.stream);
            if (r == hipSuccess) r = hipEventRecord(l.ev[b], l.stream);
        }
        if (r == hipSuccess) r = hipStreamSynchronize(l.stream);             // pinned chunks are reused by the next call
        if (r != hipSuccess) err.store((int)r);
    };
    std::vector<std::thread> threads;
    for (size_t t = 1; t < LANE_COUNT; t++) threads.emplace_back(work, t);
    work(0);
    for (auto& th : threads) th.join();
    return (hipError_t)err.load();
}

// dst_host <- src_dev, ordered after earlier work on the context's stream; complete on return.
static hipError_t copy_out(stark_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
{
    if (bytes < LANE_THRESHOLD) {
        hipError_t e = hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
        return e == hipSuccess ? hipStreamSynchronize(ctx->stream) : e;
    }
    hipError_t e = lanes_init(ctx);
    if (e == hipSuccess) e = hipEventRecord(ctx->lane_gate, ctx->stream);
    if (e != hipSuccess) return e;
    const size_t chunks = (bytes + LANE_CHUNK - 1) / LANE_CHUNK;
    std::atomic<int> err{(int)hipSuccess};
    auto work = [&](size_t t) {
        stark_ctx::Lane& l = ctx->lanes[t];
        hipError_t r = hipSetDevice(ctx->device);
        if (r == hipSuccess) r = hipStreamWaitEvent(l.stream, ctx->lane_gate, 0);
        auto issue = [&](size_t k, int b) {
            const size_t off = k * LANE_CHUNK, len = bytes - off < LANE_CHUNK ? bytes - off : LANE_CHUNK;
            hipError_t q = hipMemcpyAsync(l.pin[b], (const uint8_t*)src_dev + off, len, hipMemcpyDeviceToHost, l.stream);
            return q == hipSuccess ? hipEventRecord(l.ev[b], l.stream) : q;
        };
        if (r == hipSuccess && t < chunks) r = issue(t, 0);
        size_t turn = 0;
        for (size_t k = t; k < chunks && r == hipSuccess; k += LANE_COUNT, turn++) {
            const int b = (int)(turn & 1);
            if (k + LANE_COUNT < chunks) r = issue(k + LANE_COUNT, b ^ 1);   // next chunk flies while this one is copied out
            if (r == hipSuccess) r = hipEventSynchronize(l.ev[b]);
            if (r != hipSuccess) break;
            const size_t off = k * LANE_CHUNK, len = bytes - off < LANE_CHUNK ? bytes - off : LANE_CHUNK;
            memcpy((uint8_t*)dst_host + off, l.pin[b], len);
        }
        if (r != hipSuccess) err.store((int)r);
    };
    std::vector<std::thread> threads;
    for (size_t t = 1; t < LANE_COUNT; t++) threads.emplace_back(work, t);
    work(0);
    for (auto& th : threads) th.join();
    return (hipError_t)err.load();
}

// Takes the next slot of the context's ring for `bytes` <= SMALL_SLOT_BYTES of host data, queues their copy to
// the slot's device block on the context's stream and returns the slot; the caller launches the kernel that
// reads slot->dev and then calls small_slot_done.
static hipError_t small_slot_stage(stark_ctx* ctx, const void* src, size_t bytes, stark_ctx::SmallSlot** out)
{
    stark_ctx::SmallSlot& sl = ctx->small[ctx->small_next];
    ctx->small_next = (ctx->small_next + 1) % stark_ctx::SMALL_SLOTS;
    hipError_t e = hipSuccess;
    if (!sl.pin) {
        e = hipHostMalloc(&sl.pin, stark_ctx::SMALL_SLOT_BYTES, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&sl.dev, stark_ctx::SMALL_SLOT_BYTES);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    if (sl.pending) {
        e = hipEventSynchronize(sl.done);
        if (e != hipSuccess) return e;
        sl.pending = false;
    }
    memcpy(sl.pin, src, bytes);
    e = hipMemcpyAsync(sl.dev, sl.pin, bytes, hipMemcpyHostToDevice, ctx->stream);
    *out = &sl;
    return e;
}

static hipError_t small_slot_done(stark_ctx* ctx, stark_ctx::SmallSlot* sl)
{
    hipError_t e = hipEventRecord(sl->done, ctx->stream);
    sl->pending = e == hipSuccess;
    return e;
}

struct DevBuf {
    stark_ctx* ctx;
    void* p = nullptr;
    size_t bytes = 0;
    explicit DevBuf(stark_ctx* c) : ctx(c) {}
    int32_t alloc(size_t b)
    {
        bytes = b;
        return pool_alloc(ctx, b, &p);
    }
    ~DevBuf() { pool_free(ctx, p, bytes); }
};

static int32_t tree_build(stark_ctx* ctx, const fe* elems_dev, size_t n, uint8_t root[32], stark_tree** tree_out)
{
    if (log2_exact(n) < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    const size_t bytes = (2 * n - 1) * sizeof(digest);
    void* d = nullptr;
    ST_TRY(pool_alloc(ctx, bytes, &d));
    digest* tree = (digest*)d;
    digest* level = tree;
    size_t m = n;
    hipError_t e = launch_leaf_levels(elems_dev, n, nullptr, tree, (uint32_t)log2_exact(n), 1, ctx->stream, &level, &m);
    if (e == hipSuccess) e = launch_merkle_reduce(level, m, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned, tree + (2 * n - 2), 32, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        pool_free(ctx, d, bytes);
        return fail(STARK_EHIP, std::string("merkle commit: ") + hipGetErrorString(e));
    }
    memcpy(root, ctx->pinned, 32);
    if (tree_out) {
        *tree_out = new stark_tree{ctx, tree, n, true, (uint32_t)log2_exact(n)};
    } else {
        pool_free(ctx, d, bytes);
    }
    return STARK_OK;
}

static int32_t roots_batch(stark_ctx* ctx, const uint64_t* elems_dev, size_t n, size_t n_cols, size_t col_stride,
                           uint8_t* roots_out, bool out_is_device)
{
    if (!ctx || !elems_dev || !roots_out) return fail(STARK_EARG, "null argument");
    const int log_n = log2_exact(n);
    if (log_n < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    if (n_cols == 0) return STARK_OK;
    if (col_stride < n) return fail(STARK_ELENGTH, "column stride shorter than the column");
    // columns are committed in groups whose trees (2n-1 digests each, not retained) fit 4 GiB of scratch
    const size_t tree_digests = 2 * n - 1;
    size_t group = ((size_t)4 << 30) / (tree_digests * sizeof(digest));
    if (group == 0) group = 1;
    if (group > n_cols) group = n_cols;
    DevBuf scratch(ctx);
    ST_TRY(scratch.alloc(group * tree_digests * sizeof(digest)));
    digest* level0 = (digest*)scratch.p;
    for (size_t c0 = 0; c0 < n_cols; c0 += group) {
        const size_t g = n_cols - c0 < group ? n_cols - c0 : group;
        digest* roots_dev = nullptr;
        digest* level = level0;
        size_t m = g * n;
        HIP_TRY(launch_leaf_levels((const fe*)elems_dev + c0 * col_stride, col_stride, nullptr, level0, (uint32_t)log_n, g,
                                   ctx->stream, &level, &m));
        HIP_TRY(launch_merkle_reduce(level, m, ctx->stream, nullptr, g, &roots_dev));
        HIP_TRY(hipMemcpyAsync(roots_out + 32 * c0, roots_dev, 32 * g,
                               out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
        if (!out_is_device) HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return STARK_OK;
}

extern "C" int32_t stark_merkle_roots_fe_batch_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t n, size_t n_cols,
                                                   size_t col_stride, uint8_t* roots_out)
{
    CtxGuard guard_(ctx);
    return roots_batch(ctx, elems_dev, n, n_cols, col_stride, roots_out, false);
}

extern "C" int32_t stark_merkle_subtree_roots_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t chunk_len,
                                                  size_t n_chunks, uint8_t* roots_dev_out, stark_tree** forest_out)
{
    CtxGuard guard_(ctx);
    if (!forest_out) return roots_batch(ctx, elems_dev, chunk_len, n_chunks, chunk_len, roots_dev_out, true);
    if (!ctx || !elems_dev || !roots_dev_out) return fail(STARK_EARG, "null argument");
    const int log_c = log2_exact(chunk_len);
    if (log_c < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    const size_t total = chunk_len * n_chunks;
    if (log2_exact(total) < 0) return fail(STARK_ENOTPOW2, "a retained forest needs a power-of-two number of chunks");
    // all subtrees in one buffer laid out like one tree over all the leaves, cut off at the subtree roots
    const size_t bytes = (2 * total - 1) * sizeof(digest);
    void* d = nullptr;
    ST_TRY(pool_alloc(ctx, bytes, &d));
    digest* roots = nullptr;
    digest* level = (digest*)d;
    size_t m = total;
    hipError_t e = launch_leaf_levels((const fe*)elems_dev, chunk_len, nullptr, (digest*)d, (uint32_t)log_c, n_chunks,
                                      ctx->stream, &level, &m);
    if (e == hipSuccess) e = launch_merkle_reduce(level, m, ctx->stream, nullptr, n_chunks, &roots);
    if (e == hipSuccess)
        e = hipMemcpyAsync(roots_dev_out, roots, n_chunks * sizeof(digest), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) {
        pool_free(ctx, d, bytes);
        return fail(STARK_EHIP, std::string("subtree roots: ") + hipGetErrorString(e));
    }
    *forest_out = new stark_tree{ctx, (digest*)d, total, true, (uint32_t)log_c};
    return STARK_OK;
}

extern "C" int32_t stark_merkle_root_of_digests_dev(stark_ctx* ctx, const uint8_t* digests_dev, size_t m,
                                                    uint8_t root[32], stark_tree** tree_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || !digests_dev || !root) return fail(STARK_EARG, "null argument");
    if (log2_exact(m) < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    // the reduction writes every level behind the one it reads: work on a copy with room for them
    const size_t bytes = (2 * m - 1) * sizeof(digest);
    void* d = nullptr;
    ST_TRY(pool_alloc(ctx, bytes, &d));
    // the last launch writes the root straight into host-mapped pinned memory; one digest has no launch
    digest* root_mail = (digest*)((uint8_t*)ctx->pinned + 2048);
    hipError_t e = hipMemcpyAsync(d, digests_dev, m * sizeof(digest), hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess)
        e = m == 1 ? hipMemcpyAsync(root_mail, d, sizeof(digest), hipMemcpyDeviceToHost, ctx->stream)
                   : launch_merkle_reduce((digest*)d, m, ctx->stream, root_mail);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        pool_free(ctx, d, bytes);
        return fail(STARK_EHIP, std::string("root of digests: ") + hipGetErrorString(e));
    }
    memcpy(root, root_mail, 32);
    if (tree_out) *tree_out = new stark_tree{ctx, (digest*)d, m, true, (uint32_t)log2_exact(m)};
    else pool_free(ctx, d, bytes);
    return STARK_OK;
}

static int32_t tree_open(stark_ctx* ctx, const digest* tree, size_t n, size_t index, uint8_t* path_out,
                         size_t* depth_out, int max_levels = 64)
{
    int depth = log2_exact(n);
    if (depth < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    if (depth > max_levels) depth = max_levels;
    if (index >= n) return fail(STARK_EINDEX, "cannot open invalid index");
    if (depth_out) *depth_out = (size_t)depth;
    if (depth == 0) return STARK_OK;
    digest* stage = (digest*)((uint8_t*)ctx->pinned + 64);     // 4 KiB pinned buffer: depth <= 30 digests fit
    DevBuf tmp(ctx);
    ST_TRY(tmp.alloc((size_t)depth * sizeof(digest)));
    HIP_TRY(launch_merkle_open(tree, n, index, (digest*)tmp.p, ctx->stream, (uint32_t)depth));
    HIP_TRY(hipMemcpyAsync(stage, tmp.p, (size_t)depth * sizeof(digest), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(path_out, stage, (size_t)depth * sizeof(digest));
    return STARK_OK;
}

static int32_t tree_open_many(stark_ctx* ctx, const digest* tree, size_t n, int max_levels, const uint64_t* indices,
                              size_t count, uint8_t* paths_out, size_t* depth_out)
{
    int depth = log2_exact(n);
    if (depth < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    if (depth > max_levels) depth = max_levels;
    if (depth_out) *depth_out = (size_t)depth;
    if (count > 0xFFFFFFFFull) return fail(STARK_ELENGTH, "too many openings in one call");
    for (size_t i = 0; i < count; i++)
        if (indices[i] >= n) return fail(STARK_EINDEX, "cannot open invalid index");
    if (count == 0 || depth == 0) return STARK_OK;
    DevBuf idx(ctx), paths(ctx);
    const size_t path_bytes = count * (size_t)depth * sizeof(digest);
    ST_TRY(idx.alloc(count * sizeof(uint64_t)));
    ST_TRY(paths.alloc(path_bytes));
    HIP_TRY(hipMemcpyAsync(idx.p, indices, count * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_merkle_open_many(tree, n, (const uint64_t*)idx.p, count, (uint32_t)depth, (digest*)paths.p, ctx->stream));
    HIP_TRY(hipMemcpyAsync(paths_out, paths.p, path_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

}  // namespace

// ===========================================================================
// context
// ===========================================================================
extern "C" const char* stark_last_error(void) { return g_last_error.c_str(); }

extern "C" int32_t stark_ctx_create(int32_t device, stark_ctx** out)
{
    if (!out) return fail(STARK_EARG, "stark_ctx_create: null out");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(STARK_EARG, "stark_ctx_create: no such device");
    HIP_TRY(hipSetDevice(device));
    stark_ctx* ctx = new stark_ctx();
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(&ctx->pinned, 4096, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&ctx->pin_io, 2 * SMALL_IO_BYTES, hipHostMallocDefault);
    // each GPU holds its own copy of the static scratch array at its own address (the device was
    // made current above)
    if (e == hipSuccess) e = hipGetSymbolAddress(&ctx->static_scratch, HIP_SYMBOL(scratch));
    if (e != hipSuccess) {
        delete ctx;
        return fail(STARK_EHIP, std::string("stark_ctx_create: ") + hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    if (const char* e = std::getenv("STARK_NTT_PLANE16_FROM")) ctx->plane16_min_log = (uint32_t)std::atoi(e);
    *out = ctx;
    return STARK_OK;
}

extern "C" void stark_ctx_destroy(stark_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->ntt_tables) free_ntt_tables(kv.second);
    ctx->pow_tables.clear();
    pool_drop(ctx);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pin_io) (void)hipHostFree(ctx->pin_io);
    lanes_destroy(ctx);
    for (auto& sl : ctx->small) {
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

// Leaving a stream waits for what the context queued on it, except while either stream is being captured into a
// graph: a synchronisation from the capturing thread is not allowed then, and the stream left holds no work of its
// own that the capture could depend on if the caller bound and warmed up before capturing.
static int32_t switch_stream(stark_ctx* ctx, hipStream_t next)
{
    if (next == ctx->stream) return STARK_OK;
    hipStreamCaptureStatus from = hipStreamCaptureStatusNone, to = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &from);
    (void)hipStreamIsCapturing(next, &to);
    (void)hipGetLastError();
    if (from == hipStreamCaptureStatusNone && to == hipStreamCaptureStatusNone) HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = next;
    return STARK_OK;
}

extern "C" int32_t stark_ctx_set_stream(stark_ctx* ctx, void* hip_stream)
{
    CtxGuard guard_(ctx);
    if (!ctx) return fail(STARK_EARG, "null context");
    return switch_stream(ctx, (hipStream_t)hip_stream);
}

extern "C" int32_t stark_ctx_reset_stream(stark_ctx* ctx)
{
    CtxGuard guard_(ctx);
    if (!ctx) return fail(STARK_EARG, "null context");
    return switch_stream(ctx, ctx->own_stream);
}

extern "C" int32_t stark_ctx_synchronize(stark_ctx* ctx)
{
    CtxGuard guard_(ctx);
    if (!ctx) return fail(STARK_EARG, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_dev_alloc(stark_ctx* ctx, size_t bytes, void** dptr)
{
    CtxGuard guard_(ctx);
    if (!ctx || !dptr) return fail(STARK_EARG, "null argument");
    // through the block cache: a prover allocates the same few large vectors proof after proof, and a
    // hipMalloc / hipFree pair of that size costs more than the kernels that use it
    if (bytes == 0) bytes = 16;
    ST_TRY(pool_alloc(ctx, bytes, dptr));
    ctx->dev_blocks[*dptr] = bytes;
    return STARK_OK;
}

// Gives back what the context keeps between calls for speed: the cached scratch blocks, the ping
// buffer of the multi-pass transform and the twiddle planes (tables and planes are rebuilt on next use).
extern "C" int32_t stark_ctx_trim(stark_ctx* ctx)
{
    CtxGuard guard_(ctx);
    if (!ctx) return fail(STARK_EARG, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    pool_drop(ctx);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_elems = 0;
    for (auto& kv : ctx->ntt_tables) free_ntt_tables(kv.second);
    ctx->ntt_tables.clear();
    ctx->pow_tables.clear();
    return STARK_OK;
}

extern "C" int32_t stark_dev_free(stark_ctx* ctx, void* dptr)
{
    CtxGuard guard_(ctx);
    if (!ctx) return fail(STARK_EARG, "null context");
    if (!dptr) return STARK_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    auto it = ctx->dev_blocks.find(dptr);
    if (it == ctx->dev_blocks.end()) {
        HIP_TRY(hipFree(dptr));
        return STARK_OK;
    }
    pool_free(ctx, dptr, it->second);
    ctx->dev_blocks.erase(it);
    return STARK_OK;
}

extern "C" int32_t stark_copy_h2d(stark_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
    CtxGuard guard_(ctx);
    if (!ctx) return fail(STARK_EARG, "null context");
    HIP_TRY(copy_in(ctx, dst_dev, src_host, bytes));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_copy_d2h(stark_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
{
    CtxGuard guard_(ctx);
    if (!ctx) return fail(STARK_EARG, "null context");
    HIP_TRY(copy_out(ctx, dst_host, src_dev, bytes));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_copy_d2d(stark_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes)
{
    CtxGuard guard_(ctx);
    if (!ctx || (bytes && (!dst_dev || !src_dev))) return fail(STARK_EARG, "null argument");
    if (bytes) HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return STARK_OK;
}

// ===========================================================================
// NTT
// ===========================================================================
static int32_t check_ntt_size(size_t n, int* log_n)
{
    int l = log2_exact(n);
    if (l < 0) return fail(STARK_ENOTPOW2, "NTT size must be power of 2");
    if (l > 30) return fail(STARK_ETOOLARGE, "NTT size too large for practical use");
    *log_n = l;
    return STARK_OK;
}

extern "C" int32_t stark_ntt_batch_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n, size_t n_cols,
                                       size_t col_stride, int32_t inverse)
{
    CtxGuard guard_(ctx);
    if (!ctx || (!data_dev && n_cols)) return fail(STARK_EARG, "null argument");
    int log_n;
    ST_TRY(check_ntt_size(n, &log_n));
    if (n_cols > 1 && col_stride < n) return fail(STARK_ELENGTH, "column stride smaller than NTT size");
    fe* d = (fe*)data_dev;
    return run_ntt(ctx, d, col_stride, n, d, col_stride, (uint32_t)log_n, inverse != 0, nullptr, n_cols);
}

extern "C" int32_t stark_ntt_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n, int32_t inverse)
{
    CtxGuard guard_(ctx);
    return stark_ntt_batch_dev(ctx, data_dev, n, 1, n, inverse);
}

extern "C" int32_t stark_ntt_batch(stark_ctx* ctx, uint64_t* data, size_t n, size_t n_cols, size_t col_stride,
                                   int32_t inverse)
{
    CtxGuard guard_(ctx);
    if (!ctx || (!data && n_cols)) return fail(STARK_EARG, "null argument");
    int log_n;
    ST_TRY(check_ntt_size(n, &log_n));
    if (n_cols == 0) return STARK_OK;
    if (n_cols > 1 && col_stride < n) return fail(STARK_ELENGTH, "column stride smaller than NTT size");
    const size_t span = (n_cols - 1) * col_stride + n;
    if (span * sizeof(fe) <= SMALL_IO_BYTES) {
        memcpy(ctx->pin_io, data, span * sizeof(fe));
        ST_TRY(stark_ntt_batch_dev(ctx, (uint64_t*)ctx->pin_io, n, n_cols, col_stride, inverse));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        memcpy(data, ctx->pin_io, span * sizeof(fe));
        return STARK_OK;
    }
    DevBuf buf(ctx);
    ST_TRY(buf.alloc(span * sizeof(fe)));
    HIP_TRY(copy_in(ctx, buf.p, data, span * sizeof(fe)));
    ST_TRY(stark_ntt_batch_dev(ctx, (uint64_t*)buf.p, n, n_cols, col_stride, inverse));
    HIP_TRY(copy_out(ctx, data, buf.p, span * sizeof(fe)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_ntt(stark_ctx* ctx, uint64_t* data, size_t n, int32_t inverse)
{
    CtxGuard guard_(ctx);
    return stark_ntt_batch(ctx, data, n, 1, n, inverse);
}

extern "C" int32_t stark_coset_lde_batch_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n_coeffs,
                                             size_t coeff_stride, size_t n_cols, size_t n_out,
                                             const uint64_t offset[2], uint64_t* out_dev, size_t out_stride)
{
    CtxGuard guard_(ctx);
    if (!ctx || !offset || (n_cols && (!coeffs_dev || !out_dev))) return fail(STARK_EARG, "null argument");
    int log_n;
    ST_TRY(check_ntt_size(n_out, &log_n));
    if (n_coeffs > n_out) return fail(STARK_ELENGTH, "more coefficients than evaluation points");
    if (n_cols > 1 && (coeff_stride < n_coeffs || out_stride < n_out))
        return fail(STARK_ELENGTH, "stride smaller than column length");
    fe off{offset[0], offset[1]};
    if (!is_canonical(off)) return fail(STARK_EARG, "offset not reduced modulo p");
    const fe* twist = (!is_one(off) && n_coeffs > 0) ? &off : nullptr;
    return run_ntt(ctx, (const fe*)coeffs_dev, coeff_stride, n_coeffs, (fe*)out_dev, out_stride, (uint32_t)log_n,
                   false, twist, n_cols);
}

extern "C" int32_t stark_coset_lde_batch(stark_ctx* ctx, const uint64_t* coeffs, size_t n_coeffs,
                                         size_t coeff_stride, size_t n_cols, size_t n_out,
                                         const uint64_t offset[2], uint64_t* out, size_t out_stride)
{
    CtxGuard guard_(ctx);
    if (!ctx || !offset || (n_cols && (!coeffs || !out))) return fail(STARK_EARG, "null argument");
    int log_n;
    ST_TRY(check_ntt_size(n_out, &log_n));
    if (n_coeffs > n_out) return fail(STARK_ELENGTH, "more coefficients than evaluation points");
    if (n_cols == 0) return STARK_OK;
    if (n_cols > 1 && (coeff_stride < n_coeffs || out_stride < n_out))
        return fail(STARK_ELENGTH, "stride smaller than column length");
    const size_t in_span = (n_cols - 1) * coeff_stride + n_coeffs;
    const size_t out_span = (n_cols - 1) * out_stride + n_out;
    if (in_span * sizeof(fe) <= SMALL_IO_BYTES && out_span * sizeof(fe) <= SMALL_IO_BYTES &&
        (out_stride == n_out || n_cols == 1)) {
        uint8_t* pin_in = (uint8_t*)ctx->pin_io;
        uint8_t* pin_out = pin_in + SMALL_IO_BYTES;
        if (in_span) memcpy(pin_in, coeffs, in_span * sizeof(fe));
        ST_TRY(stark_coset_lde_batch_dev(ctx, (const uint64_t*)pin_in, n_coeffs, coeff_stride, n_cols, n_out, offset,
                                         (uint64_t*)pin_out, out_stride));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        memcpy(out, pin_out, out_span * sizeof(fe));
        return STARK_OK;
    }
    DevBuf in(ctx), ob(ctx);
    ST_TRY(in.alloc(in_span * sizeof(fe)));
    ST_TRY(ob.alloc(out_span * sizeof(fe)));
    if (in_span) HIP_TRY(copy_in(ctx, in.p, coeffs, in_span * sizeof(fe)));
    ST_TRY(stark_coset_lde_batch_dev(ctx, (const uint64_t*)in.p, n_coeffs, coeff_stride, n_cols, n_out, offset,
                                     (uint64_t*)ob.p, out_stride));
    // only the columns are defined; copy them one by one when padded
    if (out_stride == n_out || n_cols == 1) {
        HIP_TRY(copy_out(ctx, out, ob.p, out_span * sizeof(fe)));
    } else {
        for (size_t c = 0; c < n_cols; c++)
            HIP_TRY(copy_out(ctx, out + 2 * c * out_stride, (fe*)ob.p + c * out_stride, n_out * sizeof(fe)));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_vec_mul_dev(stark_ctx* ctx, const uint64_t* a_dev, const uint64_t* b_dev,
                                     uint64_t* out_dev, size_t n)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n && (!a_dev || !b_dev || !out_dev))) return fail(STARK_EARG, "null argument");
    HIP_TRY(launch_vec_mul((const fe*)a_dev, (const fe*)b_dev, (fe*)out_dev, n, ctx->F.r2, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_vec_mul(stark_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n && (!a || !b || !out))) return fail(STARK_EARG, "null argument");
    if (n == 0) return STARK_OK;
    DevBuf da(ctx), db(ctx);
    ST_TRY(da.alloc(n * sizeof(fe)));
    ST_TRY(db.alloc(n * sizeof(fe)));
    HIP_TRY(copy_in(ctx, da.p, a, n * sizeof(fe)));
    HIP_TRY(copy_in(ctx, db.p, b, n * sizeof(fe)));
    ST_TRY(stark_vec_mul_dev(ctx, (uint64_t*)da.p, (uint64_t*)db.p, (uint64_t*)da.p, n));
    HIP_TRY(copy_out(ctx, out, da.p, n * sizeof(fe)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

// ===========================================================================
// pointwise polynomial work around the transforms
// ===========================================================================
extern "C" int32_t stark_poly_compose_linear_dev(stark_ctx* ctx, uint64_t* coeffs_dev, size_t n, size_t n_cols,
                                                 size_t col_stride, const uint64_t s[2])
{
    CtxGuard guard_(ctx);
    if (!ctx || !s || (n && n_cols && !coeffs_dev)) return fail(STARK_EARG, "null argument");
    if (n == 0 || n_cols == 0) return STARK_OK;
    if (n_cols > 1 && col_stride < n) return fail(STARK_ELENGTH, "column stride shorter than the column");
    if (n_cols > 65535) return fail(STARK_ELENGTH, "too many columns in one call");
    fe x{s[0], s[1]};
    if (!is_canonical(x)) return fail(STARK_EARG, "scalar not reduced modulo p");
    if (is_one(x)) return STARK_OK;
    PowRef pt;
    ST_TRY(get_pow_tables(ctx, x, n, false, &pt));
    HIP_TRY(launch_compose_linear((fe*)coeffs_dev, n, n_cols, col_stride, (const fe*)pt->hi, (const fe*)pt->lo, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_coset_interpolate_batch_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n, size_t n_cols,
                                                     size_t col_stride, const uint64_t offset[2])
{
    CtxGuard guard_(ctx);
    if (!ctx || !offset || (n_cols && !data_dev)) return fail(STARK_EARG, "null argument");
    fe off{offset[0], offset[1]};
    if (!is_canonical(off) || fe_is_zero(off)) return fail(STARK_EARG, "offset must be a nonzero field element");
    ST_TRY(stark_ntt_batch_dev(ctx, data_dev, n, n_cols, col_stride, 1));
    fe inv = ctx->F.inverse(off);
    const uint64_t inv_w[2] = {inv.lo, inv.hi};
    return stark_poly_compose_linear_dev(ctx, data_dev, n, n_cols, col_stride, inv_w);
}

extern "C" int32_t stark_vec_inv_dev(stark_ctx* ctx, const uint64_t* a_dev, uint64_t* out_dev, size_t n)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n && (!a_dev || !out_dev))) return fail(STARK_EARG, "null argument");
    HIP_TRY(launch_vec_inv((const fe*)a_dev, (fe*)out_dev, n, ctx->F.r2, ctx->F.one_m, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_poly_lincomb_dev(stark_ctx* ctx, size_t n_terms, const uint64_t* const* srcs_dev,
                                          const size_t* lens, const size_t* shifts, const uint64_t* weights,
                                          uint64_t* out_dev, size_t n_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n_terms && (!srcs_dev || !lens || !shifts || !weights)) || (n_out && !out_dev))
        return fail(STARK_EARG, "null argument");
    if (n_terms > LINCOMB_MAX_TERMS) return fail(STARK_ELENGTH, "at most 16 terms in one call");
    LincombArgs a{};
    for (size_t k = 0; k < n_terms; k++) {
        if (lens[k] && !srcs_dev[k]) return fail(STARK_EARG, "null term");
        if (lens[k] + shifts[k] > n_out) return fail(STARK_ELENGTH, "term does not fit the output length");
        fe w{weights[2 * k], weights[2 * k + 1]};
        if (!is_canonical(w)) return fail(STARK_EARG, "scalar not reduced modulo p");
        a.src[k] = (const fe*)srcs_dev[k];
        a.len[k] = lens[k];
        a.shift[k] = shifts[k];
        a.weight_m[k] = ctx->F.to_mont(w);
    }
    a.out = (fe*)out_dev;
    a.n_out = n_out;
    a.n_terms = (uint32_t)n_terms;
    HIP_TRY(launch_lincomb(a, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_mpoly_eval_dev(stark_ctx* ctx, size_t n_vars, const uint64_t* const* cols_dev,
                                        const size_t* rotations, size_t n_terms, const uint64_t* coeffs,
                                        const uint8_t* exps, const uint64_t* mul_dev, uint64_t* out_dev, size_t n)
{
    CtxGuard guard_(ctx);
    if (!ctx || !cols_dev || (n_terms && (!coeffs || !exps)) || (n && !out_dev)) return fail(STARK_EARG, "null argument");
    if (n_vars == 0 || n_vars > MPOLY_MAX_VARS) return fail(STARK_ELENGTH, "between 1 and 16 variables");
    if (n == 0) return STARK_OK;
    MPolyArgs a{};
    for (size_t v = 0; v < n_vars; v++) {
        if (!cols_dev[v]) return fail(STARK_EARG, "null column");
        a.cols[v] = (const fe*)cols_dev[v];
        a.rot[v] = rotations ? rotations[v] % n : 0;
    }
    // term table: five-limb forms of the coefficients, then 16 exponent bytes per term
    const size_t coef_bytes = (n_terms ? n_terms : 1) * sizeof(tw5);
    std::vector<uint8_t> table(coef_bytes + (n_terms ? n_terms : 1) * MPOLY_MAX_VARS, 0);
    for (size_t k = 0; k < n_terms; k++) {
        fe c{coeffs[2 * k], coeffs[2 * k + 1]};
        if (!is_canonical(c)) return fail(STARK_EARG, "scalar not reduced modulo p");
        tw5 cm = to_tw5(ctx->F, c);
        memcpy(table.data() + k * sizeof(tw5), &cm, sizeof(tw5));
        for (size_t v = 0; v < n_vars; v++) table[coef_bytes + k * MPOLY_MAX_VARS + v] = exps[k * n_vars + v];
    }
    // small tables (the usual case: a few dozen terms) travel through the context's ring of staging slots and the
    // call returns with everything queued; a table too large for a slot goes up from pageable memory and the
    // stream is drained before that memory and its device block go away
    DevBuf tb(ctx);
    stark_ctx::SmallSlot* slot = nullptr;
    const uint8_t* table_dev;
    if (table.size() <= stark_ctx::SMALL_SLOT_BYTES) {
        HIP_TRY(small_slot_stage(ctx, table.data(), table.size(), &slot));
        table_dev = (const uint8_t*)slot->dev;
    } else {
        ST_TRY(tb.alloc(table.size()));
        HIP_TRY(hipMemcpyAsync(tb.p, table.data(), table.size(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        table_dev = (const uint8_t*)tb.p;
    }
    a.coef5 = (const tw5*)table_dev;
    a.exps = table_dev + coef_bytes;
    a.mul = (const fe*)mul_dev;
    a.out = (fe*)out_dev;
    a.n = n;
    a.n_vars = (uint32_t)n_vars;
    a.n_terms = (uint32_t)n_terms;
    a.scaled_one = to_tw5(ctx->F, fe{1, 0});
    {
        fe r156{1, 0};
        for (int b = 0; b < 156; b++) r156 = fe_add(r156, r156);
        a.scale_in = to_tw5(ctx->F, r156);
    }
    HIP_TRY(launch_mpoly_eval(a, ctx->stream));
    if (slot)
        HIP_TRY(small_slot_done(ctx, slot));
    else
        HIP_TRY(hipStreamSynchronize(ctx->stream));          // the table block returns to the cache on return
    return STARK_OK;
}

// stark_poly_degree_dev without the wait: 0 for the zero polynomial, else degree + 1, left in device memory
extern "C" int32_t stark_poly_degree_post_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n, uint64_t* result_dev)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n && !coeffs_dev) || !result_dev) return fail(STARK_EARG, "null argument");
    HIP_TRY(hipMemsetAsync(result_dev, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(launch_poly_degree((const fe*)coeffs_dev, n, (unsigned long long*)result_dev, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_poly_degree_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n, size_t* degree_out,
                                         int32_t* is_zero_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n && !coeffs_dev) || !degree_out) return fail(STARK_EARG, "null argument");
    // the kernel's atomics go to a word of device memory (one per wave that saw a nonzero coefficient; against
    // host memory each would cross the link), which is then copied out
    unsigned long long* mail = (unsigned long long*)((uint8_t*)ctx->pinned + 1024);
    DevBuf word(ctx);
    ST_TRY(word.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(word.p, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(launch_poly_degree((const fe*)coeffs_dev, n, (unsigned long long*)word.p, ctx->stream));
    HIP_TRY(hipMemcpyAsync(mail, word.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const unsigned long long r = *mail;
    *degree_out = r ? (size_t)(r - 1) : 0;                   // univariate_poly.rs:69-85: the zero polynomial has degree 0
    if (is_zero_out) *is_zero_out = r == 0;
    return STARK_OK;
}

extern "C" int32_t stark_poly_eval_points_dev(stark_ctx* ctx, const uint64_t* coeffs_dev, size_t n, size_t n_polys,
                                              size_t stride, const uint64_t* points, size_t n_points, uint64_t* out)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n && n_polys && !coeffs_dev) || (n_points && (!points || !out))) return fail(STARK_EARG, "null argument");
    if (n_points > EVAL_MAX_POINTS) return fail(STARK_ELENGTH, "at most 16 points in one call");
    if (n_polys > 65535) return fail(STARK_ELENGTH, "too many polynomials in one call");
    if (n_polys > 1 && stride < n) return fail(STARK_ELENGTH, "stride shorter than the polynomial");
    if (n_points == 0 || n_polys == 0) return STARK_OK;
    const HostField& F = ctx->F;
    if (n == 0) {                                             // the zero polynomial (univariate_poly.rs:87-96 returns zero)
        memset(out, 0, n_polys * n_points * sizeof(fe));
        return STARK_OK;
    }
    EvalArgs a{};
    a.coeffs = (const fe*)coeffs_dev;
    a.n = n;
    a.stride = stride;
    a.n_points = (uint32_t)n_points;
    const size_t blocks = (n + EVAL_BLOCK_COEFFS - 1) / EVAL_BLOCK_COEFFS;
    if (blocks > 0x7fffffffull) return fail(STARK_ETOOLARGE, "polynomial too long");
    a.n_blocks = (uint32_t)blocks;
    std::vector<fe> x4096(n_points);
    for (size_t p = 0; p < n_points; p++) {
        fe x{points[2 * p], points[2 * p + 1]};
        if (!is_canonical(x)) return fail(STARK_EARG, "point not reduced modulo p");
        a.x_m[p] = F.to_mont(x);
        fe pw = F.pow(x, 16);
        for (int k = 0; k < 8; k++) {
            a.x_pow_m[p][k] = F.to_mont(pw);
            pw = F.mul(pw, pw);
        }
        x4096[p] = pw;                                        // x^(16 * 2^8)
    }
    const size_t n_part = n_polys * blocks * n_points;
    DevBuf part(ctx);
    ST_TRY(part.alloc(n_part * sizeof(fe)));
    a.partial = (fe*)part.p;
    HIP_TRY(launch_poly_eval_points(a, (uint32_t)n_polys, ctx->stream));
    std::vector<fe> host(n_part);
    HIP_TRY(hipMemcpyAsync(host.data(), part.p, n_part * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // the blocks are the coefficients of a polynomial in x^4096: Horner on the host
    for (size_t q = 0; q < n_polys; q++)
        for (size_t p = 0; p < n_points; p++) {
            const fe ym = F.to_mont(x4096[p]);
            fe acc = fe_zero();
            for (size_t b = blocks; b-- > 0;) acc = fe_add(mont_mul(acc, ym), host[(q * blocks + b) * n_points + p]);
            out[2 * (q * n_points + p)] = acc.lo;
            out[2 * (q * n_points + p) + 1] = acc.hi;
        }
    return STARK_OK;
}

extern "C" int32_t stark_sample_bytes_dev(stark_ctx* ctx, const uint8_t* bytes_dev, size_t width, size_t n,
                                          int32_t reversed, uint64_t* out_dev)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n && (!bytes_dev || !out_dev))) return fail(STARK_EARG, "null argument");
    if (width == 0 || width > 0xFFFFFFFFull) return fail(STARK_ELENGTH, "byte strings must be 1 .. 2^32-1 bytes long");
    HIP_TRY(launch_sample_bytes(bytes_dev, (uint32_t)width, (fe*)out_dev, n, ctx->F.to_mont(fe{0, (u64)1 << 56}),
                                reversed != 0, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_fourstep_twiddle_dev(stark_ctx* ctx, uint64_t* data_dev, size_t n_total, size_t row_start,
                                              size_t n_rows, size_t n_cols, int32_t inverse)
{
    CtxGuard guard_(ctx);
    if (!ctx || (!data_dev && n_rows * n_cols)) return fail(STARK_EARG, "null argument");
    int log_n;
    ST_TRY(check_ntt_size(n_total, &log_n));
    if (n_rows == 0 || n_cols == 0) return STARK_OK;
    // the largest exponent (row_start + n_rows - 1) * (n_cols - 1) must stay below n_total
    if ((row_start + n_rows - 1) * (n_cols - 1) >= n_total && n_cols > 1)
        return fail(STARK_ELENGTH, "twiddle block exceeds the transform size");
    fe w = ctx->F.primitive_nth_root((unsigned)log_n);
    if (inverse) w = ctx->F.inverse(w);
    PowRef pt;
    ST_TRY(get_pow_tables(ctx, w, n_total, false, &pt));
    HIP_TRY(launch_fourstep_twiddle((fe*)data_dev, row_start, n_rows, n_cols, (const fe*)pt->hi, (const fe*)pt->lo,
                                    ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_fourstep_pack_dev(stark_ctx* ctx, const uint64_t* src_dev, size_t n_rows, size_t n_cols,
                                           size_t world, size_t n_total, size_t row_start, int32_t twiddle,
                                           uint64_t* dst_dev)
{
    CtxGuard guard_(ctx);
    if (!ctx || (n_rows * n_cols && (!src_dev || !dst_dev))) return fail(STARK_EARG, "null argument");
    if (world == 0 || n_cols % world) return fail(STARK_ELENGTH, "the columns do not divide among the devices");
    if (src_dev == dst_dev && n_rows * n_cols) return fail(STARK_EARG, "pack cannot run in place");
    if (n_rows == 0 || n_cols == 0) return STARK_OK;
    PowRef pt;
    if (twiddle) {
        int log_n;
        ST_TRY(check_ntt_size(n_total, &log_n));
        if (n_cols > 1 && (row_start + n_rows - 1) * (n_cols - 1) >= n_total)
            return fail(STARK_ELENGTH, "twiddle block exceeds the transform size");
        fe w = ctx->F.primitive_nth_root((unsigned)log_n);
        if (twiddle < 0) w = ctx->F.inverse(w);
        ST_TRY(get_pow_tables(ctx, w, n_total, false, &pt));
    }
    HIP_TRY(launch_fourstep_pack((const fe*)src_dev, (fe*)dst_dev, n_rows, n_cols, world, row_start,
                                 pt ? (const fe*)pt->hi : nullptr, pt ? (const fe*)pt->lo : nullptr, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_fourstep_unpack_dev(stark_ctx* ctx, const uint64_t* src_dev, size_t world, size_t n_rows,
                                             size_t run, uint64_t* dst_dev)
{
    CtxGuard guard_(ctx);
    const size_t total = world * n_rows * run;
    if (!ctx || (total && (!src_dev || !dst_dev))) return fail(STARK_EARG, "null argument");
    if (src_dev == dst_dev && total) return fail(STARK_EARG, "unpack cannot run in place");
    HIP_TRY(launch_fourstep_unpack((const fe*)src_dev, (fe*)dst_dev, world, n_rows, run, ctx->stream));
    return STARK_OK;
}

// ===========================================================================
// Merkle
// ===========================================================================
extern "C" int32_t stark_merkle_commit_fe_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t n,
                                              uint8_t root[32], stark_tree** tree_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || !elems_dev || !root) return fail(STARK_EARG, "null argument");
    return tree_build(ctx, (const fe*)elems_dev, n, root, tree_out);
}

// stark_merkle_commit_fe_dev without a retained tree and without any allocation: the tree is built
// in the static device array and only the root leaves it.  Nothing is synchronised.
extern "C" int32_t stark_merkle_root_fe_static_dev(stark_ctx* ctx, const uint64_t* elems_dev, size_t n,
                                                   uint8_t* root_dev_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || !elems_dev || !root_dev_out) return fail(STARK_EARG, "null argument");
    if (log2_exact(n) < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    if (n > STATIC_TREE_MAX_LEAVES) return fail(STARK_ETOOLARGE, "more than 262144 leaves do not fit the static scratch");
    if (!ctx->static_scratch) return fail(STARK_EHIP, "static scratch address unknown");
    if (n == 1) {                                  // the single leaf digest is the root
        HIP_TRY(launch_leaf_hash((const fe*)elems_dev, (digest*)root_dev_out, 1, ctx->stream));
        return STARK_OK;
    }
    digest* tree = (digest*)ctx->static_scratch;   // every level is written in full before it is read: nothing to clear
    digest* level = tree;
    size_t m = n;
    HIP_TRY(launch_leaf_levels((const fe*)elems_dev, n, nullptr, tree, (uint32_t)log2_exact(n), 1, ctx->stream, &level, &m));
    HIP_TRY(launch_merkle_reduce(level, m, ctx->stream, (digest*)root_dev_out));
    return STARK_OK;
}

extern "C" int32_t stark_merkle_commit_fe(stark_ctx* ctx, const uint64_t* elems, size_t n, uint8_t root[32],
                                          stark_tree** tree_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || !elems || !root) return fail(STARK_EARG, "null argument");
    if (log2_exact(n) < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    DevBuf buf(ctx);
    ST_TRY(buf.alloc(n * sizeof(fe)));
    HIP_TRY(copy_in(ctx, buf.p, elems, n * sizeof(fe)));
    return tree_build(ctx, (const fe*)buf.p, n, root, tree_out);
}

extern "C" int32_t stark_merkle_commit_bytes(stark_ctx* ctx, const uint8_t* flat, const uint64_t* offsets,
                                             size_t n, uint8_t root[32], stark_tree** tree_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || !offsets || !root) return fail(STARK_EARG, "null argument");
    if (log2_exact(n) < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    const size_t total = (size_t)offsets[n];
    if (total && !flat) return fail(STARK_EARG, "null leaf data");
    DevBuf dflat(ctx), doff(ctx);
    ST_TRY(dflat.alloc(total ? total : 16));
    ST_TRY(doff.alloc((n + 1) * sizeof(uint64_t)));
    if (total) HIP_TRY(copy_in(ctx, dflat.p, flat, total));
    HIP_TRY(hipMemcpyAsync(doff.p, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    const size_t bytes = (2 * n - 1) * sizeof(digest);
    void* d = nullptr;
    ST_TRY(pool_alloc(ctx, bytes, &d));
    digest* tree = (digest*)d;
    hipError_t e = launch_leaf_hash_bytes((const uint8_t*)dflat.p, (const uint64_t*)doff.p, tree, n, ctx->stream);
    if (e == hipSuccess) e = launch_merkle_reduce(tree, n, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned, tree + (2 * n - 2), 32, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        pool_free(ctx, d, bytes);
        return fail(STARK_EHIP, std::string("merkle commit: ") + hipGetErrorString(e));
    }
    memcpy(root, ctx->pinned, 32);
    if (tree_out) *tree_out = new stark_tree{ctx, tree, n, true, (uint32_t)log2_exact(n)};
    else pool_free(ctx, d, bytes);
    return STARK_OK;
}

extern "C" int32_t stark_merkle_open(stark_tree* tree, size_t index, uint8_t* path_out, size_t* depth_out)
{
    CtxGuard guard_(tree ? tree->ctx : nullptr);
    if (!tree || !path_out) return fail(STARK_EARG, "null argument");
    return tree_open(tree->ctx, tree->d, tree->n, index, path_out, depth_out, (int)tree->levels);
}

extern "C" int32_t stark_merkle_open_many(stark_tree* tree, const uint64_t* indices, size_t count, uint8_t* paths_out,
                                          size_t* depth_out)
{
    CtxGuard guard_(tree ? tree->ctx : nullptr);
    if (!tree || (count && (!indices || !paths_out))) return fail(STARK_EARG, "null argument");
    return tree_open_many(tree->ctx, tree->d, tree->n, (int)tree->levels, indices, count, paths_out, depth_out);
}

extern "C" size_t stark_tree_num_leaves(const stark_tree* tree) { return tree ? tree->n : 0; }

extern "C" int32_t stark_tree_download(stark_tree* tree, uint8_t* out)
{
    CtxGuard guard_(tree ? tree->ctx : nullptr);
    if (!tree || !out) return fail(STARK_EARG, "null argument");
    if (tree->levels != (uint32_t)log2_exact(tree->n)) return fail(STARK_EARG, "a forest of subtrees has no upper levels to download");
    HIP_TRY(copy_out(tree->ctx, out, tree->d, (2 * tree->n - 1) * sizeof(digest)));
    HIP_TRY(hipStreamSynchronize(tree->ctx->stream));
    return STARK_OK;
}

extern "C" void stark_tree_free(stark_tree* tree)
{
    CtxGuard guard_(tree ? tree->ctx : nullptr);
    if (!tree) return;
    if (tree->owns) {
        (void)hipStreamSynchronize(tree->ctx->stream);
        pool_free(tree->ctx, tree->d, (2 * tree->n - 1) * sizeof(digest));
    }
    delete tree;
}

// ===========================================================================
// FRI
// ===========================================================================
extern "C" size_t stark_fri_num_rounds(size_t domain_length, size_t expansion_factor, size_t num_colinearity_tests)
{
    size_t len = domain_length, rounds = 0;
    while (len > expansion_factor && 4 * num_colinearity_tests < len) {
        len /= 2;
        rounds++;
    }
    return rounds;
}

// power tables of omega^-1 covering exponents < N
static int32_t fold_tables(stark_ctx* ctx, fe omega, size_t N, PowRef* out)
{
    return get_pow_tables(ctx, ctx->F.inverse(omega), N, true, out);
}

// the two constants of a fold (fri.rs:133-138 rearranged: out = b + (a - b)(2^-1 + c omega^-i), c = 2^-1 alpha / offset)
static void fold_constants(const HostField& F, fe alpha, fe offset_inv, FoldArgs& a)
{
    static const fe two_inv = F.inverse(fe{2, 0});
    a.c5 = to_tw5(F, F.mul(F.mul(two_inv, alpha), offset_inv));
    a.half5 = to_tw5(F, two_inv);
}

extern "C" int32_t stark_fri_fold_dev(stark_ctx* ctx, const uint64_t* codeword_dev, size_t N,
                                      const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2],
                                      uint64_t* out_dev)
{
    CtxGuard guard_(ctx);
    if (!ctx || !codeword_dev || !alpha || !offset || !omega || !out_dev) return fail(STARK_EARG, "null argument");
    if (N < 2 || (N & 1)) return fail(STARK_ELENGTH, "codeword length must be even");
    fe al{alpha[0], alpha[1]}, of{offset[0], offset[1]}, om{omega[0], omega[1]};
    if (!is_canonical(al) || !is_canonical(of) || !is_canonical(om)) return fail(STARK_EARG, "scalar not reduced modulo p");
    PowRef pt;
    ST_TRY(fold_tables(ctx, om, N / 2, &pt));
    FoldArgs a{};
    a.cw = (const fe*)codeword_dev;
    a.out = (fe*)out_dev;
    a.leaf_out = nullptr;
    a.hi = (const tw5*)pt->hi;
    a.lo = (const tw5*)pt->lo;
    fold_constants(ctx->F, al, ctx->F.inverse(of), a);
    a.half = N / 2;
    a.shift = 0;
    HIP_TRY(launch_fri_fold(a, ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_fri_fold_shard_dev(stark_ctx* ctx, const uint64_t* local_dev, size_t local_len,
                                            const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2],
                                            size_t chunk_len, size_t period, size_t base, uint64_t* out_dev,
                                            uint8_t* chunk_roots_dev_out, stark_tree** forest_out)
{
    CtxGuard guard_(ctx);
    if (!ctx || !local_dev || !alpha || !offset || !omega || !out_dev) return fail(STARK_EARG, "null argument");
    const int log_chunk = log2_exact(chunk_len), log_period = log2_exact(period);
    if (log_chunk < 0 || log_period < 0) return fail(STARK_ENOTPOW2, "chunk length and period must be powers of two");
    if (chunk_len > period || base % chunk_len != 0 || base + chunk_len > period)
        return fail(STARK_EARG, "chunk does not lie inside one period");
    // the partner of position i is i + (layer length)/2: it is local only if the local share is a
    // whole, even number of chunks
    if (local_len == 0 || local_len % (2 * chunk_len) != 0)
        return fail(STARK_ELENGTH, "local share must be an even number of chunks");
    fe al{alpha[0], alpha[1]}, of{offset[0], offset[1]}, om{omega[0], omega[1]};
    if (!is_canonical(al) || !is_canonical(of) || !is_canonical(om)) return fail(STARK_EARG, "scalar not reduced modulo p");
    const size_t half = local_len / 2, chunks_out = half / chunk_len;
    PowRef pt;
    ST_TRY(fold_tables(ctx, om, chunks_out * period, &pt));     // positions of the outputs are < (layer length)/2
    FoldArgs a{};
    a.cw = (const fe*)local_dev;
    a.out = (fe*)out_dev;
    a.leaf_out = nullptr;
    a.hi = (const tw5*)pt->hi;
    a.lo = (const tw5*)pt->lo;
    fold_constants(ctx->F, al, ctx->F.inverse(of), a);
    a.half = half;
    a.shift = 0;
    a.log_chunk = (uint32_t)log_chunk;
    a.log_period = (uint32_t)log_period;
    a.base = base;
    if (!chunk_roots_dev_out) {
        if (forest_out) return fail(STARK_EARG, "a forest is only kept together with its roots");
        HIP_TRY(launch_fri_fold(a, ctx->stream));
        return STARK_OK;
    }
    if (forest_out && log2_exact(half) < 0)
        return fail(STARK_ENOTPOW2, "a retained forest needs a power-of-two number of chunks");
    // leaf digests of the outputs in the same pass, then one subtree per output chunk
    const size_t bytes = (2 * half - 1) * sizeof(digest);
    void* d = nullptr;
    ST_TRY(pool_alloc(ctx, bytes, &d));
    a.leaf_out = (digest*)d;
    digest* roots = nullptr;
    hipError_t e = launch_fri_fold(a, ctx->stream);
    if (e == hipSuccess) e = launch_merkle_reduce(a.leaf_out, half, ctx->stream, nullptr, chunks_out, &roots);
    if (e == hipSuccess)
        e = hipMemcpyAsync(chunk_roots_dev_out, roots, chunks_out * sizeof(digest), hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) {
        pool_free(ctx, d, bytes);
        return fail(STARK_EHIP, std::string("fold shard: ") + hipGetErrorString(e));
    }
    if (forest_out) *forest_out = new stark_tree{ctx, (digest*)d, half, true, (uint32_t)log_chunk};
    else pool_free(ctx, d, bytes);
    return STARK_OK;
}

extern "C" int32_t stark_fri_fold(stark_ctx* ctx, const uint64_t* codeword, size_t N, const uint64_t alpha[2],
                                  const uint64_t offset[2], const uint64_t omega[2], uint64_t* out)
{
    CtxGuard guard_(ctx);
    if (!ctx || !codeword || !out) return fail(STARK_EARG, "null argument");
    if (N < 2 || (N & 1)) return fail(STARK_ELENGTH, "codeword length must be even");
    DevBuf in(ctx), ob(ctx);
    ST_TRY(in.alloc(N * sizeof(fe)));
    ST_TRY(ob.alloc(N / 2 * sizeof(fe)));
    HIP_TRY(copy_in(ctx, in.p, codeword, N * sizeof(fe)));
    ST_TRY(stark_fri_fold_dev(ctx, (const uint64_t*)in.p, N, alpha, offset, omega, (uint64_t*)ob.p));
    HIP_TRY(copy_out(ctx, out, ob.p, N / 2 * sizeof(fe)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

static int32_t fri_commit_impl(stark_ctx* ctx, const fe* cw_src, bool src_is_host, size_t N, const uint64_t offset[2],
                               const uint64_t omega[2], size_t expansion_factor, size_t num_colinearity_tests,
                               stark_transcript_cb cb, void* user, stark_fri_layers** layers_out)
{
    if (!ctx || !cw_src || !offset || !omega || !cb || !layers_out) return fail(STARK_EARG, "null argument");
    const size_t rounds = stark_fri_num_rounds(N, expansion_factor, num_colinearity_tests);
    if (rounds == 0) return fail(STARK_EROUNDS, "Cannot do FRI with less than 1 round");
    if (log2_exact(N) < 0) return fail(STARK_ENOTPOW2, "length must be power of two");
    const HostField& F = ctx->F;
    fe om{omega[0], omega[1]}, of{offset[0], offset[1]};
    if (!is_canonical(om) || !is_canonical(of)) return fail(STARK_EARG, "scalar not reduced modulo p");
    // fri.rs:116: omega^(N-1) == omega^-1 at every round; omega^N == 1 implies it for all of them
    if (fe_is_zero(om) || !fe_eq(F.pow(om, (u128)N - 1), F.inverse(om)))
        return fail(STARK_EORDER, "error in commit: omega does not have the right order!");

    std::unique_ptr<stark_fri_layers> L(new stark_fri_layers());
    L->ctx = ctx;
    size_t total_elems = 0, total_digests = 0;
    for (size_t r = 0; r < rounds; r++) {
        L->len.push_back(N >> r);
        total_elems += N >> r;
        total_digests += 2 * (N >> r) - 1;
    }
    L->cw_bytes = total_elems * sizeof(fe);
    L->tree_bytes = total_digests * sizeof(digest);
    void* p = nullptr;
    ST_TRY(pool_alloc(ctx, L->cw_bytes, &p));
    L->cw_arena = (fe*)p;
    int32_t rc = pool_alloc(ctx, L->tree_bytes, &p);
    if (rc != STARK_OK) {
        pool_free(ctx, L->cw_arena, L->cw_bytes);
        return rc;
    }
    L->tree_arena = (digest*)p;
    {
        fe* c = L->cw_arena;
        digest* t = L->tree_arena;
        for (size_t r = 0; r < rounds; r++) {
            L->cw.push_back(c);
            L->tree.push_back(t);
            c += L->len[r];
            t += 2 * L->len[r] - 1;
        }
    }
    L->roots.resize(rounds);

    auto cleanup = [&](int32_t code) {
        (void)hipStreamSynchronize(ctx->stream);
        pool_free(ctx, L->cw_arena, L->cw_bytes);
        pool_free(ctx, L->tree_arena, L->tree_bytes);
        return code;
    };
#define FRI_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return cleanup(fail(STARK_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_))); \
    } while (0)

    PowRef pt;
    rc = fold_tables(ctx, om, N / 2, &pt);
    if (rc != STARK_OK) return cleanup(rc);

    if (src_is_host)                                                                                   // fri.rs:130
        FRI_HIP(copy_in(ctx, L->cw[0], cw_src, N * sizeof(fe)));
    // Roots arrive through host-mapped pinned memory, written by the kernel that
    // produces them; a one-leaf layer has no such kernel and is copied instead.
    digest* root_mail = (digest*)((uint8_t*)ctx->pinned + 2048);
    {
        digest* level = L->tree[0];
        size_t m = N;
        const uint32_t log_N = (uint32_t)log2_exact(N);
        if (src_is_host) FRI_HIP(launch_leaf_levels(L->cw[0], N, nullptr, L->tree[0], log_N, 1, ctx->stream, &level, &m));
        else FRI_HIP(launch_leaf_levels(cw_src, N, L->cw[0], L->tree[0], log_N, 1, ctx->stream, &level, &m));   // device source: copy in the hash pass
        FRI_HIP(launch_merkle_reduce(level, m, ctx->stream, root_mail));                                  // fri.rs:119
    }
    // 2^-1 alpha offset_r^-1 per round: the two inverses are taken once (a Fermat inverse is ~400
    // host multiplications) and offset_r^-1 = (offset^-1)^(2^r) follows by squaring (fri.rs:142)
    fe offset_inv_r = F.inverse(of);
    for (size_t r = 0; r < rounds; r++) {
        const size_t n = L->len[r];
        if (n == 1)
            FRI_HIP(hipMemcpyAsync(root_mail, L->tree[r], 32, hipMemcpyDeviceToHost, ctx->stream));
        FRI_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(L->roots[r].data(), root_mail, 32);
        const bool last = (r == rounds - 1);
        uint64_t alpha_raw[2] = {0, 0};
        if (cb(user, (uint32_t)r, last ? 1 : 0, L->roots[r].data(), alpha_raw) != 0)                   // fri.rs:120,126-127
            return cleanup(fail(STARK_ECALLBACK, "transcript callback failed"));
        if (last) break;                                                                               // fri.rs:123
        fe alpha{alpha_raw[0], alpha_raw[1]};
        if (!is_canonical(alpha)) return cleanup(fail(STARK_EARG, "alpha not reduced modulo p"));
        FoldArgs a{};
        a.cw = L->cw[r];
        a.out = L->cw[r + 1];
        a.leaf_out = L->tree[r + 1];
        a.hi = (const tw5*)pt->hi;
        a.lo = (const tw5*)pt->lo;
        fold_constants(F, alpha, offset_inv_r, a);
        a.half = n / 2;
        a.shift = (uint32_t)r;
        FRI_HIP(launch_fri_fold_commit(a, ctx->stream, root_mail));                                   // fri.rs:133-139,119
        offset_inv_r = F.mul(offset_inv_r, offset_inv_r);
    }
#undef FRI_HIP
    *layers_out = L.release();
    return STARK_OK;
}

extern "C" int32_t stark_fri_commit_dev(stark_ctx* ctx, const uint64_t* codeword_dev, size_t N,
                                        const uint64_t offset[2], const uint64_t omega[2], size_t expansion_factor,
                                        size_t num_colinearity_tests, stark_transcript_cb cb, void* user,
                                        stark_fri_layers** layers_out)
{
    CtxGuard guard_(ctx);
    return fri_commit_impl(ctx, (const fe*)codeword_dev, false, N, offset, omega, expansion_factor,
                           num_colinearity_tests, cb, user, layers_out);
}

extern "C" int32_t stark_fri_commit(stark_ctx* ctx, const uint64_t* codeword, size_t N, const uint64_t offset[2],
                                    const uint64_t omega[2], size_t expansion_factor, size_t num_colinearity_tests,
                                    stark_transcript_cb cb, void* user, stark_fri_layers** layers_out)
{
    CtxGuard guard_(ctx);
    return fri_commit_impl(ctx, (const fe*)codeword, true, N, offset, omega, expansion_factor,
                           num_colinearity_tests, cb, user, layers_out);
}

extern "C" size_t stark_fri_layers_count(const stark_fri_layers* l) { return l ? l->len.size() : 0; }

extern "C" size_t stark_fri_layer_len(const stark_fri_layers* l, size_t layer)
{
    return (l && layer < l->len.size()) ? l->len[layer] : 0;
}

extern "C" int32_t stark_fri_layer_root(const stark_fri_layers* l, size_t layer, uint8_t root[32])
{
    if (!l || !root || layer >= l->len.size()) return fail(STARK_EARG, "bad layer");
    memcpy(root, l->roots[layer].data(), 32);
    return STARK_OK;
}

extern "C" int32_t stark_fri_layer_download(stark_fri_layers* l, size_t layer, uint64_t* out)
{
    CtxGuard guard_(l ? l->ctx : nullptr);
    if (!l || !out || layer >= l->len.size()) return fail(STARK_EARG, "bad layer");
    HIP_TRY(copy_out(l->ctx, out, l->cw[layer], l->len[layer] * sizeof(fe)));
    HIP_TRY(hipStreamSynchronize(l->ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_fri_layer_read(stark_fri_layers* l, size_t layer, const uint64_t* indices, size_t count,
                                        uint64_t* out)
{
    CtxGuard guard_(l ? l->ctx : nullptr);
    if (!l || !indices || !out || layer >= l->len.size()) return fail(STARK_EARG, "bad layer");
    if (count == 0) return STARK_OK;
    for (size_t i = 0; i < count; i++)
        if (indices[i] >= l->len[layer]) return fail(STARK_EINDEX, "codeword index out of range");
    stark_ctx* ctx = l->ctx;
    DevBuf di(ctx), dov(ctx);
    ST_TRY(di.alloc(count * sizeof(uint64_t)));
    ST_TRY(dov.alloc(count * sizeof(fe)));
    HIP_TRY(hipMemcpyAsync(di.p, indices, count * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_gather_fe(l->cw[layer], (const uint64_t*)di.p, (fe*)dov.p, count, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, dov.p, count * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_vec_read_dev(stark_ctx* ctx, const uint64_t* vec_dev, size_t n, const uint64_t* indices,
                                      size_t count, uint64_t* out)
{
    CtxGuard guard_(ctx);
    if (!ctx || (count && (!vec_dev || !indices || !out))) return fail(STARK_EARG, "null argument");
    if (count == 0) return STARK_OK;
    for (size_t i = 0; i < count; i++)
        if (indices[i] >= n) return fail(STARK_EINDEX, "codeword index out of range");
    DevBuf di(ctx), dov(ctx);
    ST_TRY(di.alloc(count * sizeof(uint64_t)));
    ST_TRY(dov.alloc(count * sizeof(fe)));
    HIP_TRY(hipMemcpyAsync(di.p, indices, count * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_gather_fe((const fe*)vec_dev, (const uint64_t*)di.p, (fe*)dov.p, count, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, dov.p, count * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_fri_layer_open(stark_fri_layers* l, size_t layer, size_t index, uint8_t* path_out,
                                        size_t* depth_out)
{
    CtxGuard guard_(l ? l->ctx : nullptr);
    if (!l || !path_out || layer >= l->len.size()) return fail(STARK_EARG, "bad layer");
    return tree_open(l->ctx, l->tree[layer], l->len[layer], index, path_out, depth_out);
}

extern "C" int32_t stark_fri_layer_open_many(stark_fri_layers* l, size_t layer, const uint64_t* indices, size_t count,
                                             uint8_t* paths_out, size_t* depth_out)
{
    CtxGuard guard_(l ? l->ctx : nullptr);
    if (!l || layer >= l->len.size() || (count && (!indices || !paths_out))) return fail(STARK_EARG, "bad layer");
    return tree_open_many(l->ctx, l->tree[layer], l->len[layer], 64, indices, count, paths_out, depth_out);
}

extern "C" int32_t stark_fri_layer_query(stark_fri_layers* l, size_t layer, const uint64_t* c_indices, size_t count,
                                         uint64_t* values_out, uint8_t* paths_out, size_t* depth_out,
                                         size_t* next_depth_out)
{
    CtxGuard guard_(l ? l->ctx : nullptr);
    if (!l || layer + 1 >= l->len.size() || !c_indices || !values_out || !paths_out) return fail(STARK_EARG, "bad layer");
    if (count == 0) return STARK_OK;
    if (count > 0x7FFFFFFFull / 3) return fail(STARK_ELENGTH, "too many openings in one call");
    const size_t n = l->len[layer], half = n / 2, n_next = l->len[layer + 1];
    const int depth = log2_exact(n), depth_next = log2_exact(n_next);
    for (size_t i = 0; i < count; i++)
        if (c_indices[i] >= half || c_indices[i] >= n_next) return fail(STARK_EINDEX, "cannot open invalid index");
    if (depth_out) *depth_out = (size_t)depth;
    if (next_depth_out) *next_depth_out = (size_t)depth_next;
    stark_ctx* ctx = l->ctx;
    // one staging block: [3*count indices | 3*count values | paths], one copy in, one copy out, one sync
    const size_t idx_bytes = 3 * count * sizeof(uint64_t), val_bytes = 3 * count * sizeof(fe);
    const size_t path_digests = count * (2 * (size_t)depth + (size_t)depth_next);
    std::vector<uint64_t> idx(3 * count);
    for (size_t i = 0; i < count; i++) {
        idx[i] = c_indices[i];                   // a
        idx[count + i] = c_indices[i] + half;    // b (fri.rs:160-163)
        idx[2 * count + i] = c_indices[i];       // c, in the next layer
    }
    DevBuf stage(ctx);
    ST_TRY(stage.alloc(idx_bytes + val_bytes + path_digests * sizeof(digest) + 64));
    uint64_t* d_idx = (uint64_t*)stage.p;
    fe* d_val = (fe*)((uint8_t*)stage.p + idx_bytes);
    digest* d_path = (digest*)((uint8_t*)stage.p + idx_bytes + val_bytes);
    HIP_TRY(hipMemcpyAsync(d_idx, idx.data(), idx_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_gather_fe(l->cw[layer], d_idx, d_val, 2 * count, ctx->stream));
    HIP_TRY(launch_gather_fe(l->cw[layer + 1], d_idx + 2 * count, d_val + 2 * count, count, ctx->stream));
    HIP_TRY(launch_merkle_open_many(l->tree[layer], n, d_idx, 2 * count, (uint32_t)depth, d_path, ctx->stream));
    HIP_TRY(launch_merkle_open_many(l->tree[layer + 1], n_next, d_idx + 2 * count, count, (uint32_t)depth_next,
                                    d_path + 2 * count * (size_t)depth, ctx->stream));
    HIP_TRY(hipMemcpyAsync(values_out, d_val, val_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (path_digests)
        HIP_TRY(hipMemcpyAsync(paths_out, d_path, path_digests * sizeof(digest), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" int32_t stark_fri_query_all(stark_fri_layers* l, const uint64_t* top_indices, size_t count,
                                       uint64_t* values_out, uint8_t* paths_out)
{
    CtxGuard guard_(l ? l->ctx : nullptr);
    if (!l || l->len.empty() || !top_indices || !values_out || !paths_out) return fail(STARK_EARG, "null argument");
    const size_t rounds = l->len.size() - 1;
    if (count == 0 || rounds == 0) return STARK_OK;
    if (count > 0x7FFFFFFFull / 3) return fail(STARK_ELENGTH, "too many openings in one call");
    for (size_t i = 0; i < count; i++)
        if (top_indices[i] >= l->len[0]) return fail(STARK_EINDEX, "cannot open invalid index");
    stark_ctx* ctx = l->ctx;
    // one staging block for all rounds: [round table | indices | values | paths]; one copy in, ONE launch that
    // answers every opening of every round, two copies out, one synchronisation
    static_assert(sizeof(QueryRound) % sizeof(uint64_t) == 0, "the round table precedes the 64-bit indices");
    const size_t tab_words = rounds * (sizeof(QueryRound) / sizeof(uint64_t));
    const size_t val_bytes = rounds * 3 * count * sizeof(fe);
    std::vector<uint64_t> in((tab_words + rounds * 3 * count + 3) / 4 * 4);       // whole 32-byte units: values and paths follow
    size_t path_digests = 0;
    for (size_t r = 0; r < rounds; r++) {
        const size_t n = l->len[r], n_next = l->len[r + 1], half = n / 2;
        if (n_next < half) return fail(STARK_ELENGTH, "a layer is shorter than half of the one before");
        const int depth = log2_exact(n), depth_next = log2_exact(n_next);
        const QueryRound q{l->cw[r], l->cw[r + 1], l->tree[r], l->tree[r + 1], n, n_next, path_digests,
                           (uint32_t)depth, (uint32_t)depth_next};
        memcpy(in.data() + r * (sizeof(QueryRound) / sizeof(uint64_t)), &q, sizeof(q));
        path_digests += count * (2 * (size_t)depth + (size_t)depth_next);
        uint64_t* ir = in.data() + tab_words + r * 3 * count;
        for (size_t i = 0; i < count; i++) {
            const uint64_t c = top_indices[i] % half;
            ir[i] = c;                           // a
            ir[count + i] = c + half;            // b (fri.rs:160-163)
            ir[2 * count + i] = c;               // c, in the next layer
        }
    }
    const size_t in_bytes = in.size() * sizeof(uint64_t);
    DevBuf stage(ctx);
    ST_TRY(stage.alloc(in_bytes + val_bytes + path_digests * sizeof(digest) + 64));
    const QueryRound* d_tab = (const QueryRound*)stage.p;
    const uint64_t* d_idx = (const uint64_t*)stage.p + tab_words;
    fe* d_val = (fe*)((uint8_t*)stage.p + in_bytes);
    digest* d_path = (digest*)((uint8_t*)stage.p + in_bytes + val_bytes);
    HIP_TRY(hipMemcpyAsync(stage.p, in.data(), in_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_fri_query_all(d_tab, rounds, d_idx, count, d_val, d_path, ctx->stream));
    HIP_TRY(hipMemcpyAsync(values_out, d_val, val_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (path_digests)
        HIP_TRY(hipMemcpyAsync(paths_out, d_path, path_digests * sizeof(digest), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STARK_OK;
}

extern "C" void stark_fri_layers_free(stark_fri_layers* l)
{
    CtxGuard guard_(l ? l->ctx : nullptr);
    if (!l) return;
    (void)hipStreamSynchronize(l->ctx->stream);
    pool_free(l->ctx, l->cw_arena, l->cw_bytes);
    pool_free(l->ctx, l->tree_arena, l->tree_bytes);
    delete l;
}
