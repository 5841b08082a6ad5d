// C shims over the C++ host mirror (stark_host.hpp) so that the Python test
// suite can drive it.  Test plumbing only: a Rust or C++ caller uses the
// mirror / the C ABI directly.  Every function that can raise stark::Panic
// returns -1 and stores the message for sh_last_panic().
#include <cstdlib>
#include <cstring>
#include <string>

#include "stark_host.hpp"
#include "stark_prover.hpp"
#include "stark_scaled.hpp"

using namespace stark;

static thread_local std::string g_panic;

#define GUARD(...)                        \
    try {                                 \
        __VA_ARGS__                       \
    } catch (const std::exception& e) {   \
        g_panic = e.what();               \
        return -1;                        \
    }

static FieldElement ld(const uint64_t* p) { return FieldElement{((u128)p[1] << 64) | p[0]}; }
static void st(uint64_t* p, const FieldElement& x)
{
    p[0] = (uint64_t)x.value;
    p[1] = (uint64_t)(x.value >> 64);
}
static std::vector<FieldElement> ldv(const uint64_t* p, size_t n)
{
    std::vector<FieldElement> v(n);
    for (size_t i = 0; i < n; i++) v[i] = ld(p + 2 * i);
    return v;
}
static void stv(uint64_t* p, const std::vector<FieldElement>& v)
{
    for (size_t i = 0; i < v.size(); i++) st(p + 2 * i, v[i]);
}
static std::vector<std::vector<uint8_t>> leaves_of(const uint8_t* flat, const uint64_t* offs, size_t n)
{
    std::vector<std::vector<uint8_t>> out(n);
    for (size_t i = 0; i < n; i++) out[i].assign(flat + offs[i], flat + offs[i + 1]);
    return out;
}

extern "C" {

const char* sh_last_panic() { return g_panic.c_str(); }

// ---- field.rs ---------------------------------------------------------------
int sh_fe_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
    GUARD(
        FieldElement x = ld(a), y = b ? ld(b) : FieldElement::zero(), r;
        switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x / y; break;
        case 4: r = -x; break;
        case 5: r = x.inverse(); break;
        default: throw Panic("bad op");
        }
        st(out, r);
        return 0;)
}
int sh_fe_pow(const uint64_t* a, const uint64_t* e, uint64_t* out)
{
    GUARD(st(out, ld(a).pow(((u128)e[1] << 64) | e[0])); return 0;)
}
int sh_primitive_nth_root(const uint64_t* n, uint64_t* out)
{
    GUARD(st(out, FieldElement::primitive_nth_root(((u128)n[1] << 64) | n[0])); return 0;)
}
int sh_fe_sample(const uint8_t* bytes, size_t len, uint64_t* out)
{
    GUARD(st(out, FieldElement::sample(std::vector<uint8_t>(bytes, bytes + len))); return 0;)
}
size_t sh_fe_to_string(const uint64_t* a, char* buf)
{
    std::string s = ld(a).to_string();
    std::memcpy(buf, s.data(), s.size());
    return s.size();
}
size_t sh_bincode_fe(const uint64_t* a, uint8_t* buf)
{
    std::vector<uint8_t> b = wire::bincode_field_element(ld(a));
    std::memcpy(buf, b.data(), b.size());
    return b.size();
}

// ---- hashes -------------------------------------------------------------------
void sh_blake2b(const uint8_t* in, size_t len, uint8_t* out, size_t outlen)
{
    std::vector<uint8_t> d = hash::blake2b(in, len, outlen);
    std::memcpy(out, d.data(), outlen);
}
void sh_shake256(const uint8_t* in, size_t len, uint8_t* out, size_t outlen)
{
    std::vector<uint8_t> d = hash::shake256(in, len, outlen);
    std::memcpy(out, d.data(), outlen);
}

// ---- proof_stream.rs -------------------------------------------------------------
void* sh_ps_new() { return new ProofStream(); }
void sh_ps_free(void* ps) { delete static_cast<ProofStream*>(ps); }
void sh_ps_push(void* ps, const char* s, size_t len) { static_cast<ProofStream*>(ps)->push(std::string(s, len)); }
int sh_ps_pull(void* ps, char* buf, size_t* len)
{
    GUARD(
        std::string s = static_cast<ProofStream*>(ps)->pull();
        if (buf) std::memcpy(buf, s.data(), s.size());
        *len = s.size();
        return 0;)
}
size_t sh_ps_count(void* ps) { return static_cast<ProofStream*>(ps)->objects.size(); }
size_t sh_ps_object(void* ps, size_t i, char* buf)
{
    const std::string& s = static_cast<ProofStream*>(ps)->objects[i];
    if (buf) std::memcpy(buf, s.data(), s.size());
    return s.size();
}
size_t sh_ps_serialize(void* ps, uint8_t* buf)
{
    std::vector<uint8_t> b = static_cast<ProofStream*>(ps)->serialize();
    if (buf) std::memcpy(buf, b.data(), b.size());
    return b.size();
}
void* sh_ps_deserialize(const uint8_t* bytes, size_t len)
{
    try {
        return new ProofStream(ProofStream::deserialize(std::vector<uint8_t>(bytes, bytes + len)));
    } catch (const std::exception& e) {
        g_panic = e.what();
        return nullptr;
    }
}
void sh_ps_fiat_shamir(void* ps, int verifier, uint8_t* out, size_t nbytes)
{
    ProofStream* p = static_cast<ProofStream*>(ps);
    std::vector<uint8_t> d = verifier ? p->verifier_fiat_shamir(nbytes) : p->prover_fiat_shamir(nbytes);
    std::memcpy(out, d.data(), nbytes);
}

// ---- ntt.rs / univariate_poly.rs (device) ------------------------------------------
int sh_ntt(uint64_t* data, size_t len, size_t n, int inverse)
{
    GUARD(
        NTT ntt(n);
        std::vector<FieldElement> v = ldv(data, len);
        if (inverse) ntt.inverse(v); else ntt.forward(v);
        stv(data, v);
        return 0;)
}
int sh_ntt_evaluate(const uint64_t* coeffs, size_t n_coeffs, size_t n, uint64_t* out)
{
    GUARD(stv(out, NTT(n).evaluate(ldv(coeffs, n_coeffs))); return 0;)
}
int sh_ntt_multiply(const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out)
{
    GUARD(stv(out, NTT(8).multiply(ldv(a, na), ldv(b, nb))); return 0;)
}
int sh_eval_domain(const uint64_t* coeffs_high_first, size_t n_coeffs, const uint64_t* domain, size_t n, uint64_t* out)
{
    GUARD(
        Polynomial p{ldv(coeffs_high_first, n_coeffs)};
        stv(out, p.eval_domain(ldv(domain, n)));
        return 0;)
}
int sh_lagrange(const uint64_t* x, const uint64_t* y, size_t n, uint64_t* out, size_t* out_len)
{
    GUARD(
        Polynomial p = Polynomial::lagrange(ldv(x, n), ldv(y, n));
        stv(out, p.coeffs);
        *out_len = p.coeffs.size();
        return 0;)
}

// ---- merkle.rs ---------------------------------------------------------------------
int sh_merkle_commit(const uint8_t* flat, const uint64_t* offs, size_t n, uint8_t* root)
{
    GUARD(HashOutput r = Merkle::commit(leaves_of(flat, offs, n)); std::memcpy(root, r.data(), 32); return 0;)
}
int sh_merkle_open(size_t index, const uint8_t* flat, const uint64_t* offs, size_t n, uint8_t* path, size_t* depth)
{
    GUARD(
        std::vector<HashOutput> p = Merkle::open(index, leaves_of(flat, offs, n));
        for (size_t k = 0; k < p.size(); k++) std::memcpy(path + 32 * k, p[k].data(), 32);
        *depth = p.size();
        return 0;)
}
int sh_merkle_verify(const uint8_t* root, size_t index, const uint8_t* path, size_t depth, const uint8_t* leaf,
                     size_t leaf_len)
{
    GUARD(
        HashOutput r;
        std::memcpy(r.data(), root, 32);
        std::vector<HashOutput> p(depth);
        for (size_t k = 0; k < depth; k++) std::memcpy(p[k].data(), path + 32 * k, 32);
        return Merkle::verify(r, index, p, std::vector<uint8_t>(leaf, leaf + leaf_len)) ? 1 : 0;)
}
int sh_merkle_commit_host(const uint8_t* flat, const uint64_t* offs, size_t n, uint8_t* root)
{
    GUARD(HashOutput r = Merkle::commit_host(leaves_of(flat, offs, n)); std::memcpy(root, r.data(), 32); return 0;)
}

// ---- fri.rs ----------------------------------------------------------------------------
static Fri make_fri(const uint64_t* offset, const uint64_t* omega, size_t n, size_t expansion, size_t ncol)
{
    return Fri(ld(offset), ld(omega), n, expansion, ncol);
}
int sh_fri_num_rounds(const uint64_t* offset, const uint64_t* omega, size_t n, size_t expansion, size_t ncol)
{
    GUARD(return (int)make_fri(offset, omega, n, expansion, ncol).num_rounds();)
}
int sh_fri_sample_indices(const uint8_t* seed, size_t seed_len, size_t size, size_t reduced, size_t number,
                          uint64_t* out)
{
    GUARD(
        uint64_t one[2] = {1, 0};
        Fri f = make_fri(one, one, 1 << 20, 4, 1);
        std::vector<size_t> idx = f.sample_indices(std::vector<uint8_t>(seed, seed + seed_len), size, reduced, number);
        for (size_t i = 0; i < idx.size(); i++) out[i] = idx[i];
        return 0;)
}
// layers_out: all codewords concatenated (may be NULL); returns the number of layers
int sh_fri_commit(const uint64_t* cw, size_t n, const uint64_t* offset, const uint64_t* omega, size_t expansion,
                  size_t ncol, void* ps, uint64_t* layers_out)
{
    GUARD(
        Fri f = make_fri(offset, omega, n, expansion, ncol);
        std::vector<std::vector<FieldElement>> layers = f.commit(ldv(cw, n), *static_cast<ProofStream*>(ps));
        size_t off = 0;
        if (layers_out)
            for (auto& l : layers) { stv(layers_out + 2 * off, l); off += l.size(); }
        return (int)layers.size();)
}
// commit of a device-resident codeword (device pointer from stark_dev_alloc on the process
// context); the layers stay on the device and are released before returning
int sh_fri_commit_resident(const void* cw_dev, size_t n, const uint64_t* offset, const uint64_t* omega, size_t expansion,
                           size_t ncol, void* ps)
{
    GUARD(
        Fri f = make_fri(offset, omega, n, expansion, ncol);
        std::unique_ptr<LayersHandle> h = f.commit_resident((const uint64_t*)cw_dev, *static_cast<ProofStream*>(ps));
        return (int)stark_fri_layers_count(h->l);)
}
// Fri::prove (fri.rs:212-254) of a codeword resident on the device: commit, then the query phase
// answered from the retained layers; returns the number of indices written (2 * ncol).
int sh_fri_prove_resident(const void* cw_dev, size_t n, const uint64_t* offset, const uint64_t* omega, size_t expansion,
                          size_t ncol, void* ps, uint64_t* indices_out)
{
    GUARD(
        Fri f = make_fri(offset, omega, n, expansion, ncol);
        std::vector<size_t> idx = f.prove_resident((const uint64_t*)cw_dev, *static_cast<ProofStream*>(ps));
        for (size_t i = 0; i < idx.size(); i++) indices_out[i] = idx[i];
        return (int)idx.size();)
}
// the process-wide device context the mirror runs on (for callers that stage device buffers)
void* sh_context()
{
    try {
        return (void*)context();
    } catch (const std::exception& e) {
        g_panic = e.what();
        return nullptr;
    }
}
int sh_fri_prove(const uint64_t* cw, size_t n, const uint64_t* offset, const uint64_t* omega, size_t expansion,
                 size_t ncol, void* ps, uint64_t* indices_out)
{
    GUARD(
        Fri f = make_fri(offset, omega, n, expansion, ncol);
        std::vector<size_t> idx = f.prove(ldv(cw, n), *static_cast<ProofStream*>(ps));
        for (size_t i = 0; i < idx.size(); i++) indices_out[i] = idx[i];
        return (int)idx.size();)
}
// prove through the reference-shaped commit() + query() (codewords by value, trees rebuilt)
int sh_fri_prove_by_value(const uint64_t* cw, size_t n, const uint64_t* offset, const uint64_t* omega,
                          size_t expansion, size_t ncol, void* psv)
{
    GUARD(
        Fri f = make_fri(offset, omega, n, expansion, ncol);
        ProofStream& ps = *static_cast<ProofStream*>(psv);
        std::vector<std::vector<FieldElement>> cws = f.commit(ldv(cw, n), ps);
        std::vector<size_t> top = f.sample_indices(ps.prover_fiat_shamir(32), cws[0].size() / 2, cws.back().size(), ncol);
        std::vector<size_t> indices = top;
        for (size_t i = 0; i + 1 < cws.size(); i++) {
            for (size_t& idx : indices) idx %= cws[i].size() / 2;
            indices = f.query(cws[i], cws[i + 1], indices, ps);
        }
        return 0;)
}
// returns 1 (accept) / 0 (reject) / -1 (panic); points as (index, value) pairs
int sh_fri_verify(const uint64_t* offset, const uint64_t* omega, size_t n, size_t expansion, size_t ncol, void* ps,
                  uint64_t* point_idx, uint64_t* point_val, size_t* n_points)
{
    GUARD(
        Fri f = make_fri(offset, omega, n, expansion, ncol);
        std::vector<std::pair<size_t, FieldElement>> pts;
        bool ok = f.verify(*static_cast<ProofStream*>(ps), pts);
        if (n_points) {
            for (size_t i = 0; i < pts.size(); i++) {
                if (point_idx) point_idx[i] = pts[i].first;
                if (point_val) st(point_val + 2 * i, pts[i].second);
            }
            *n_points = pts.size();
        }
        return ok ? 1 : 0;)
}

// ---- univariate_poly.rs algebra (coefficients highest first) -------------------------------
// op: 0 add, 1 sub, 2 mul, 3 div, 4 compose, 5 pow (b[0] = exponent), 6 zeroifier_domain(a), 7 neg
int sh_poly_op(int op, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t* nout)
{
    GUARD(
        Polynomial pa{ldv(a, na)}, pb{ldv(b, nb)}, r;
        switch (op) {
        case 0: r = pa + pb; break;
        case 1: r = pa - pb; break;
        case 2: r = pa * pb; break;
        case 3: r = pa / pb; break;
        case 4: r = pa.compose(pb); break;
        case 5: r = pa.pow(((u128)b[1] << 64) | b[0]); break;
        case 6: r = Polynomial::zeroifier_domain(pa.coeffs); break;
        case 7: r = -pa; break;
        default: throw Panic("bad op");
        }
        if (out) stv(out, r.coeffs);
        *nout = r.coeffs.size();
        return 0;)
}

// Polynomial::save then Polynomial::load through `path`; returns the file's bytes
int sh_poly_save_load(const uint64_t* a, size_t na, const char* path, uint64_t* out, size_t* nout)
{
    GUARD(
        Polynomial{ldv(a, na)}.save(path);
        Polynomial r = Polynomial::load(path);
        stv(out, r.coeffs);
        *nout = r.coeffs.size();
        return 0;)
}

// ---- rescue_prime.rs -----------------------------------------------------------------------
// round_constants: 108 elements, mds / mds_inv: 4 elements each (row major), alpha_inv: 1 element
int sh_rescue_params(uint64_t* round_constants, uint64_t* mds, uint64_t* mds_inv, uint64_t* alpha_inv)
{
    GUARD(
        RescuePrime rp;
        stv(round_constants, rp.round_constants);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) {
                st(mds + 2 * (2 * i + j), rp.MDS[i][j]);
                st(mds_inv + 2 * (2 * i + j), rp.MDS_inv[i][j]);
            }
        alpha_inv[0] = (uint64_t)rp.alpha_inv;
        alpha_inv[1] = (uint64_t)(rp.alpha_inv >> 64);
        return 0;)
}
int sh_rescue_hash(const uint64_t* in, uint64_t* out)
{
    GUARD(st(out, RescuePrime().hash(ld(in))); return 0;)
}
// number of stored terms of each transition constraint, and each one's value at a point of 5 elements
int sh_rescue_air(const uint64_t* omicron, const uint64_t* point5, uint64_t* n_terms, uint64_t* values)
{
    GUARD(
        RescuePrime rp;
        std::vector<MPolynomial> air = rp.transition_constraints(ld(omicron));
        std::vector<FieldElement> pt = ldv(point5, 5);
        for (size_t i = 0; i < air.size(); i++) {
            n_terms[i] = air[i].dict.size();
            st(values + 2 * i, air[i].eval(pt));
        }
        return (int)air.size();)
}

// ---- stark.rs ------------------------------------------------------------------------------
// Proof that output = RescuePrime::hash(input) (the flow of stark.rs:731-776).  `random` feeds
// the prover's draws in order, 17 bytes each; returns the number of bytes consumed.
int sh_stark_rescue_prove(const uint64_t* input, size_t expansion, size_t ncol, size_t security, const uint8_t* random,
                          size_t n_random, uint8_t* proof_out, size_t* proof_len, uint64_t* output)
{
    GUARD(
        RescuePrime rp;
        FieldElement in = ld(input), out = rp.hash(in);
        Stark stark(expansion, ncol, security, rp.m, rp.N + 1);
        size_t used = 0;
        RandomSource rng = [&](uint8_t* dst, size_t n) {
            if (used + n > n_random) throw Panic("random bytes exhausted");
            std::memcpy(dst, random + used, n);
            used += n;
        };
        std::vector<uint8_t> proof =
            stark.prove(rp.trace(in), rp.transition_constraints(stark.omicron), rp.boundary_constraints(out), rng);
        if (proof_out) {
            if (*proof_len < proof.size()) throw Panic("proof buffer too small");
            std::memcpy(proof_out, proof.data(), proof.size());
        }
        *proof_len = proof.size();
        if (output) st(output, out);
        return (int)used;)
}
// 1 accept / 0 reject / -1 panic
int sh_stark_rescue_verify(const uint64_t* output, size_t expansion, size_t ncol, size_t security, const uint8_t* proof,
                           size_t proof_len)
{
    GUARD(
        RescuePrime rp;
        Stark stark(expansion, ncol, security, rp.m, rp.N + 1);
        return stark.verify(std::vector<uint8_t>(proof, proof + proof_len), rp.transition_constraints(stark.omicron),
                            rp.boundary_constraints(ld(output)))
                   ? 1
                   : 0;)
}

// ---- the long-trace prover (stark_scaled.hpp) ----------------------------------
// A Rescue-Prime chain of `rows` rows with its trace, AIR, constant columns and the ScaledStark made for it.
struct ChainJob {
    RescueChain chain;
    ScaledStark stark;
    std::vector<std::vector<FieldElement>> trace, constants;
    std::vector<FieldElement> columns;               // the trace by columns (register s at s * rows), made on first use
    std::vector<MPolynomial> air;
    std::unique_ptr<ScaledStark::Prepared> constant_polys;   // the statement's public polynomials, made once
    bool use_prepared = true;                        // off: prove and verify interpolate the constant columns themselves
    ChainJob(const FieldElement& input, size_t rows, size_t expansion, size_t ncol, size_t security)
        : chain(rows), stark(expansion, ncol, security, chain.rp.m, rows), trace(chain.trace(input)),
          constants(chain.constant_columns()), air(chain.air()), constant_polys(stark.prepare(constants))
    {
    }
    ScaledStark::TraceColumns trace_columns()
    {
        const size_t rows = trace.size(), regs = chain.rp.m;
        if (columns.empty()) {
            columns.resize(rows * regs);
            for (size_t c = 0; c < rows; c++)
                for (size_t s = 0; s < regs; s++) columns[s * rows + c] = trace[c][s];
        }
        return ScaledStark::TraceColumns{columns.data(), rows};
    }
};

void* sh_chain_new(const uint64_t* input, size_t rows, size_t expansion, size_t ncol, size_t security)
{
    try {
        return new ChainJob(ld(input), rows, expansion, ncol, security);
    } catch (const std::exception& e) {
        g_panic = e.what();
        return nullptr;
    }
}
void sh_chain_free(void* h) { delete (ChainJob*)h; }
int sh_chain_output(void* h, uint64_t* out)
{
    GUARD(st(out, ((ChainJob*)h)->trace.back()[0]); return 0;)
}
// random bytes one proof draws: 17 per randomizer value and per randomizer coefficient
size_t sh_chain_random_bytes(void* h)
{
    ChainJob* j = (ChainJob*)h;
    return 17 * (j->stark.num_randomizers * j->stark.num_registers + j->stark.max_degree(j->air) + 1);
}
// adds 1 to one trace value: the trace no longer follows the round function there
void sh_chain_corrupt(void* h, size_t row, size_t reg)
{
    ChainJob* j = (ChainJob*)h;
    j->trace.at(row).at(reg) = j->trace[row][reg] + FieldElement::one();
    j->columns.clear();
}
void sh_chain_use_prepared_constants(void* h, int on) { ((ChainJob*)h)->use_prepared = on != 0; }
size_t sh_chain_fri_domain_length(void* h) { return ((ChainJob*)h)->stark.fri_domain_length; }
// returns the random bytes used; timings (may be NULL) receives host, interpolate, quotients, lde, merkle, fri,
// openings, total in milliseconds
int sh_chain_prove(void* h, const uint8_t* random, size_t n_random, uint8_t* proof_out, size_t* proof_len,
                   double* timings)
{
    GUARD(
        ChainJob* j = (ChainJob*)h;
        size_t used = 0;
        RandomSource rng = [&](uint8_t* dst, size_t n) {
            if (used + n > n_random) throw Panic("random bytes exhausted");
            std::memcpy(dst, random + used, n);
            used += n;
        };
        ProveTimings tm;
        std::vector<uint8_t> proof = j->stark.prove(j->trace, j->air, j->constants,
                                                    j->chain.boundary_constraints(j->trace.back()[0]), rng, &tm,
                                                    j->use_prepared ? j->constant_polys.get() : nullptr);
        if (proof_out) {
            if (*proof_len < proof.size()) throw Panic("proof buffer too small");
            std::memcpy(proof_out, proof.data(), proof.size());
        }
        *proof_len = proof.size();
        if (timings) {
            const double t[8] = {tm.host, tm.interpolate, tm.quotients, tm.lde, tm.merkle, tm.fri, tm.openings, tm.total};
            std::memcpy(timings, t, sizeof t);
        }
        return (int)std::min<size_t>(used, 0x7fffffff);)
}
// the same through ScaledStark::prove_columns: the trace is handed over by columns (kept that way by the job)
int sh_chain_prove_columns(void* h, const uint8_t* random, size_t n_random, uint8_t* proof_out, size_t* proof_len,
                           double* timings)
{
    GUARD(
        ChainJob* j = (ChainJob*)h;
        size_t used = 0;
        RandomSource rng = [&](uint8_t* dst, size_t n) {
            if (used + n > n_random) throw Panic("random bytes exhausted");
            std::memcpy(dst, random + used, n);
            used += n;
        };
        ProveTimings tm;
        std::vector<uint8_t> proof = j->stark.prove_columns(j->trace_columns(), j->air, j->constants,
                                                            j->chain.boundary_constraints(j->trace.back()[0]), rng, &tm,
                                                            j->use_prepared ? j->constant_polys.get() : nullptr);
        if (proof_out) {
            if (*proof_len < proof.size()) throw Panic("proof buffer too small");
            std::memcpy(proof_out, proof.data(), proof.size());
        }
        *proof_len = proof.size();
        if (timings) {
            const double t[8] = {tm.host, tm.interpolate, tm.quotients, tm.lde, tm.merkle, tm.fri, tm.openings, tm.total};
            std::memcpy(timings, t, sizeof t);
        }
        return (int)std::min<size_t>(used, 0x7fffffff);)
}

// ---- the steps of ScaledStark::prove one at a time, for a prover spread over several GPUs: every rank runs
// them alike and does the FRI-domain work (extensions, commitments, FRI, openings) on its share
struct ChainSession {
    ChainJob* job;
    const uint8_t* random;
    size_t n_random, used = 0;
    RandomSource rng;
    ScaledStark::Run run;                            // destroyed first: joins the helper thread that reads `rng`
    ChainSession(ChainJob* j, const uint8_t* r, size_t n) : job(j), random(r), n_random(n)
    {
        rng = [this](uint8_t* dst, size_t k) {
            if (used + k > n_random) throw Panic("random bytes exhausted");
            std::memcpy(dst, random + used, k);
            used += k;
        };
    }
};
// `random` must stay valid until sh_session_free
void* sh_session_begin(void* h, const uint8_t* random, size_t n_random)
{
    ChainSession* s = nullptr;
    try {
        ChainJob* j = (ChainJob*)h;
        s = new ChainSession(j, random, n_random);
        j->stark.begin(s->run, j->trace_columns(), j->air, j->constants.size(),
                       j->chain.boundary_constraints(j->trace.back()[0]), s->rng,
                       j->use_prepared ? j->constant_polys.get() : nullptr);
        return s;
    } catch (const std::exception& e) {
        g_panic = e.what();
        delete s;
        return nullptr;
    }
}
void sh_session_free(void* v) { delete (ChainSession*)v; }
// out: registers, boundary-quotient coefficients per register, their column stride, randomizer / combination
// coefficients, FRI domain length, expansion factor, colinearity tests
void sh_session_shape(void* v, uint64_t out[8])
{
    ChainSession* s = (ChainSession*)v;
    const ScaledStark& st = s->job->stark;
    const uint64_t vals[8] = {s->run.R, s->run.n_rows, s->run.n_b, s->run.max_deg + 1, st.fri_domain_length,
                              st.expansion_factor, st.num_colinearity_tests, 0};
    std::memcpy(out, vals, sizeof vals);
}
// device pointers of the step results: 0 boundary quotients, 1 randomizer, 2 combination (NULL before their step)
const void* sh_session_coefficients(void* v, int which)
{
    ChainSession* s = (ChainSession*)v;
    const DeviceVector* d = which == 0 ? s->run.bq.get() : which == 1 ? s->run.randomizer.get() : s->run.combination.get();
    return d ? d->words() : nullptr;
}
int sh_session_transition(void* v)
{
    GUARD(
        ChainSession* s = (ChainSession*)v;
        s->job->stark.transition_quotients(s->run, s->job->air, s->job->constants);
        s->job->stark.randomizer_polynomial(s->run);
        return 0;)
}
int sh_session_combine(void* v, const uint8_t* challenge, size_t len)
{
    GUARD(
        ChainSession* s = (ChainSession*)v;
        s->job->stark.combine(s->run, std::vector<uint8_t>(challenge, challenge + len));
        return 0;)
}
// the degree assertions of the steps; timings as sh_chain_prove gives them (the caller's own laps not included)
int sh_session_finish(void* v, double* timings)
{
    GUARD(
        ChainSession* s = (ChainSession*)v;
        s->job->stark.check_degrees(s->run);
        if (timings) {
            const ProveTimings& tm = s->run.tm;
            const double t[8] = {tm.host, tm.interpolate, tm.quotients, tm.lde, tm.merkle, tm.fri, tm.openings, tm.total};
            std::memcpy(timings, t, sizeof t);
        }
        return 0;)
}

// 1 accept / 0 reject / -1 panic
int sh_chain_verify(void* h, const uint64_t* output, const uint8_t* proof, size_t proof_len)
{
    GUARD(
        ChainJob* j = (ChainJob*)h;
        return j->stark.verify(std::vector<uint8_t>(proof, proof + proof_len), j->air, j->constants,
                               j->chain.boundary_constraints(ld(output)),
                               j->use_prepared ? j->constant_polys.get() : nullptr)
                   ? 1
                   : 0;)
}
// Polynomial::lagrange over (q^i, ys[c][i]), i < m, cols columns back to back, through GeometricDomain; also
// the zeroifier of the first m points (m + 1 coefficients, lowest first) if zeroifier_out is not NULL
int sh_geometric_interpolate(const uint64_t* q, size_t n_table, const uint64_t* ys, size_t m, size_t cols,
                             uint64_t* coeffs_out, uint64_t* zeroifier_out)
{
    GUARD(
        GeometricDomain geo(ld(q), n_table);
        DeviceVector in(std::max<size_t>(1, m * cols)), out(std::max<size_t>(1, m * cols));
        in.upload(ldv(ys, m * cols));
        geo.interpolate(in.words(), m, cols, m, out.words(), m);
        stv(coeffs_out, out.download(0, m * cols));
        if (zeroifier_out) stv(zeroifier_out, geo.zeroifier(m));
        return 0;)
}

}  // extern "C"
