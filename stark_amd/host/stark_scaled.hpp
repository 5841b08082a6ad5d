// Stark::prove / Stark::verify (stark.rs:223-723) for traces far longer than the
// 28 rows the reference is written for, with the polynomial algebra on the
// device.  The prover keeps the reference's objects and their order on the proof
// stream -- randomizer rows, trace polynomials, boundary quotients, transition
// quotients, randomizer polynomial, weights, nonlinear combination, FRI proof,
// openings -- and computes each of them without the reference's O(n^2)..O(n^3)
// algebra (lagrange_traditional univariate_poly.rs:147-164, schoolbook Mul
// :403-420, long Div :437-484, compose :203-221, zeroifier_domain :254-264):
//
//  * the trace domain omicron^0 .. omicron^(n-1) (stark.rs:255-257) is a
//    geometric progression.  Interpolation over it is two power-series products
//    (Newton basis of a geometric progression through the q-binomial theorem;
//    Bostan & Schost 2005) and its zeroifier has closed-form coefficients;
//  * products, exact divisions and the evaluation of the constraints happen on
//    the values of the polynomials on a coset of the omicron subgroup, between a
//    coset transform and its inverse;
//  * the AIR is an MPolynomial over [x, current registers, next registers,
//    constant columns]: a constant column is a public vector with one value per
//    transition (the round constants of that step), interpolated over
//    omicron^0 .. omicron^(T-2) as RescuePrime::round_constants_polynomials
//    (rescue_prime.rs:309-359) does for its 27 rounds, instead of being lifted
//    into the variable x, which for a T-row trace is an expression of about
//    30*T terms.  Substituting the column polynomials for their variables gives
//    the reference's constraints back.
//
// Interpolants, zeroifiers and exact quotients are unique, so these are the
// reference's polynomials coefficient for coefficient, and at the reference's
// own size the proof is the reference algorithm's proof byte for byte.
#pragma once
#include <atomic>
#include <chrono>
#include <future>
#include <mutex>
#include <thread>

#include "stark_prover.hpp"

namespace stark {

inline size_t next_pow2(size_t n)
{
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

inline void fe_words(const FieldElement& x, uint64_t w[2])
{
    w[0] = (uint64_t)x.value;
    w[1] = (uint64_t)(x.value >> 64);
}

// A block of device memory holding `count` field elements.
struct DeviceVector {
    DeviceBuffer buf;
    size_t count;
    explicit DeviceVector(size_t n) : buf(n * 16), count(n) {}
    uint64_t* words(size_t element = 0) const { return buf.words() + 2 * element; }
    void upload(const std::vector<FieldElement>& v, size_t at = 0) const
    {
        if (at + v.size() > count) throw Panic("upload past the end of a device vector");
        if (!v.empty()) check(stark_copy_h2d(context(), words(at), v.data(), v.size() * 16));
    }
    std::vector<FieldElement> download(size_t at, size_t n) const
    {
        std::vector<FieldElement> v(n);
        if (n) check(stark_copy_d2h(context(), v.data(), words(at), n * 16));
        return v;
    }
};
static_assert(sizeof(FieldElement) == 16, "FieldElement is the packed {lo, hi} image");

// Tables over u_i = prod_{j=1..i} (q^j - 1), i <= n, for the points q^0, q^1, ...:
//   inv_u[i] = 1/u_i,  e[i] = (-1)^i q^(i(i-1)/2) / u_i,  t2[i] = q^(-i(i-1)/2) u_i.
// By the q-binomial theorem prod_{j<k} (x - q^j) = sum_j (-1)^(k-j) q^C(k-j,2) [k over j]_q x^j with
// [k over j]_q = u_k / (u_j u_(k-j)); the Newton coefficients c_k of the interpolant of (q^i, y_i) satisfy
// y_i / u_i = sum_k (c_k q^C(k,2)) / u_(i-k), and 1 / (sum_i x^i / u_i) = sum_i e[i] x^i.
struct GeometricDomain {
    FieldElement q;
    size_t n = 0;                 // points the tables cover
    size_t M = 0;                 // transform length of the two products, >= 2n - 1
    std::vector<FieldElement> inv_u, e;
    std::vector<FieldElement> u;
    std::unique_ptr<DeviceVector> d_inv_u, d_t2, d_ehat, d_erevhat;

    GeometricDomain(const FieldElement& q_, size_t n_) : q(q_), n(n_)
    {
        using namespace starkf;
        const HostField& F = host_field();
        const size_t len = n + 1;
        // everything in Montgomery form, one mont_mul per product
        std::vector<fe> um(len), qcm(len), pre(len);
        const fe one = F.one_m, qm = F.to_mont(q.raw());
        fe qi = one;
        um[0] = one;
        qcm[0] = one;
        for (size_t i = 1; i < len; i++) {
            qcm[i] = mont_mul(qcm[i - 1], qi);                 // q^C(i,2) = q^C(i-1,2) q^(i-1)
            qi = mont_mul(qi, qm);                             // q^i
            um[i] = mont_mul(um[i - 1], fe_sub(qi, one));
        }
        auto batch_inverse = [&](const std::vector<fe>& v) {
            std::vector<fe> out(v.size());
            fe acc = one;
            for (size_t i = 0; i < v.size(); i++) {
                if (fe_is_zero(v[i])) throw Panic("the points of a geometric domain must be distinct");
                pre[i] = acc;
                acc = mont_mul(acc, v[i]);
            }
            fe inv = F.to_mont(F.inverse(F.from_mont(acc)));
            for (size_t i = v.size(); i-- > 0;) {
                out[i] = mont_mul(inv, pre[i]);
                inv = mont_mul(inv, v[i]);
            }
            return out;
        };
        std::vector<fe> ium = batch_inverse(um), iqcm = batch_inverse(qcm);
        inv_u.resize(len);
        e.resize(len);
        u.resize(len);
        std::vector<FieldElement> t2(len);
        const fe raw_one{1, 0};
        for (size_t i = 0; i < len; i++) {
            inv_u[i] = FieldElement::wrap(mont_mul(ium[i], raw_one));
            fe ei = mont_mul(mont_mul(qcm[i], ium[i]), raw_one);
            e[i] = FieldElement::wrap((i & 1) ? fe_neg(ei) : ei);
            t2[i] = FieldElement::wrap(mont_mul(mont_mul(iqcm[i], um[i]), raw_one));
            u[i] = FieldElement::wrap(mont_mul(um[i], raw_one));
        }
        // device tables: inv_u, t2, and the transforms of e[0..n) and of its reversal
        M = next_pow2(2 * n > 1 ? 2 * n - 1 : 1);
        d_inv_u.reset(new DeviceVector(n));
        d_t2.reset(new DeviceVector(n));
        d_inv_u->upload(std::vector<FieldElement>(inv_u.begin(), inv_u.begin() + n));
        d_t2->upload(std::vector<FieldElement>(t2.begin(), t2.begin() + n));
        DeviceVector tmp(2 * n);
        tmp.upload(std::vector<FieldElement>(e.begin(), e.begin() + n));
        tmp.upload(std::vector<FieldElement>(e.rend() - n, e.rend()), n);   // e[n-1], ..., e[0]
        d_ehat.reset(new DeviceVector(M));
        d_erevhat.reset(new DeviceVector(M));
        const uint64_t one_w[2] = {1, 0};
        check(stark_coset_lde_batch_dev(context(), tmp.words(0), n, n, 1, M, one_w, d_ehat->words(), M));
        check(stark_coset_lde_batch_dev(context(), tmp.words(n), n, n, 1, M, one_w, d_erevhat->words(), M));
        check(stark_ctx_synchronize(context()));
    }

    // Polynomial::lagrange over (q^i, ys[c][i]), i < m, for `cols` columns at once (univariate_poly.rs:127-164):
    // column c is the m values at ys + c * in_stride; its m coefficients (lowest first) go to out + c * out_stride.
    // Two products per column, each one transform, a pointwise product with a table and an inverse transform.
    void interpolate(const uint64_t* ys_dev, size_t m, size_t cols, size_t in_stride, uint64_t* out_dev,
                     size_t out_stride) const
    {
        if (m == 0 || cols == 0) return;
        if (m > n) throw Panic("more points than the geometric domain holds");
        stark_ctx* ctx = context();
        DeviceVector a(cols * m), w(cols * M);
        const uint64_t one_w[2] = {1, 0};
        for (size_t c = 0; c < cols; c++)                              // y_i / u_i
            check(stark_vec_mul_dev(ctx, ys_dev + 2 * c * in_stride, d_inv_u->words(), a.words(c * m), m));
        check(stark_coset_lde_batch_dev(ctx, a.words(), m, m, cols, M, one_w, w.words(), M));
        for (size_t c = 0; c < cols; c++) check(stark_vec_mul_dev(ctx, w.words(c * M), d_ehat->words(), w.words(c * M), M));
        check(stark_ntt_batch_dev(ctx, w.words(), M, cols, M, 1));     // low m terms: c_k q^C(k,2)
        for (size_t c = 0; c < cols; c++)                              // c_k u_k
            check(stark_vec_mul_dev(ctx, w.words(c * M), d_t2->words(), a.words(c * m), m));
        check(stark_coset_lde_batch_dev(ctx, a.words(), m, m, cols, M, one_w, w.words(), M));
        for (size_t c = 0; c < cols; c++) check(stark_vec_mul_dev(ctx, w.words(c * M), d_erevhat->words(), w.words(c * M), M));
        check(stark_ntt_batch_dev(ctx, w.words(), M, cols, M, 1));     // term n-1+j: p_j u_j
        for (size_t c = 0; c < cols; c++)
            check(stark_vec_mul_dev(ctx, w.words(c * M + n - 1), d_inv_u->words(), out_dev + 2 * c * out_stride, m));
        check(stark_ctx_synchronize(ctx));                             // the work buffers go out of scope
    }

    // Polynomial::zeroifier_domain over q^0 .. q^(m-1) (univariate_poly.rs:254-264): m + 1 coefficients, lowest first
    std::vector<FieldElement> zeroifier(size_t m) const
    {
        if (m > n) throw Panic("more points than the geometric domain holds");
        std::vector<FieldElement> z(m + 1);
        for (size_t j = 0; j <= m; j++) z[j] = u[m] * e[m - j] * inv_u[j];
        return z;
    }
};

// milliseconds spent by ScaledStark::prove, by what the time went to
struct ProveTimings {
    double host = 0, interpolate = 0, quotients = 0, lde = 0, merkle = 0, fri = 0, openings = 0, total = 0;
};

struct ScaledStark {
    size_t expansion_factor, num_colinearity_tests, security_level, num_randomizers, num_registers,
        original_trace_length;
    size_t omicron_domain_length, fri_domain_length;
    FieldElement generator, omega, omicron;
    Fri fri;
    GeometricDomain geo;
    // kept from one proof to the next: the transition zeroifier's coefficients on the device (they depend
    // on the trace length only) and the staging buffer of the randomizer's raw bytes
    mutable std::unique_ptr<DeviceVector> tz_coeffs, coset_x, coset_tz_inv;
    mutable size_t coset_n_q = 0;
    mutable std::vector<uint8_t> random_staging;
    mutable std::vector<FieldElement> column_staging;         // the trace by columns, on its way to the device
    mutable std::mutex staging_mu;

    // stark.rs:34-86 without the assertion on omicron's value (:65) and without the table of
    // omicron's powers (:63), which only the zeroifier needed
    ScaledStark(size_t expansion_factor_, size_t num_colinearity_tests_, size_t security_level_, size_t num_registers_,
                size_t original_trace_length_)
        : expansion_factor(expansion_factor_), num_colinearity_tests(num_colinearity_tests_),
          security_level(security_level_), num_randomizers(4 * num_colinearity_tests_), num_registers(num_registers_),
          original_trace_length(original_trace_length_),
          omicron_domain_length(Stark::domain_length_for(original_trace_length_ + 4 * num_colinearity_tests_)),
          fri_domain_length(omicron_domain_length * expansion_factor_), generator(FieldElement::generator()),
          omega(FieldElement::primitive_nth_root(fri_domain_length)),
          omicron(FieldElement::primitive_nth_root(omicron_domain_length)),
          fri(FieldElement::generator(), FieldElement::primitive_nth_root(fri_domain_length), fri_domain_length,
              expansion_factor_, num_colinearity_tests_),
          geo(FieldElement::primitive_nth_root(omicron_domain_length), original_trace_length_ + 4 * num_colinearity_tests_)
    {
        if (original_trace_length < 2) throw Panic("a trace needs at least one transition");
    }

    // ---- boundary (:89-131): a handful of points per register, the reference's own algebra
    std::vector<Polynomial> boundary_zeroifiers(const std::vector<BoundaryCondition>& boundary) const
    {
        std::vector<Polynomial> out;
        for (size_t s = 0; s < num_registers; s++) {
            std::vector<FieldElement> points;
            for (const auto& b : boundary)
                if (std::get<1>(b) == s) points.push_back(omicron.pow(std::get<0>(b)));
            out.push_back(Polynomial::zeroifier_domain(points));
        }
        return out;
    }
    std::vector<Polynomial> boundary_interpolants(const std::vector<BoundaryCondition>& boundary) const
    {
        std::vector<Polynomial> out;
        for (size_t s = 0; s < num_registers; s++) {
            std::vector<FieldElement> domain, values;
            for (const auto& b : boundary)
                if (std::get<1>(b) == s) {
                    domain.push_back(omicron.pow(std::get<0>(b)));
                    values.push_back(std::get<2>(b));
                }
            out.push_back(Polynomial::lagrange(domain, values));
        }
        return out;
    }
    // ---- degree bounds (:140-202); a constant column is a polynomial of degree (rows - 1) - 1
    std::vector<size_t> transition_degree_bounds(const std::vector<MPolynomial>& air) const
    {
        const size_t trace_degree = original_trace_length + num_randomizers - 1;
        std::vector<size_t> bounds;
        for (const MPolynomial& a : air) {
            if (a.dict.empty()) throw Panic("called `Option::unwrap()` on a `None` value");
            size_t mx = 0;
            for (const auto& kv : a.dict) {
                size_t sum = 0;
                for (size_t i = 0; i < kv.first.size(); i++) {
                    const size_t d = i == 0 ? 1 : i <= 2 * num_registers ? trace_degree : original_trace_length - 2;
                    sum += d * (size_t)kv.first[i];
                }
                mx = std::max(mx, sum);
            }
            bounds.push_back(mx);
        }
        return bounds;
    }
    std::vector<size_t> transition_quotient_degree_bounds(const std::vector<MPolynomial>& air) const
    {
        std::vector<size_t> b;
        for (size_t d : transition_degree_bounds(air)) b.push_back(d - (original_trace_length - 1));
        return b;
    }
    std::vector<size_t> boundary_quotient_degree_bounds(size_t randomized_trace_length,
                                                        const std::vector<BoundaryCondition>& boundary) const
    {
        std::vector<size_t> b;
        for (const Polynomial& bz : boundary_zeroifiers(boundary)) b.push_back(randomized_trace_length - 1 - bz.degree());
        return b;
    }
    size_t max_degree(const std::vector<MPolynomial>& air) const
    {
        std::vector<size_t> b = transition_quotient_degree_bounds(air);
        size_t md = *std::max_element(b.begin(), b.end());
        if (md == 0) return 0;
        size_t bits = 0;
        while (md >> bits) bits++;
        return ((size_t)1 << bits) - 1;
    }
    std::vector<FieldElement> sample_weights(size_t number, const std::vector<uint8_t>& randomness) const   // :205-220
    {
        std::vector<FieldElement> weights;
        for (size_t i = 0; i < number; i++) {
            hash::Blake2b h(32);
            h.update(randomness);
            uint8_t le[8];
            for (int k = 0; k < 8; k++) le[k] = (uint8_t)((uint64_t)i >> (8 * k));
            h.update(le, 8);
            weights.push_back(FieldElement::sample(h.finalize()));
        }
        return weights;
    }

    // the constant columns (one value per transition each) as coefficient vectors on the device:
    // column k at words(k * (T - 1))
    std::unique_ptr<DeviceVector> constant_polynomials(const std::vector<std::vector<FieldElement>>& columns) const
    {
        const size_t steps = original_trace_length - 1;
        std::unique_ptr<DeviceVector> out(new DeviceVector(std::max<size_t>(1, columns.size() * steps)));
        if (columns.empty()) return out;
        DeviceVector vals(columns.size() * steps);
        for (size_t k = 0; k < columns.size(); k++) {
            if (columns[k].size() != steps) throw Panic("a constant column holds one value per transition");
            vals.upload(columns[k], k * steps);
        }
        geo.interpolate(vals.words(), steps, columns.size(), steps, out->words(), steps);
        return out;
    }

    // What depends on the statement only, made once and handed to every proof and verification of it: the
    // constant columns' polynomials and, from the first proof on, their values on the coset the transition
    // quotients are computed on.  (The reference receives the round-constant polynomials inside its transition
    // constraints, made by rescue_prime.rs:309-359 before Stark::prove is called.)
    struct Prepared {
        std::unique_ptr<DeviceVector> polys;                    // column k at words(k * (T - 1))
        mutable std::unique_ptr<DeviceVector> coset_values;     // column k at words(k * n_q)
        mutable size_t n_q = 0;
    };
    std::unique_ptr<Prepared> prepare(const std::vector<std::vector<FieldElement>>& columns) const
    {
        std::unique_ptr<Prepared> p(new Prepared());
        p->polys = constant_polynomials(columns);
        return p;
    }

    // term table of one constraint for stark_mpoly_eval_dev
    struct TermTable {
        size_t n_vars = 0;
        std::vector<uint64_t> coeffs;
        std::vector<uint8_t> exps;
    };
    static TermTable term_table(const MPolynomial& a, size_t n_vars)
    {
        TermTable t;
        t.n_vars = n_vars;
        for (const auto& kv : a.dict) {
            if (kv.second.is_zero()) continue;
            uint64_t w[2];
            fe_words(kv.second, w);
            t.coeffs.push_back(w[0]);
            t.coeffs.push_back(w[1]);
            for (size_t v = 0; v < n_vars; v++) {
                const u128 ex = v < kv.first.size() ? kv.first[v] : 0;
                if (ex > 255) throw Panic("exponent too large for the device term table");
                t.exps.push_back((uint8_t)ex);
            }
            for (size_t v = n_vars; v < kv.first.size(); v++)
                if (kv.first[v]) throw Panic("constraint uses a variable that has no column");
        }
        return t;
    }

    // ------------------------------------------------------------------------------------------------
    // Stark::prove (:223-471) as the steps it is made of.  What the steps hand to one another lives on the
    // device in a Run.  prove() runs them on one GPU; a prover spread over several GPUs runs the same steps on
    // every rank and only does differently what happens on the FRI domain (low-degree extensions, commitments,
    // FRI, openings), which is why those are not in the steps.
    // ------------------------------------------------------------------------------------------------
    // The trace by columns: register s at cycle c is host[s * stride + c].
    struct TraceColumns {
        const FieldElement* host = nullptr;
        size_t stride = 0;
    };
    struct Run {
        typedef std::chrono::steady_clock Clock;
        std::unique_lock<std::mutex> busy;                  // one proof at a time per ScaledStark (the caches below)
        ProveTimings tm;
        Clock::time_point t_begin, mark;
        size_t R = 0, T = 0, n_rows = 0, n_b = 0, n_q = 0, K = 0, n_vars = 0, max_deg = 0, n_random = 0, n_air = 0;
        std::unique_ptr<DeviceVector> tp, bq, tq, randomizer, combination, made, posted;
        std::unique_ptr<DeviceBuffer> raw_random;
        const DeviceVector* constants = nullptr;
        const Prepared* prepared = nullptr;
        std::vector<size_t> bq_bounds, tq_bounds;
        std::future<void> drawn;
        Run() = default;
        Run(const Run&) = delete;
        ~Run()
        {
            if (drawn.valid()) drawn.wait();                // the helper thread writes into this Run's buffers
        }
        void lap(double& slot)                              // device work is asynchronous: settle it, then charge the lap
        {
            check(stark_ctx_synchronize(context()));
            const auto now = Clock::now();
            slot += std::chrono::duration<double, std::milli>(now - mark).count();
            mark = now;
        }
        // what the random source threw on the helper thread, if anything, as the failure to report
        void surface_draw_error()
        {
            if (!drawn.valid()) return;
            try {
                drawn.get();
            } catch (const std::exception& e) {
                throw Panic(e.what());
            }
        }
    };
    mutable std::mutex prove_mu;

    // a second context on the same GPU: the random bytes go up through it (its own stream and staging lanes)
    // while the proof's kernels run on the first
    static stark_ctx* upload_context()
    {
        static stark_ctx* ctx = [] {
            stark_ctx* c = nullptr;
            int dev = 0;
            const char* e = std::getenv("STARK_DEVICE");
            if (!e || !*e) e = std::getenv("LOCAL_RANK");
            if (e && *e) dev = std::atoi(e);
            check(stark_ctx_create(dev, &c));
            return c;
        }();
        return ctx;
    }

    // :237-288: randomizer rows, trace polynomials, boundary quotients (coefficients in run.bq, column s at
    // s * run.n_b).  `rng` is read as ONE byte stream in the reference's drawing order: 17 bytes per randomizer
    // value row by row (:237-253), then 17 bytes per randomizer coefficient, highest degree first (:344-353); the
    // second part is drawn and sent to the device on a helper thread while the device works, so `rng` and
    // `run` must stay alive until randomizer_polynomial() or the Run's destructor has joined it.
    void begin(Run& run, const TraceColumns& trace, const std::vector<MPolynomial>& air, size_t n_constant_columns,
               const std::vector<BoundaryCondition>& boundary, const RandomSource& rng,
               const Prepared* prepared_constants) const
    {
        stark_ctx* ctx = context();
        run.busy = std::unique_lock<std::mutex>(prove_mu);
        run.t_begin = run.mark = Run::Clock::now();
        const size_t R = run.R = num_registers, T = run.T = original_trace_length;
        const size_t n_rows = run.n_rows = T + num_randomizers;
        run.K = n_constant_columns;
        run.n_vars = 1 + 2 * R + run.K;
        run.n_air = air.size();
        if (run.n_vars > 16) throw Panic("at most 16 variables in a constraint");
        if (!trace.host || trace.stride < T) throw Panic("trace columns shorter than the trace");
        uint64_t gen_w[2];
        fe_words(generator, gen_w);

        // ---- trace columns with their randomizers (:237-253)
        run.tp.reset(new DeviceVector(R * n_rows));
        DeviceVector trace_values(R * n_rows);
        {
            std::vector<uint8_t> bytes(17 * num_randomizers * R);
            rng(bytes.data(), bytes.size());
            std::vector<FieldElement> tail(num_randomizers);
            for (size_t s = 0; s < R; s++) {
                check(stark_copy_h2d(ctx, trace_values.words(s * n_rows), trace.host + s * trace.stride, T * 16));
                for (size_t k = 0; k < num_randomizers; k++) {
                    const uint8_t* b = bytes.data() + 17 * (k * R + s);
                    tail[k] = FieldElement::sample(std::vector<uint8_t>(b, b + 17));
                }
                trace_values.upload(tail, s * n_rows + T);
            }
        }
        // the randomizer polynomial's raw bytes (:344-353) are the next thing the random source yields: tens of
        // megabytes for a long trace, drawn and uploaded (through a context of its own) on a helper thread while
        // the device works on the trace.  The helper starts after the trace has gone up: side by side the two
        // copies fight for the host's memory system and the trace upload takes 7 ms instead of 1.
        run.max_deg = max_degree(air);
        run.n_random = 17 * (run.max_deg + 1);
        if (random_staging.size() < run.n_random) random_staging.resize(run.n_random);
        run.raw_random.reset(new DeviceBuffer(run.n_random));
        {
            uint8_t* staging = random_staging.data();
            void* dst = run.raw_random->p;
            const size_t n_random = run.n_random;
            const RandomSource* source = &rng;
            run.drawn = std::async(std::launch::async, [staging, dst, n_random, source] {
                (*source)(staging, n_random);
                check(stark_copy_h2d(upload_context(), dst, staging, n_random));
            });
        }
        run.lap(run.tm.host);
        // ---- interpolation (:260-275)
        geo.interpolate(trace_values.words(), n_rows, R, n_rows, run.tp->words(), n_rows);
        if (prepared_constants) {
            run.prepared = prepared_constants;
            run.constants = prepared_constants->polys.get();
            if (run.constants->count < std::max<size_t>(1, run.K * (T - 1)))
                throw Panic("prepared constant polynomials do not match the constant columns");
        }
        run.lap(run.tm.interpolate);

        // ---- boundary quotients (:277-288): (trace polynomial - interpolant) / zeroifier on a coset
        std::vector<Polynomial> interpolants = boundary_interpolants(boundary), zeroifiers = boundary_zeroifiers(boundary);
        const size_t n_b = run.n_b = next_pow2(n_rows);
        run.bq.reset(new DeviceVector(R * n_b));           // quotient coefficients, column s at s * n_b
        run.posted.reset(new DeviceVector(R + air.size())); // the degrees, read once at the end (one word each)
        DeviceVector& bq = *run.bq;
        {
            DeviceVector small(R * 2 * 8), den(R * n_b);
            for (size_t s = 0; s < R; s++) {
                if (interpolants[s].coeffs.size() > 8 || zeroifiers[s].coeffs.size() > 8)
                    throw Panic("more boundary conditions on one register than this prover handles");
                std::vector<FieldElement> lo_i(interpolants[s].coeffs.rbegin(), interpolants[s].coeffs.rend());
                std::vector<FieldElement> lo_z(zeroifiers[s].coeffs.rbegin(), zeroifiers[s].coeffs.rend());
                small.upload(lo_i, s * 16);
                small.upload(lo_z, s * 16 + 8);
                const uint64_t* srcs[2] = {run.tp->words(s * n_rows), small.words(s * 16)};
                const size_t lens[2] = {n_rows, lo_i.size()}, shifts[2] = {0, 0};
                uint64_t weights[4] = {1, 0, 0, 0};
                fe_words(-FieldElement::one(), weights + 2);
                check(stark_poly_lincomb_dev(ctx, 2, srcs, lens, shifts, weights, bq.words(s * n_b), n_rows));
                check(stark_coset_lde_batch_dev(ctx, small.words(s * 16 + 8), lo_z.size(), 0, 1, n_b, gen_w,
                                                den.words(s * n_b), n_b));
            }
            DeviceVector num(R * n_b);
            check(stark_coset_lde_batch_dev(ctx, bq.words(), n_rows, n_b, R, n_b, gen_w, num.words(), n_b));
            check(stark_vec_inv_dev(ctx, den.words(), den.words(), R * n_b));
            check(stark_vec_mul_dev(ctx, num.words(), den.words(), bq.words(), R * n_b));
            check(stark_coset_interpolate_batch_dev(ctx, bq.words(), n_b, R, n_b, gen_w));
            for (size_t s = 0; s < R; s++)
                check(stark_poly_degree_post_dev(ctx, bq.words(s * n_b), n_b, run.posted->words(s)));
            check(stark_ctx_synchronize(ctx));                // the work buffers go out of scope
        }
        run.bq_bounds = boundary_quotient_degree_bounds(n_rows, boundary);
        run.tq_bounds = transition_quotient_degree_bounds(air);
        run.lap(run.tm.quotients);
    }

    // :309-341: transition quotients by values on a coset of a subgroup that contains omicron; the composition
    // with omicron * x (:319-325) is the same column read omicron's place count further on.  Coefficients in
    // run.tq, constraint a at a * run.n_q.
    void transition_quotients(Run& run, const std::vector<MPolynomial>& air,
                              const std::vector<std::vector<FieldElement>>& constant_columns) const
    {
        stark_ctx* ctx = context();
        run.mark = Run::Clock::now();                         // what the caller did since the last step is not this step's
        const size_t R = run.R, T = run.T, n_rows = run.n_rows, K = run.K, n_vars = run.n_vars, n_o = omicron_domain_length;
        uint64_t gen_w[2];
        fe_words(generator, gen_w);
        if (!run.constants) {
            run.made = constant_polynomials(constant_columns);
            run.constants = run.made.get();
            run.lap(run.tm.interpolate);
        }
        // the quotients have max_deg + 1 coefficients at most; their values are taken on a coset of that many
        // points (the omicron domain's size in the reference's own configuration), never fewer than the
        // omicron domain so that multiplying x by omicron stays a whole number of places
        const size_t n_q = run.n_q = std::max(n_o, run.max_deg + 1), omicron_places = n_q / n_o;
        run.tq.reset(new DeviceVector(air.size() * n_q));
        DeviceVector& tq = *run.tq;
        {
            // values on the coset that do not depend on the trace are kept from proof to proof: x itself and the
            // transition zeroifier's inverse (they depend on the trace length and the constraints' degree only) in
            // this object, the constant columns' values with the prepared statement
            if (coset_n_q != n_q) {
                coset_x.reset(new DeviceVector(n_q));
                coset_tz_inv.reset(new DeviceVector(n_q));
                DeviceVector x_poly(2);
                x_poly.upload({FieldElement::zero(), FieldElement::one()});
                check(stark_coset_lde_batch_dev(ctx, x_poly.words(), 2, 2, 1, n_q, gen_w, coset_x->words(), n_q));
                if (!tz_coeffs) {
                    tz_coeffs.reset(new DeviceVector(T));
                    tz_coeffs->upload(geo.zeroifier(T - 1));   // :134-137
                }
                check(stark_coset_lde_batch_dev(ctx, tz_coeffs->words(), T, T, 1, n_q, gen_w, coset_tz_inv->words(), n_q));
                check(stark_vec_inv_dev(ctx, coset_tz_inv->words(), coset_tz_inv->words(), n_q));
                check(stark_ctx_synchronize(ctx));
                coset_n_q = n_q;
            }
            std::unique_ptr<DeviceVector> own_constants;
            const DeviceVector* constant_values = nullptr;
            if (K) {
                if (run.prepared) {
                    if (run.prepared->n_q != n_q) {
                        run.prepared->coset_values.reset(new DeviceVector(K * n_q));
                        check(stark_coset_lde_batch_dev(ctx, run.constants->words(), T - 1, T - 1, K, n_q, gen_w,
                                                        run.prepared->coset_values->words(), n_q));
                        run.prepared->n_q = n_q;
                    }
                    constant_values = run.prepared->coset_values.get();
                } else {
                    own_constants.reset(new DeviceVector(K * n_q));
                    check(stark_coset_lde_batch_dev(ctx, run.constants->words(), T - 1, T - 1, K, n_q, gen_w,
                                                    own_constants->words(), n_q));
                    constant_values = own_constants.get();
                }
            }
            DeviceVector cols(R * n_q);
            check(stark_coset_lde_batch_dev(ctx, run.tp->words(), n_rows, n_rows, R, n_q, gen_w, cols.words(), n_q));
            const DeviceVector& tz_values = *coset_tz_inv;
            // variables: x, registers now, registers one step on (rotation), constants
            std::vector<const uint64_t*> ptrs(n_vars);
            std::vector<size_t> rot(n_vars, 0);
            ptrs[0] = coset_x->words();
            for (size_t s = 0; s < R; s++) {
                ptrs[1 + s] = cols.words(s * n_q);
                ptrs[1 + R + s] = cols.words(s * n_q);
                rot[1 + R + s] = omicron_places;
            }
            for (size_t k = 0; k < K; k++) ptrs[1 + 2 * R + k] = constant_values->words(k * n_q);
            for (size_t a = 0; a < air.size(); a++) {
                TermTable tt = term_table(air[a], n_vars);
                check(stark_mpoly_eval_dev(ctx, n_vars, ptrs.data(), rot.data(), tt.coeffs.size() / 2, tt.coeffs.data(),
                                           tt.exps.data(), tz_values.words(), tq.words(a * n_q), n_q));
            }
            check(stark_coset_interpolate_batch_dev(ctx, tq.words(), n_q, air.size(), n_q, gen_w));
            for (size_t a = 0; a < air.size(); a++)            // :379-380, read with the other degrees at the end
                check(stark_poly_degree_post_dev(ctx, tq.words(a * n_q), n_q, run.posted->words(R + a)));
            check(stark_ctx_synchronize(ctx));
        }
        run.lap(run.tm.quotients);
    }

    // :344-353: the randomizer polynomial's coefficients (lowest first) in run.randomizer
    void randomizer_polynomial(Run& run) const
    {
        stark_ctx* ctx = context();
        run.mark = Run::Clock::now();
        run.randomizer.reset(new DeviceVector(run.max_deg + 1));
        run.drawn.get();                                      // rethrows what the random source threw
        check(stark_sample_bytes_dev(ctx, (const uint8_t*)run.raw_random->p, 17, run.max_deg + 1, 1, run.randomizer->words()));
        run.lap(run.tm.host);
    }

    // :372-404: the weights from the transcript's challenge and the weighted sum, coefficients in run.combination
    void combine(Run& run, const std::vector<uint8_t>& challenge) const
    {
        stark_ctx* ctx = context();
        run.mark = Run::Clock::now();
        const size_t R = run.R, n_air = run.n_air, max_deg = run.max_deg;
        std::vector<FieldElement> weights = sample_weights(1 + 2 * n_air + 2 * R, challenge);
        run.combination.reset(new DeviceVector(max_deg + 1));
        DeviceVector& combination = *run.combination;
        std::vector<const uint64_t*> srcs;
        std::vector<size_t> lens, shifts;
        auto term = [&](const uint64_t* p, size_t len, size_t shift) {
            srcs.push_back(p);
            lens.push_back(len);
            shifts.push_back(shift);
        };
        term(run.randomizer->words(), max_deg + 1, 0);
        for (size_t a = 0; a < n_air; a++) {
            term(run.tq->words(a * run.n_q), run.tq_bounds[a] + 1, 0);
            term(run.tq->words(a * run.n_q), run.tq_bounds[a] + 1, max_deg - run.tq_bounds[a]);
        }
        for (size_t s = 0; s < R; s++) {
            term(run.bq->words(s * run.n_b), run.bq_bounds[s] + 1, 0);
            term(run.bq->words(s * run.n_b), run.bq_bounds[s] + 1, max_deg - run.bq_bounds[s]);
        }
        std::vector<uint64_t> w(2 * weights.size());
        for (size_t i = 0; i < weights.size(); i++) fe_words(weights[i], &w[2 * i]);
        // at most 16 terms per pass: the first pass writes, later passes add to what is there
        for (size_t at = 0; at < srcs.size();) {
            std::vector<const uint64_t*> s2;
            std::vector<size_t> l2, h2;
            std::vector<uint64_t> w2;
            if (at) {
                s2.push_back(combination.words());
                l2.push_back(max_deg + 1);
                h2.push_back(0);
                w2.push_back(1);
                w2.push_back(0);
            }
            while (at < srcs.size() && s2.size() < 16) {
                s2.push_back(srcs[at]);
                l2.push_back(lens[at]);
                h2.push_back(shifts[at]);
                w2.push_back(w[2 * at]);
                w2.push_back(w[2 * at + 1]);
                at++;
            }
            check(stark_poly_lincomb_dev(ctx, s2.size(), s2.data(), l2.data(), h2.data(), w2.data(), combination.words(),
                                         max_deg + 1));
        }
        run.lap(run.tm.quotients);
    }

    // the degree assertions of the steps above (:379-380 and the exactness of the boundary division), from the
    // words the steps posted on the device: one read
    void check_degrees(Run& run) const
    {
        std::vector<FieldElement> words = run.posted->download(0, run.R + run.n_air);
        auto degree = [&](size_t i) {
            const uint64_t r = (uint64_t)words[i].value;
            return r ? (size_t)(r - 1) : (size_t)0;             // univariate_poly.rs:69-85: the zero polynomial has degree 0
        };
        for (size_t s = 0; s < run.R; s++)
            if (degree(s) > run.bq_bounds[s]) throw Panic("the trace does not meet its boundary conditions");
        for (size_t a = 0; a < run.n_air; a++)
            if (degree(run.R + a) != run.tq_bounds[a])
                throw Panic("assertion failed: tq_degrees == self.transition_quotient_degree_bounds(transition_constraints.clone())");
    }

    // Stark::prove (:223-471).  trace[c][s] is register s at cycle c; constant_columns[k][c] is constant k of
    // the transition from cycle c to c + 1; the AIR's variables are [x, registers at c, registers at c + 1,
    // constants].  `prepared`, if given, is constant_polynomials(constant_columns) made beforehand: the columns
    // are public parameters of the statement (the reference receives the round-constant polynomials inside its
    // transition constraints, made by rescue_prime.rs:309-359 before Stark::prove is called).
    // One proof at a time per ScaledStark object (a second caller waits).
    std::vector<uint8_t> prove(const std::vector<std::vector<FieldElement>>& trace, const std::vector<MPolynomial>& air,
                               const std::vector<std::vector<FieldElement>>& constant_columns,
                               const std::vector<BoundaryCondition>& boundary, const RandomSource& rng,
                               ProveTimings* timings = nullptr, const Prepared* prepared = nullptr) const
    {
        const auto t0 = std::chrono::steady_clock::now();
        if (trace.size() != original_trace_length) throw Panic("trace length differs from the one this Stark was made for");
        const size_t R = num_registers, T = original_trace_length;
        // rows to columns: every row is a heap object of its own, so this is a walk through memory; a few threads
        // take a range of cycles each.  (The staging vector is kept from proof to proof: a fresh one costs a page
        // fault per 4 KiB written, more than the copy itself at 2^20 rows.)
        std::unique_lock<std::mutex> staging(staging_mu);
        if (column_staging.size() != R * T) column_staging.assign(R * T, FieldElement::zero());
        {
            const size_t workers = T >= ((size_t)1 << 16) ? 8 : 1;
            std::vector<std::thread> pool;
            std::vector<int> bad(workers, 0);
            for (size_t w = 0; w < workers; w++)
                pool.emplace_back([&, w] {
                    for (size_t c = T * w / workers; c < T * (w + 1) / workers; c++) {
                        if (trace[c].size() != R) {
                            bad[w] = 1;
                            return;
                        }
                        for (size_t s = 0; s < R; s++) column_staging[s * T + c] = trace[c][s];
                    }
                });
            for (auto& th : pool) th.join();
            for (int b : bad)
                if (b) throw Panic("trace row has the wrong number of registers");
        }
        const double transposed = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::vector<uint8_t> proof =
            prove_columns(TraceColumns{column_staging.data(), T}, air, constant_columns, boundary, rng, timings, prepared);
        if (timings) {
            timings->host += transposed;
            timings->total += transposed;
        }
        return proof;
    }

    // The same with the trace handed over by columns (no transposition, no staging copy).
    std::vector<uint8_t> prove_columns(const TraceColumns& trace, const std::vector<MPolynomial>& air,
                                       const std::vector<std::vector<FieldElement>>& constant_columns,
                                       const std::vector<BoundaryCondition>& boundary, const RandomSource& rng,
                                       ProveTimings* timings = nullptr, const Prepared* prepared = nullptr) const
    {
        Run run;
        try {
            return prove_steps(run, trace, air, constant_columns, boundary, rng, timings, prepared);
        } catch (...) {
            run.surface_draw_error();                         // a failure of the random source is the one to report
            throw;
        }
    }

    std::vector<uint8_t> prove_steps(Run& run, const TraceColumns& trace, const std::vector<MPolynomial>& air,
                                     const std::vector<std::vector<FieldElement>>& constant_columns,
                                     const std::vector<BoundaryCondition>& boundary, const RandomSource& rng,
                                     ProveTimings* timings, const Prepared* prepared) const
    {
        stark_ctx* ctx = context();
        const size_t R = num_registers, n_fri = fri_domain_length;
        uint64_t gen_w[2];
        fe_words(generator, gen_w);
        ProofStream ps;
        begin(run, trace, air, constant_columns.size(), boundary, rng, prepared);

        // ---- commit to the boundary quotients on the FRI domain (:291-306)
        DeviceVector bq_codewords(R * n_fri);
        check(stark_coset_lde_batch_dev(ctx, run.bq->words(), run.n_rows, run.n_b, R, n_fri, gen_w, bq_codewords.words(), n_fri));
        run.lap(run.tm.lde);
        std::vector<std::unique_ptr<TreeHandle>> bq_trees;
        for (size_t s = 0; s < R; s++) {
            bq_trees.emplace_back(new TreeHandle());
            HashOutput root;
            check(stark_merkle_commit_fe_dev(ctx, bq_codewords.words(s * n_fri), n_fri, root.data(), &bq_trees.back()->t));
            ps.push(wire::hex_encode(root.data(), 32));
        }
        run.lap(run.tm.merkle);

        transition_quotients(run, air, constant_columns);
        randomizer_polynomial(run);
        // ---- the randomizer's codeword and commitment (:356-360)
        DeviceVector rz_codeword(n_fri);
        check(stark_coset_lde_batch_dev(ctx, run.randomizer->words(), run.max_deg + 1, run.max_deg + 1, 1, n_fri, gen_w,
                                        rz_codeword.words(), n_fri));
        run.lap(run.tm.lde);
        TreeHandle rz_tree;
        {
            HashOutput root;
            check(stark_merkle_commit_fe_dev(ctx, rz_codeword.words(), n_fri, root.data(), &rz_tree.t));
            ps.push(wire::hex_encode(root.data(), 32));
        }
        run.lap(run.tm.merkle);

        combine(run, ps.prover_fiat_shamir(32));
        DeviceVector comb_codeword(n_fri);
        check(stark_coset_lde_batch_dev(ctx, run.combination->words(), run.max_deg + 1, run.max_deg + 1, 1, n_fri, gen_w,
                                        comb_codeword.words(), n_fri));                         // :406
        run.lap(run.tm.lde);
        std::vector<size_t> indices = fri.prove_resident(comb_codeword.words(), ps);            // :409-410
        std::sort(indices.begin(), indices.end());
        run.lap(run.tm.fri);

        // ---- openings (:415-464)
        std::vector<size_t> duplicated = indices;
        for (size_t i : indices) duplicated.push_back((i + expansion_factor) % n_fri);
        std::sort(duplicated.begin(), duplicated.end());
        const std::vector<uint64_t> dup64(duplicated.begin(), duplicated.end()), idx64(indices.begin(), indices.end());
        auto elements = [&](const uint64_t* codeword, const std::vector<uint64_t>& at) {
            std::vector<FieldElement> v(at.size());
            check(stark_vec_read_dev(ctx, codeword, n_fri, at.data(), at.size(), (uint64_t*)v.data()));
            return v;
        };
        for (size_t s = 0; s < R; s++) {
            const std::vector<std::vector<HashOutput>> paths = Merkle::open_tree_many(bq_trees[s]->t, dup64);
            const std::vector<FieldElement> leaves = elements(bq_codewords.words(s * n_fri), dup64);
            for (size_t k = 0; k < duplicated.size(); k++) {
                ps.push(wire::json_field_element(leaves[k]));
                ps.push(wire::json_hash_path(paths[k]));
            }
        }
        const std::vector<std::vector<HashOutput>> rz_paths = Merkle::open_tree_many(rz_tree.t, idx64);
        const std::vector<FieldElement> rz_leaves = elements(rz_codeword.words(), idx64);
        for (size_t k = 0; k < indices.size(); k++) {
            ps.push(wire::json_field_element(rz_leaves[k]));
            ps.push(wire::json_hash_path(rz_paths[k]));
        }
        check_degrees(run);
        std::vector<uint8_t> proof = ps.serialize();                                            // :470
        run.lap(run.tm.openings);
        run.tm.total = std::chrono::duration<double, std::milli>(Run::Clock::now() - run.t_begin).count();
        if (timings) *timings = run.tm;
        return proof;
    }

    // Stark::verify (:474-723) with the constant columns' polynomials evaluated at the opened points
    bool verify(const std::vector<uint8_t>& proof, const std::vector<MPolynomial>& air,
                const std::vector<std::vector<FieldElement>>& constant_columns,
                const std::vector<BoundaryCondition>& boundary, const Prepared* prepared = nullptr) const
    {
        size_t max_cycle = 0;
        for (const auto& b : boundary) max_cycle = std::max(max_cycle, std::get<0>(b));
        const size_t trace_length = 1 + max_cycle, randomized_trace_length = trace_length + num_randomizers;
        if (trace_length != original_trace_length) throw Panic("boundary conditions do not fit this Stark's trace length");
        ProofStream ps = ProofStream::deserialize(proof);
        std::vector<std::string> bq_roots;
        for (size_t s = 0; s < num_registers; s++) bq_roots.push_back(ps.pull());
        std::string randomizer_root = ps.pull();
        std::vector<Polynomial> interpolants = boundary_interpolants(boundary), zeroifiers = boundary_zeroifiers(boundary);
        std::vector<FieldElement> weights =
            sample_weights(1 + 2 * air.size() + 2 * interpolants.size(), ps.verifier_fiat_shamir(32));
        std::vector<std::pair<size_t, FieldElement>> polynomial_values;
        if (!fri.verify(ps, polynomial_values)) return false;
        std::stable_sort(polynomial_values.begin(), polynomial_values.end(),
                         [](const std::pair<size_t, FieldElement>& a, const std::pair<size_t, FieldElement>& b) {
                             return a.first < b.first;
                         });
        std::vector<size_t> indices, duplicated;
        for (const auto& iv : polynomial_values) indices.push_back(iv.first);
        duplicated = indices;
        for (size_t i : indices) duplicated.push_back((i + expansion_factor) % fri.domain_length);
        std::sort(duplicated.begin(), duplicated.end());
        auto root_of = [](const std::string& hex) {
            std::vector<uint8_t> b = wire::hex_decode(hex);
            HashOutput r{};
            if (b.size() != 32) throw Panic("invalid root length");
            std::memcpy(r.data(), b.data(), 32);
            return r;
        };
        std::vector<std::map<size_t, FieldElement>> leafs;                                   // :531-559
        for (size_t r = 0; r < bq_roots.size(); r++) {
            std::map<size_t, FieldElement> m;
            for (size_t i : duplicated) {
                FieldElement leaf = wire::json_field_element_parse(ps.pull());
                m[i] = leaf;
                std::vector<HashOutput> path = wire::json_hash_path_parse(ps.pull());
                if (!Merkle::verify(root_of(bq_roots[r]), i, path, wire::bincode_field_element(leaf))) return false;
            }
            leafs.push_back(m);
        }
        std::map<size_t, FieldElement> randomizer;                                           // :562-586
        for (size_t i : indices) {
            FieldElement leaf = wire::json_field_element_parse(ps.pull());
            randomizer[i] = leaf;
            std::vector<HashOutput> path = wire::json_hash_path_parse(ps.pull());
            if (!Merkle::verify(root_of(randomizer_root), i, path, wire::bincode_field_element(leaf))) return false;
        }
        const size_t steps = original_trace_length - 1, K = constant_columns.size();
        const size_t max_deg = max_degree(air);
        std::vector<size_t> tq_bounds = transition_quotient_degree_bounds(air);
        std::vector<size_t> bq_bounds = boundary_quotient_degree_bounds(randomized_trace_length, boundary);
        const starkf::HostField& F = host_field();
        // The long evaluations.  At every opened point: the K constant columns' polynomials (T - 1
        // coefficients each), interpolated and evaluated on the device, from which only the values come back;
        // and the transition zeroifier as the product it is (:666, one factor per trace point), cut into runs
        // of consecutive factors that a few threads multiply up side by side.
        std::vector<FieldElement> xs;
        for (size_t idx : indices) xs.push_back(generator * omega.pow(idx));
        std::vector<FieldElement> constant_values(indices.size() * K), tz_values(indices.size());
        if (K && !indices.empty()) {
            std::unique_ptr<DeviceVector> made;
            if (!prepared) made = constant_polynomials(constant_columns);
            const DeviceVector* polys = prepared ? prepared->polys.get() : made.get();
            if (polys->count < K * steps) throw Panic("prepared constant polynomials do not match the constant columns");
            // one evaluation call takes at most 16 points (2 * num_colinearity_tests are opened)
            for (size_t at = 0; at < xs.size(); at += 16) {
                const size_t m = std::min<size_t>(16, xs.size() - at);
                std::vector<uint64_t> pts(2 * m), vals(2 * K * m);
                for (size_t i = 0; i < m; i++) fe_words(xs[at + i], &pts[2 * i]);
                check(stark_poly_eval_points_dev(context(), polys->words(), steps, K, steps, pts.data(), m, vals.data()));
                for (size_t k = 0; k < K; k++)
                    for (size_t i = 0; i < m; i++) {
                        starkf::fe v{vals[2 * (k * m + i)], vals[2 * (k * m + i) + 1]};
                        constant_values[(at + i) * K + k] = FieldElement::wrap(v);
                    }
            }
        }
        {
            const size_t factors = original_trace_length - 1;
            const size_t runs = factors >= ((size_t)1 << 14) ? 8 : 1, per = (factors + runs - 1) / runs;
            const size_t tasks = indices.size() * runs;
            std::vector<starkf::fe> part(tasks, F.one_m);
            std::atomic<size_t> next{0};
            const starkf::fe om = F.to_mont(omicron.raw());
            auto work = [&] {
                for (size_t task = next++; task < tasks; task = next++) {
                    const size_t i = task / runs, lo = (task % runs) * per, hi = std::min(factors, lo + per);
                    const starkf::fe xm = F.to_mont(xs[i].raw());
                    starkf::fe tz = F.one_m, w = F.to_mont(omicron.pow(lo).raw());
                    for (size_t j = lo; j < hi; j++) {
                        tz = starkf::mont_mul(tz, starkf::fe_sub(xm, w));
                        w = starkf::mont_mul(w, om);
                    }
                    part[task] = tz;
                }
            };
            std::vector<std::thread> pool;
            for (size_t w = 1; w < std::min<size_t>(8, tasks); w++) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
            for (size_t i = 0; i < indices.size(); i++) {
                starkf::fe tz = F.one_m;
                for (size_t r = 0; r < runs; r++) tz = starkf::mont_mul(tz, part[i * runs + r]);
                tz_values[i] = FieldElement::wrap(F.from_mont(tz));
            }
        }
        for (size_t i = 0; i < indices.size(); i++) {                                        // :589-719
            size_t current_index = indices[i];
            FieldElement xcur = xs[i];
            size_t next_index = (current_index + expansion_factor) % fri.domain_length;
            FieldElement xnext = xcur * omicron;
            std::vector<FieldElement> point{xcur}, next_trace;
            for (size_t s = 0; s < num_registers; s++) {
                point.push_back(leafs[s].at(current_index) * zeroifiers[s].eval(xcur) + interpolants[s].eval(xcur));
                next_trace.push_back(leafs[s].at(next_index) * zeroifiers[s].eval(xnext) + interpolants[s].eval(xnext));
            }
            point.insert(point.end(), next_trace.begin(), next_trace.end());
            for (size_t k = 0; k < K; k++) point.push_back(constant_values[i * K + k]);
            FieldElement tz_inv = tz_values[i].inverse();
            std::vector<FieldElement> terms{randomizer.at(current_index)};
            for (size_t s = 0; s < air.size(); s++) {
                FieldElement qv = air[s].eval(point) * tz_inv;
                terms.push_back(qv);
                terms.push_back(qv * xcur.pow(max_deg - tq_bounds[s]));
            }
            for (size_t s = 0; s < num_registers; s++) {
                FieldElement bqv = leafs[s].at(current_index);
                terms.push_back(bqv);
                terms.push_back(bqv * xcur.pow(max_deg - bq_bounds[s]));
            }
            FieldElement combination = FieldElement::zero();
            for (size_t j = 0; j < terms.size(); j++) combination = combination + terms[j] * weights.at(j);
            if (!(combination == polynomial_values[i].second)) return false;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------
// the workload: Rescue-Prime rounds chained over a long trace
// ---------------------------------------------------------------------------
struct RescueChain {
    RescuePrime rp;
    size_t rows;
    explicit RescueChain(size_t rows_) : rows(rows_) {}

    // RescuePrime::trace (rescue_prime.rs:230-293) continued past N rounds: row r + 1 is the round function on
    // row r with the constants of round r mod N.  rows = N + 1 is the reference's trace.
    std::vector<std::vector<FieldElement>> trace(const FieldElement& input_element) const
    {
        // the round function on Montgomery forms: one product per multiplication
        using namespace starkf;
        const HostField& F = host_field();
        const size_t m = rp.m;
        std::vector<fe> mds(m * m), rc(rp.round_constants.size());
        for (size_t i = 0; i < m; i++)
            for (size_t j = 0; j < m; j++) mds[i * m + j] = F.to_mont(rp.MDS[i][j].raw());
        for (size_t i = 0; i < rc.size(); i++) rc[i] = F.to_mont(rp.round_constants[i].raw());
        auto pow_m = [&](fe b, u128 e) {
            fe r = F.one_m;
            int top = 127;
            while (top > 0 && !((e >> top) & 1)) top--;
            for (int i = top; i >= 0; i--) {
                r = mont_mul(r, r);
                if ((e >> i) & 1) r = mont_mul(r, b);
            }
            return r;
        };
        std::vector<fe> state(m, fe{0, 0}), tmp(m);
        state[0] = F.to_mont(input_element.raw());
        std::vector<std::vector<FieldElement>> out;
        out.reserve(rows);
        auto emit = [&]() {
            std::vector<FieldElement> row(m);
            for (size_t i = 0; i < m; i++) row[i] = FieldElement::wrap(F.from_mont(state[i]));
            out.push_back(row);
        };
        emit();
        for (size_t r = 0; r + 1 < rows; r++) {
            const size_t rr = r % rp.N;
            for (int half = 0; half < 2; half++) {
                for (size_t i = 0; i < m; i++) state[i] = half == 0 ? mont_mul(mont_mul(state[i], state[i]), state[i])
                                                                     : pow_m(state[i], rp.alpha_inv);
                for (size_t i = 0; i < m; i++) {
                    fe acc{0, 0};
                    for (size_t j = 0; j < m; j++) acc = fe_add(acc, mont_mul(mds[i * m + j], state[j]));
                    tmp[i] = fe_add(acc, rc[2 * rr * m + half * m + i]);
                }
                state = tmp;
            }
            emit();
        }
        return out;
    }
    std::vector<BoundaryCondition> boundary_constraints(const FieldElement& output_element) const   // rescue_prime.rs:296-306
    {
        return {BoundaryCondition{0, 1, FieldElement::zero()}, BoundaryCondition{rows - 1, 0, output_element}};
    }
    // first-step constants of every register, then second-step constants, one value per transition
    // (the values round_constants_polynomials interpolates, rescue_prime.rs:309-359)
    std::vector<std::vector<FieldElement>> constant_columns() const
    {
        const size_t m = rp.m;
        std::vector<std::vector<FieldElement>> cols(2 * m, std::vector<FieldElement>(rows - 1));
        for (size_t r = 0; r + 1 < rows; r++)
            for (size_t i = 0; i < m; i++) {
                cols[i][r] = rp.round_constants[2 * (r % rp.N) * m + i];
                cols[m + i][r] = rp.round_constants[2 * (r % rp.N) * m + m + i];
            }
        return cols;
    }
    // transition_constraints (rescue_prime.rs:363-394) over [x, previous state, next state, first-step
    // constants, second-step constants]
    std::vector<MPolynomial> air() const
    {
        const size_t m = rp.m;
        std::vector<MPolynomial> v = MPolynomial::variables(1 + 4 * m);
        std::vector<MPolynomial> out;
        for (size_t i = 0; i < m; i++) {
            MPolynomial lhs = MPolynomial::constant(0);
            for (size_t k = 0; k < m; k++) lhs = lhs + MPolynomial::constant(rp.MDS[i][k].value) * v[1 + k].pow(rp.alpha);
            lhs = lhs + v[1 + 2 * m + i];
            MPolynomial rhs = MPolynomial::constant(0);
            for (size_t k = 0; k < m; k++)
                rhs = rhs + MPolynomial::constant(rp.MDS_inv[i][k].value) * (v[1 + m + k] - v[1 + 3 * m + k]);
            out.push_back(lhs - rhs.pow(rp.alpha));
        }
        return out;
    }
};

}  // namespace stark
