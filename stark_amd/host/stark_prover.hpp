// The caller of the hot path, mirrored above the C ABI: multivariate constraint
// polynomials (multivariate_poly.rs), the Rescue-Prime AIR (rescue_prime.rs)
// and Stark::prove / Stark::verify (stark.rs).  The polynomial algebra stays on
// the host as in the reference; every low-degree extension, Merkle commitment,
// FRI commitment and opening of the prover runs on the device, and the
// codewords and their trees stay resident between the commitment and the
// openings (the reference re-serialises and re-hashes a whole codeword for
// every opening, stark.rs:436-438,458-460).
#pragma once
#include <algorithm>
#include <functional>
#include <map>
#include <random>
#include <tuple>

#include "stark_host.hpp"

namespace stark {

// ---------------------------------------------------------------------------
// multivariate_poly.rs
// ---------------------------------------------------------------------------
// Dictionary from exponent vectors to coefficients (:24-26).  Keys keep the
// length they were built with and cancelled terms stay stored, as in the
// reference: the degree bounds of stark.rs:140-168 run over every stored term.
struct MPolynomial {
    typedef std::vector<u128> Key;
    std::map<Key, FieldElement> dict;

    static MPolynomial zero() { return MPolynomial{}; }
    static MPolynomial constant(u128 element)                                               // :33-35
    {
        MPolynomial m;
        m.dict[Key{0}] = FieldElement::from(element);
        return m;
    }
    bool is_zero() const                                                                    // :38
    {
        for (const auto& kv : dict)
            if (!kv.second.is_zero()) return false;
        return true;
    }
    static std::vector<MPolynomial> variables(size_t num_variables)                         // :113-130
    {
        std::vector<MPolynomial> out;
        for (size_t i = 0; i < num_variables; i++) {
            Key e(num_variables, 0);
            e[i] = 1;
            MPolynomial m;
            m.dict[e] = FieldElement::one();
            out.push_back(m);
        }
        return out;
    }
    static MPolynomial lift(const Polynomial& poly, size_t variable_index)                  // :133-146
    {
        if (poly.is_zero()) return zero();
        MPolynomial x = variables(variable_index + 1)[variable_index];
        MPolynomial acc = zero();
        size_t i = 0;
        for (auto it = poly.coeffs.rbegin(); it != poly.coeffs.rend(); ++it, ++i)
            acc = acc + constant(it->value) * x.pow(i);
        return acc;
    }
    MPolynomial operator+(const MPolynomial& rhs) const                                     // :150-172
    {
        MPolynomial r = *this;
        for (const auto& kv : rhs.dict) {
            auto it = r.dict.find(kv.first);
            if (it != r.dict.end()) it->second = it->second + kv.second;
            else r.dict.insert(kv);
        }
        return r;
    }
    MPolynomial operator-() const                                                           // :184-197
    {
        MPolynomial r = *this;
        for (auto& kv : r.dict) kv.second = -kv.second;
        return r;
    }
    MPolynomial operator-(const MPolynomial& rhs) const { return *this + (-rhs); }          // :175-181
    MPolynomial operator*(const MPolynomial& rhs) const                                     // :200-237
    {
        MPolynomial r;
        for (const auto& l : dict)
            for (const auto& q : rhs.dict) {
                Key e = l.first;
                for (size_t idx = 0; idx < q.first.size(); idx++) {
                    if (idx < e.size()) e[idx] += q.first[idx];
                    else e.push_back(q.first[idx]);
                }
                FieldElement c = l.second * q.second;
                auto it = r.dict.find(e);
                if (it != r.dict.end()) it->second = it->second + c;
                else r.dict[e] = c;
            }
        return r;
    }
    // :91-108.  The reference starts from a key as long as whichever key its hash
    // map yields first; that choice only pads keys with zeros and no value
    // computed from the dictionary depends on it.  Here: the first key in order.
    MPolynomial pow(u128 exponent) const
    {
        if (is_zero()) return zero();
        if (exponent == 0) return constant(1);
        MPolynomial acc;
        acc.dict[Key(dict.begin()->first.size(), 0)] = FieldElement::one();
        int top = 127;
        while (!((exponent >> top) & 1)) top--;
        for (int i = top; i >= 0; i--) {
            acc = acc * acc;
            if ((exponent >> i) & 1) acc = acc * *this;
        }
        return acc;
    }
    FieldElement eval(const std::vector<FieldElement>& point) const                         // :42-72
    {
        FieldElement acc = FieldElement::zero();
        for (const auto& kv : dict) {
            FieldElement prod = kv.second;
            for (size_t i = 0; i < kv.first.size(); i++) {
                if (i >= point.size()) continue;
                if (kv.first[i]) prod = prod * point[i].pow(kv.first[i]);
            }
            acc = acc + prod;
        }
        return acc;
    }
    Polynomial eval_symbolic(const std::vector<Polynomial>& point) const                    // :75-88
    {
        Polynomial acc{{FieldElement::zero()}};
        for (const auto& kv : dict) {
            Polynomial prod{{kv.second}};
            for (size_t i = 0; i < kv.first.size(); i++) prod = prod * point.at(i).pow(kv.first[i]);
            acc = acc + prod;
        }
        return acc;
    }
};

// ---------------------------------------------------------------------------
// rescue_prime.rs
// ---------------------------------------------------------------------------
typedef std::tuple<size_t, size_t, FieldElement> BoundaryCondition;      // (cycle, register, value)

struct RescuePrime {
    size_t m = 2, rate = 1, capacity = 1, N = 27;                         // :30-33
    u128 alpha = 3, alpha_inv;
    FieldElement MDS[2][2], MDS_inv[2][2];
    std::vector<FieldElement> round_constants;

    // Parameters of :27-169.  The 108 round constants are those of the
    // Rescue-Prime specification: SHAKE256("Rescue-XLIX(p,m,capacity,128)") cut
    // into 17-byte little-endian integers, reduced mod p; alpha^-1 is the
    // inverse of 3 modulo p-1 and MDS^-1 the inverse of [[-3,4],[-12,13]].
    RescuePrime()
    {
        const u128 p = FieldElement::modulus();
        // 3 * alpha_inv = 1 + 2 (p - 1): with p - 1 = 3 q + 1, alpha_inv = 2 q + 1
        alpha_inv = 2 * ((p - 2) / 3) + 1;
        if (!(FieldElement::from(5).pow(alpha).pow(alpha_inv) == FieldElement::from(5))) throw Panic("alpha inverse");
        MDS[0][0] = FieldElement::from(p - 3);  MDS[0][1] = FieldElement::from(4);
        MDS[1][0] = FieldElement::from(p - 12); MDS[1][1] = FieldElement::from(13);
        FieldElement inv9 = FieldElement::from(9).inverse();
        MDS_inv[0][0] = FieldElement::from(13) * inv9;     MDS_inv[0][1] = FieldElement::from(p - 4) * inv9;
        MDS_inv[1][0] = FieldElement::from(12) * inv9;     MDS_inv[1][1] = FieldElement::from(p - 3) * inv9;
        const size_t bpi = (128 + 7) / 8 + 1;                              // bytes per constant: ceil(bits(p) / 8) + 1
        const size_t n_const = 2 * m * N;
        std::string seed = "Rescue-XLIX(" + FieldElement{p}.to_string() + "," + std::to_string(m) + "," +
                           std::to_string(capacity) + ",128)";
        std::vector<uint8_t> stream = hash::shake256((const uint8_t*)seed.data(), seed.size(), bpi * n_const);
        for (size_t i = 0; i < n_const; i++) {
            // little-endian integer of bpi bytes, reduced mod p
            FieldElement acc = FieldElement::zero();
            for (size_t j = bpi; j-- > 0;) acc = acc * FieldElement::from(256) + FieldElement::from(stream[bpi * i + j]);
            round_constants.push_back(acc);
        }
    }

    std::vector<FieldElement> round(std::vector<FieldElement> state, size_t r) const         // :182-222
    {
        for (size_t i = 0; i < m; i++) state[i] = state[i].pow(alpha);
        std::vector<FieldElement> temp(m, FieldElement::zero());
        for (size_t i = 0; i < m; i++)
            for (size_t j = 0; j < m; j++) temp[i] = temp[i] + MDS[i][j] * state[j];
        for (size_t i = 0; i < m; i++) state[i] = temp[i] + round_constants[2 * r * m + i];
        for (size_t i = 0; i < m; i++) state[i] = state[i].pow(alpha_inv);
        temp.assign(m, FieldElement::zero());
        for (size_t i = 0; i < m; i++)
            for (size_t j = 0; j < m; j++) temp[i] = temp[i] + MDS[i][j] * state[j];
        for (size_t i = 0; i < m; i++) state[i] = temp[i] + round_constants[2 * r * m + m + i];
        return state;
    }
    FieldElement hash(const FieldElement& input_element) const                               // :172-227
    {
        std::vector<FieldElement> state(m, FieldElement::zero());
        state[0] = input_element;
        for (size_t r = 0; r < N; r++) state = round(state, r);
        return state[0];
    }
    std::vector<std::vector<FieldElement>> trace(const FieldElement& input_element) const   // :230-293
    {
        std::vector<FieldElement> state(m, FieldElement::zero());
        state[0] = input_element;
        std::vector<std::vector<FieldElement>> out{state};
        for (size_t r = 0; r < N; r++) {
            state = round(state, r);
            out.push_back(state);
        }
        return out;
    }
    std::vector<BoundaryCondition> boundary_constraints(const FieldElement& output_element) const   // :296-306
    {
        return {BoundaryCondition{0, 1, FieldElement::zero()}, BoundaryCondition{N, 0, output_element}};
    }
    void round_constants_polynomials(const FieldElement& omicron, std::vector<MPolynomial>& first,
                                     std::vector<MPolynomial>& second) const                 // :309-359
    {
        std::vector<FieldElement> domain;
        for (size_t r = 0; r < N; r++) domain.push_back(omicron.pow(r));
        for (size_t i = 0; i < m; i++) {
            std::vector<FieldElement> values;
            for (size_t r = 0; r < N; r++) values.push_back(round_constants[2 * r * m + i]);
            first.push_back(MPolynomial::lift(Polynomial::lagrange(domain, values), 0));
        }
        for (size_t i = 0; i < m; i++) {
            std::vector<FieldElement> values;
            for (size_t r = 0; r < N; r++) values.push_back(round_constants[2 * r * m + m + i]);
            second.push_back(MPolynomial::lift(Polynomial::lagrange(domain, values), 0));
        }
    }
    std::vector<MPolynomial> transition_constraints(const FieldElement& omicron) const      // :363-394
    {
        std::vector<MPolynomial> first, second;
        round_constants_polynomials(omicron, first, second);
        std::vector<MPolynomial> variables = MPolynomial::variables(1 + 2 * m);
        std::vector<MPolynomial> air;
        for (size_t i = 0; i < m; i++) {
            MPolynomial lhs = MPolynomial::constant(0);
            for (size_t k = 0; k < m; k++) lhs = lhs + MPolynomial::constant(MDS[i][k].value) * variables[1 + k].pow(alpha);
            lhs = lhs + first[i];
            MPolynomial rhs = MPolynomial::constant(0);
            for (size_t k = 0; k < m; k++)
                rhs = rhs + MPolynomial::constant(MDS_inv[i][k].value) * (variables[1 + m + k] - second[k]);
            rhs = rhs.pow(alpha);
            air.push_back(lhs - rhs);
        }
        return air;
    }
};

// ---------------------------------------------------------------------------
// stark.rs
// ---------------------------------------------------------------------------
// n -> n random bytes; stands where the reference calls rand::thread_rng().fill_bytes
// (stark.rs:244-249,345-351) so that a proof can be reproduced from its random bytes.
typedef std::function<void(uint8_t*, size_t)> RandomSource;

inline RandomSource system_random()
{
    return [](uint8_t* out, size_t n) {
        std::random_device rd;
        for (size_t i = 0; i < n; i++) out[i] = (uint8_t)rd();
    };
}

// device memory owned for the length of a proof
struct DeviceBuffer {
    void* p = nullptr;
    explicit DeviceBuffer(size_t bytes) { check(stark_dev_alloc(context(), bytes ? bytes : 16, &p)); }
    ~DeviceBuffer() { if (p) stark_dev_free(context(), p); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    uint64_t* words() const { return (uint64_t*)p; }
};

struct Stark {
    size_t expansion_factor, num_colinearity_tests, security_level, num_randomizers, num_registers,
        original_trace_length;
    FieldElement generator, omega, omicron;
    std::vector<FieldElement> omicron_domain;
    size_t omicron_domain_length, fri_domain_length;
    Fri fri;

    // :34-86.  The reference also asserts that omicron is the 128th root it was
    // written for (:65), which ties it to one trace length; check_omicron keeps
    // that assertion for drop-in behaviour and can be lifted for other sizes.
    Stark(size_t expansion_factor_, size_t num_colinearity_tests_, size_t security_level_, size_t num_registers_,
          size_t original_trace_length_, bool check_omicron = true)
        : expansion_factor(expansion_factor_), num_colinearity_tests(num_colinearity_tests_),
          security_level(security_level_), num_randomizers(4 * num_colinearity_tests_), num_registers(num_registers_),
          original_trace_length(original_trace_length_), generator(FieldElement::generator()),
          omicron_domain_length(domain_length_for(original_trace_length_ + 4 * num_colinearity_tests_)),
          fri_domain_length(omicron_domain_length * expansion_factor_),
          fri(FieldElement::generator(), FieldElement::primitive_nth_root(fri_domain_length), fri_domain_length,
              expansion_factor_, num_colinearity_tests_)
    {
        omega = FieldElement::primitive_nth_root(fri_domain_length);
        omicron = FieldElement::primitive_nth_root(omicron_domain_length);
        FieldElement w = FieldElement::one();
        for (size_t i = 0; i < omicron_domain_length; i++) {
            omicron_domain.push_back(w);
            w = w * omicron;
        }
        if (check_omicron && omicron.to_string() != "65907963977709178563567092354521124432")
            throw Panic("assertion failed: omicron.value == 65907963977709178563567092354521124432");
    }
    static size_t domain_length_for(size_t randomized_trace_length)                           // :50-55
    {
        size_t product = randomized_trace_length * 2, bits = 0;
        while (product >> bits) bits++;
        return (size_t)1 << bits;
    }

    std::vector<Polynomial> boundary_zeroifiers(const std::vector<BoundaryCondition>& boundary) const   // :89-107
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
    std::vector<Polynomial> boundary_interpolants(const std::vector<BoundaryCondition>& boundary) const   // :110-131
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
    Polynomial transition_zeroifier() const                                                 // :134-137
    {
        return Polynomial::zeroifier_domain(
            std::vector<FieldElement>(omicron_domain.begin(), omicron_domain.begin() + (original_trace_length - 1)));
    }
    std::vector<size_t> transition_degree_bounds(const std::vector<MPolynomial>& tc) const   // :140-168
    {
        std::vector<size_t> point_degrees{1};
        for (size_t i = 0; i < 2 * num_randomizers; i++) point_degrees.push_back(original_trace_length + num_randomizers - 1);
        std::vector<size_t> maxes;
        for (const MPolynomial& a : tc) {
            if (a.dict.empty()) throw Panic("called `Option::unwrap()` on a `None` value");
            size_t mx = 0;
            for (const auto& kv : a.dict) {
                size_t sum = 0;
                for (size_t i = 0; i < kv.first.size() && i < point_degrees.size(); i++)
                    sum += point_degrees[i] * (size_t)kv.first[i];
                mx = std::max(mx, sum);
            }
            maxes.push_back(mx);
        }
        return maxes;
    }
    std::vector<size_t> transition_quotient_degree_bounds(const std::vector<MPolynomial>& tc) const   // :171-178
    {
        std::vector<size_t> b;
        for (size_t d : transition_degree_bounds(tc)) b.push_back(d - (original_trace_length - 1));
        return b;
    }
    std::vector<size_t> boundary_quotient_degree_bounds(size_t randomized_trace_length,
                                                        const std::vector<BoundaryCondition>& boundary) const   // :181-188
    {
        std::vector<size_t> b;
        for (const Polynomial& bz : boundary_zeroifiers(boundary)) b.push_back(randomized_trace_length - 1 - bz.degree());
        return b;
    }
    size_t max_degree(const std::vector<MPolynomial>& tc) const                             // :191-202
    {
        std::vector<size_t> b = transition_quotient_degree_bounds(tc);
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

    // A set of polynomials evaluated on the FRI domain generator * omega^i in one
    // batched low-degree extension, resident on the device, one Merkle tree each.
    struct Codewords {
        size_t n = 0, count = 0;
        std::unique_ptr<DeviceBuffer> data;
        std::vector<std::unique_ptr<TreeHandle>> trees;
        std::vector<HashOutput> roots;
        const uint64_t* column(size_t c) const { return data->words() + 2 * c * n; }
        FieldElement at(size_t c, size_t i) const
        {
            uint64_t v[2];
            check(stark_copy_d2h(context(), v, column(c) + 2 * i, 16));
            return FieldElement{((u128)v[1] << 64) | v[0]};
        }
        std::vector<HashOutput> open(size_t c, size_t i) const { return Merkle::open_tree(trees[c]->t, i); }
        // the paths of many indices of one column in one launch
        std::vector<std::vector<HashOutput>> open_many(size_t c, const std::vector<uint64_t>& idx) const
        {
            return Merkle::open_tree_many(trees[c]->t, idx);
        }
    };
    Codewords commit_polynomials(const std::vector<Polynomial>& polys, bool with_trees) const
    {
        Codewords cw;
        cw.n = fri_domain_length;
        cw.count = polys.size();
        size_t stride = 1;
        for (const Polynomial& p : polys) stride = std::max(stride, p.coeffs.size());
        if (stride > cw.n) throw Panic("polynomial longer than the FRI domain");     // Horner would not truncate (univariate_poly.rs:44-54)
        std::vector<uint64_t> host(2 * stride * polys.size(), 0);
        for (size_t c = 0; c < polys.size(); c++) {
            const std::vector<FieldElement>& k = polys[c].coeffs;                     // highest first -> lowest first
            for (size_t j = 0; j < k.size(); j++) {
                const FieldElement& x = k[k.size() - 1 - j];
                host[2 * (c * stride + j)] = (uint64_t)x.value;
                host[2 * (c * stride + j) + 1] = (uint64_t)(x.value >> 64);
            }
        }
        DeviceBuffer coeffs(host.size() * 8);
        check(stark_copy_h2d(context(), coeffs.p, host.data(), host.size() * 8));
        cw.data.reset(new DeviceBuffer(cw.n * cw.count * 16));
        uint64_t off[2] = {(uint64_t)generator.value, (uint64_t)(generator.value >> 64)};
        check(stark_coset_lde_batch_dev(context(), coeffs.words(), stride, stride, polys.size(), cw.n, off,
                                        cw.data->words(), cw.n));
        if (with_trees)
            for (size_t c = 0; c < polys.size(); c++) {
                cw.trees.emplace_back(new TreeHandle());
                cw.roots.emplace_back();
                check(stark_merkle_commit_fe_dev(context(), cw.column(c), cw.n, cw.roots.back().data(),
                                                 &cw.trees.back()->t));
            }
        return cw;
    }

    std::vector<uint8_t> prove(std::vector<std::vector<FieldElement>> trace, const std::vector<MPolynomial>& tc,
                               const std::vector<BoundaryCondition>& boundary,
                               const RandomSource& rng = system_random()) const            // :223-471
    {
        ProofStream ps;
        auto draw = [&]() {
            std::vector<uint8_t> b(17);
            rng(b.data(), b.size());
            return FieldElement::sample(b);
        };
        for (size_t k = 0; k < num_randomizers; k++) {                                      // :237-253
            std::vector<FieldElement> row;
            for (size_t s = 0; s < num_registers; s++) row.push_back(draw());
            trace.push_back(row);
        }
        std::vector<FieldElement> trace_domain;
        for (size_t i = 0; i < trace.size(); i++) trace_domain.push_back(omicron.pow(i));
        std::vector<Polynomial> trace_polynomials;                                          // :260-275
        for (size_t s = 0; s < num_registers; s++) {
            std::vector<FieldElement> single;
            for (size_t c = 0; c < trace.size(); c++) single.push_back(trace[c].at(s));
            trace_polynomials.push_back(Polynomial::lagrange(trace_domain, single));
        }
        std::vector<Polynomial> interpolants = boundary_interpolants(boundary), zeroifiers = boundary_zeroifiers(boundary);
        std::vector<Polynomial> boundary_quotients;                                         // :282-288
        for (size_t s = 0; s < num_registers; s++)
            boundary_quotients.push_back((trace_polynomials[s] - interpolants[s]) / zeroifiers[s]);
        Codewords bq = commit_polynomials(boundary_quotients, true);                        // :291-306
        for (size_t s = 0; s < num_registers; s++) ps.push(wire::hex_encode(bq.roots[s].data(), 32));

        std::vector<Polynomial> point{Polynomial{{FieldElement::one(), FieldElement::zero()}}};   // :309-325
        point.insert(point.end(), trace_polynomials.begin(), trace_polynomials.end());
        for (const Polynomial& tp : trace_polynomials) point.push_back(tp.compose(Polynomial{{omicron, FieldElement::zero()}}));
        std::vector<Polynomial> transition_quotients;                                       // :328-341
        Polynomial tz = transition_zeroifier();
        for (const MPolynomial& a : tc) transition_quotients.push_back(a.eval_symbolic(point) / tz);

        size_t max_deg = max_degree(tc);
        Polynomial randomizer_poly;                                                         // :344-353
        for (size_t i = 0; i < max_deg + 1; i++) randomizer_poly.coeffs.push_back(draw());
        Codewords rz = commit_polynomials({randomizer_poly}, true);                         // :356-360
        ps.push(wire::hex_encode(rz.roots[0].data(), 32));

        std::vector<FieldElement> weights =
            sample_weights(1 + 2 * transition_quotients.size() + 2 * boundary_quotients.size(), ps.prover_fiat_shamir(32));
        std::vector<size_t> tq_bounds = transition_quotient_degree_bounds(tc);              // :379-380
        for (size_t i = 0; i < transition_quotients.size(); i++)
            if (transition_quotients[i].degree() != tq_bounds[i])
                throw Panic("assertion failed: tq_degrees == self.transition_quotient_degree_bounds(transition_constraints.clone())");
        Polynomial x{{FieldElement::one(), FieldElement::zero()}};                           // :385-398
        std::vector<Polynomial> terms{randomizer_poly};
        for (size_t i = 0; i < transition_quotients.size(); i++) {
            terms.push_back(transition_quotients[i]);
            terms.push_back(x.pow(max_deg - tq_bounds[i]) * transition_quotients[i]);
        }
        std::vector<size_t> bq_bounds = boundary_quotient_degree_bounds(trace.size(), boundary);
        for (size_t i = 0; i < num_registers; i++) {
            terms.push_back(boundary_quotients[i]);
            terms.push_back(x.pow(max_deg - bq_bounds[i]) * boundary_quotients[i]);
        }
        Polynomial combination{{FieldElement::zero()}};                                     // :401-404
        for (size_t i = 0; i < terms.size(); i++) combination = combination + Polynomial{{weights[i]}} * terms[i];
        Codewords comb = commit_polynomials({combination}, false);                          // :406
        std::vector<size_t> indices = fri.prove_resident(comb.column(0), ps);               // :409-410
        std::sort(indices.begin(), indices.end());
        std::vector<size_t> duplicated = indices;                                           // :415-419
        for (size_t i : indices) duplicated.push_back((i + expansion_factor) % fri.domain_length);
        std::sort(duplicated.begin(), duplicated.end());
        const std::vector<uint64_t> dup64(duplicated.begin(), duplicated.end()), idx64(indices.begin(), indices.end());
        for (size_t s = 0; s < num_registers; s++) {                                        // :429-443
            const std::vector<std::vector<HashOutput>> paths = bq.open_many(s, dup64);
            for (size_t k = 0; k < duplicated.size(); k++) {
                ps.push(wire::json_field_element(bq.at(s, duplicated[k])));
                ps.push(wire::json_hash_path(paths[k]));
            }
        }
        const std::vector<std::vector<HashOutput>> rz_paths = rz.open_many(0, idx64);
        for (size_t k = 0; k < indices.size(); k++) {                                       // :449-464
            ps.push(wire::json_field_element(rz.at(0, indices[k])));
            ps.push(wire::json_hash_path(rz_paths[k]));
        }
        return ps.serialize();                                                              // :470
    }

    bool verify(const std::vector<uint8_t>& proof, const std::vector<MPolynomial>& tc,
                const std::vector<BoundaryCondition>& boundary) const                       // :474-723
    {
        size_t max_cycle = 0;
        for (const auto& b : boundary) max_cycle = std::max(max_cycle, std::get<0>(b));
        size_t randomized_trace_length = 1 + max_cycle + num_randomizers;
        ProofStream ps = ProofStream::deserialize(proof);
        std::vector<std::string> bq_roots;
        for (size_t s = 0; s < num_registers; s++) bq_roots.push_back(ps.pull());
        std::string randomizer_root = ps.pull();
        std::vector<Polynomial> interpolants = boundary_interpolants(boundary), zeroifiers = boundary_zeroifiers(boundary);
        std::vector<FieldElement> weights =
            sample_weights(1 + 2 * tc.size() + 2 * interpolants.size(), ps.verifier_fiat_shamir(32));
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
        Polynomial tz = transition_zeroifier();
        size_t max_deg = max_degree(tc);
        std::vector<size_t> tq_bounds = transition_quotient_degree_bounds(tc);
        std::vector<size_t> bq_bounds = boundary_quotient_degree_bounds(randomized_trace_length, boundary);
        for (size_t i = 0; i < indices.size(); i++) {                                        // :589-719
            size_t current_index = indices[i];
            FieldElement xcur = generator * omega.pow(current_index);
            size_t next_index = (current_index + expansion_factor) % fri.domain_length;
            FieldElement xnext = xcur * omicron;
            std::vector<FieldElement> point{xcur}, next_trace;
            for (size_t s = 0; s < num_registers; s++) {
                point.push_back(leafs[s].at(current_index) * zeroifiers[s].eval(xcur) + interpolants[s].eval(xcur));
                next_trace.push_back(leafs[s].at(next_index) * zeroifiers[s].eval(xnext) + interpolants[s].eval(xnext));
            }
            point.insert(point.end(), next_trace.begin(), next_trace.end());
            std::vector<FieldElement> terms{randomizer.at(current_index)};
            for (size_t s = 0; s < tc.size(); s++) {
                FieldElement q = tc[s].eval(point) / tz.eval(xcur);
                terms.push_back(q);
                terms.push_back(q * xcur.pow(max_deg - tq_bounds[s]));
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

}  // namespace stark
