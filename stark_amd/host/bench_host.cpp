// Timing harness over the C++ mirror of the reference API, shaped like the reference's own
// Criterion groups: ntt_operations (benches/polynomial_benchmarks.rs:155-192; sizes 8..1024 and
// 4096, random u64-sized values as :16-21) and fri_commit (benches/fri_benchmarks.rs:43-74:
// small/medium/large, setup outside the timed region).  Every call goes through the by-value
// signatures the reference has, so packing, the host<->device copies and the launches are in.
// usage: ./bench_host [iterations]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "stark_host.hpp"

using namespace stark;
typedef std::chrono::steady_clock Clock;

static std::vector<FieldElement> random_u64_elements(size_t n, std::mt19937_64& rng)
{
    std::vector<FieldElement> v(n);
    for (auto& x : v) x = FieldElement::from((u128)rng());
    return v;
}

template <class F> static double time_us(int iters, F&& f)
{
    f();                                                   // warm-up: tables, context
    auto t0 = Clock::now();
    for (int i = 0; i < iters; i++) f();
    return std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / iters;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 20;
    std::mt19937_64 rng(0x5741524B);
    try {
        std::printf("group ntt_operations (us per call)\n%8s %16s %16s\n", "size", "ntt_interpolate", "ntt_evaluate");
        for (size_t size : {8, 16, 32, 64, 128, 256, 512, 1024, 4096}) {
            const std::vector<FieldElement> coeffs = random_u64_elements(size, rng);
            const std::vector<FieldElement> evaluations = random_u64_elements(size, rng);
            const NTT ntt(size);
            volatile uint64_t sink = 0;
            const double ti = time_us(iters, [&] { sink = sink + (uint64_t)ntt.interpolate(evaluations)[0].value; });
            const double te = time_us(iters, [&] { sink = sink + (uint64_t)ntt.evaluate(coeffs)[0].value; });
            std::printf("%8zu %16.1f %16.1f\n", size, ti, te);
        }
        std::printf("group fri_commit (us per call)\n%8s %8s %10s %6s %12s\n", "name", "domain", "expansion", "tests", "commit");
        struct Cfg { const char* name; size_t domain_length, expansion_factor, num_colinearity_tests; };
        for (const Cfg& c : {Cfg{"small", 64, 4, 2}, Cfg{"medium", 256, 4, 4}, Cfg{"large", 1024, 8, 8}}) {
            const FieldElement omega = FieldElement::primitive_nth_root(c.domain_length);
            const Fri fri(FieldElement::generator(), omega, c.domain_length, c.expansion_factor, c.num_colinearity_tests);
            Polynomial polynomial;                         // degree domain/expansion - 1, u32 coefficients
            for (size_t i = 0; i < c.domain_length / c.expansion_factor; i++)
                polynomial.coeffs.push_back(FieldElement::from((u128)(uint32_t)rng()));
            std::vector<FieldElement> domain;
            FieldElement x = FieldElement::one();
            for (size_t i = 0; i < c.domain_length; i++) { domain.push_back(x); x = x * omega; }
            const std::vector<FieldElement> codeword = polynomial.eval_domain(domain);
            volatile size_t sink = 0;
            const double t = time_us(iters, [&] {
                ProofStream proof_stream;
                sink = sink + fri.commit(codeword, proof_stream).size();
            });
            std::printf("%8s %8zu %10zu %6zu %12.1f\n", c.name, c.domain_length, c.expansion_factor, c.num_colinearity_tests, t);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "bench_host: %s\n", e.what());
        return 1;
    }
    return 0;
}
