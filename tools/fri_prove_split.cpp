// Where Fri::prove's time beyond the commit goes (2^24 codeword, blowup 4, 17 colinearity tests):
// commit, the query round trip (stark_fri_query_all), and the formatting of the answers into the stream.
// build: g++ -O2 -std=c++17 -pthread -Iinclude tools/fri_prove_split.cpp -o tools/_build/fri_prove_split -Lstark_amd/csrc -lstark_hip -Wl,-rpath,'$ORIGIN/../../stark_amd/csrc'
#include "../stark_amd/host/stark_host.hpp"
#include <chrono>
#include <cstdio>
using namespace stark;
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = (size_t)1 << 24, t = 17;
    stark_ctx* ctx = context();
    std::vector<uint64_t> host(2 * n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        host[2 * i] = s;
        host[2 * i + 1] = (s * 0x9E3779B97F4A7C15ull) >> 9;     // below p
    }
    void* d = nullptr;
    check(stark_dev_alloc(ctx, 16 * n, &d));
    check(stark_copy_h2d(ctx, d, host.data(), 16 * n));
    FieldElement omega = FieldElement::primitive_nth_root(n);
    Fri fri(FieldElement::generator(), omega, n, 4, t);
    for (int rep = 0; rep < 6; rep++) {
        ProofStream ps;
        double t0 = now();
        std::unique_ptr<LayersHandle> h = fri.commit_resident((const uint64_t*)d, ps);
        double t1 = now();
        size_t layers = stark_fri_layers_count(h->l);
        std::vector<size_t> top = fri.sample_indices(ps.prover_fiat_shamir(32), n / 2, stark_fri_layer_len(h->l, layers - 1), t);
        double t2 = now();
        std::vector<uint64_t> top64(top.begin(), top.begin() + t), vals((layers - 1) * 6 * t);
        size_t digests = 0;
        for (size_t i = 0; i + 1 < layers; i++) {
            size_t a = 0, b = 0;
            while (((size_t)1 << a) < stark_fri_layer_len(h->l, i)) a++;
            while (((size_t)1 << b) < stark_fri_layer_len(h->l, i + 1)) b++;
            digests += t * (2 * a + b);
        }
        std::vector<uint8_t> raw(32 * digests + 32);
        check(stark_fri_query_all(h->l, top64.data(), t, vals.data(), raw.data()));
        double t3 = now();
        ProofStream ps2;
        double t4 = now();
        std::vector<size_t> idx = fri.prove_resident((const uint64_t*)d, ps2);
        double t5 = now();
        h.reset();
        double t6 = now();
        std::printf("commit %.3f  sample %.3f  query_all %.3f  | whole prove %.3f  (free of layers %.3f)\n", t1 - t0, t2 - t1, t3 - t2,
                    t5 - t4, t6 - t5);
    }
    return 0;
}
