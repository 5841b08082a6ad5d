// Host-side mirror of the reference crate's hot-path API (zk_stark,
// src/modules/{field,ntt,univariate_poly,merkle,proof_stream,fri}.rs) written
// above the C ABI of include/stark_hip.h.  Same type and method names, same
// argument meaning, same failure texts (the reference panics; these throw
// stark::Panic carrying the same message).  Bulk arithmetic goes to the GPU
// through the C ABI; scalars, the transcript and the verifier stay here.
//
// The reference's toolchain (Rust) is not available where this was written,
// so this C++ layer is what exercises the boundary; a Rust binding of the same
// entry points replaces it where cargo is available.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/stark_hip.h"
#include "../csrc/field_host.hpp"
#include "hash_host.hpp"

namespace stark {

typedef unsigned __int128 u128;

struct Panic : std::runtime_error {
    explicit Panic(const std::string& m) : std::runtime_error(m) {}
};

inline void check(int32_t rc)
{
    if (rc != STARK_OK) throw Panic(stark_last_error());
}

// One lazily created device context per process (the reference has no state).
// One process drives one GPU: STARK_DEVICE, else the launcher's LOCAL_RANK, else 0.
inline stark_ctx* context()
{
    static stark_ctx* ctx = [] {
        const char* e = std::getenv("STARK_DEVICE");
        if (!e || !*e) e = std::getenv("LOCAL_RANK");
        int dev = (e && *e) ? std::atoi(e) : 0;
        stark_ctx* c = nullptr;
        check(stark_ctx_create(dev, &c));
        return c;
    }();
    return ctx;
}

inline const starkf::HostField& host_field()
{
    static const starkf::HostField F;
    return F;
}

// ---------------------------------------------------------------------------
// field.rs
// ---------------------------------------------------------------------------
struct FieldElement {
    u128 value = 0;                                    // canonical, field.rs:11-14

    static u128 modulus() { return ((u128)starkf::P_HI << 64) | starkf::P_LO; }            // field.rs:32
    static FieldElement from(u128 v) { return FieldElement{v % modulus()}; }              // field.rs:18-26
    static FieldElement generator() { return FieldElement{((u128)starkf::GEN_HI << 64) | starkf::GEN_LO}; }  // :29
    static FieldElement zero() { return FieldElement{0}; }
    static FieldElement one() { return FieldElement{1}; }
    bool is_zero() const { return value == 0; }

    FieldElement pow(u128 exp) const { return wrap(host_field().pow(raw(), exp)); }         // field.rs:42-56
    FieldElement inverse() const { return wrap(host_field().inverse(raw())); }               // field.rs:67-93
    static FieldElement primitive_nth_root(u128 n)                                           // field.rs:96-107
    {
        if (!(n <= ((u128)1 << 119) && n != 0 && (n & (n - 1)) == 0))
            throw Panic("Field does not have nth root of unity where n > 2^119 or not power of two.");
        unsigned log_n = 0;
        while (((u128)1 << log_n) < n) log_n++;
        return wrap(host_field().primitive_nth_root(log_n));
    }
    static FieldElement sample(const std::vector<uint8_t>& bytes)                            // field.rs:110-116
    {
        const starkf::HostField& F = host_field();
        starkf::fe acc{0, 0};
        const starkf::fe b256{256, 0};
        for (uint8_t b : bytes) acc = starkf::fe_add(F.mul(acc, b256), starkf::fe{b, 0});
        return wrap(acc);
    }

    FieldElement operator+(const FieldElement& o) const { return wrap(starkf::fe_add(raw(), o.raw())); }
    FieldElement operator-(const FieldElement& o) const { return wrap(starkf::fe_sub(raw(), o.raw())); }
    FieldElement operator*(const FieldElement& o) const { return wrap(host_field().mul(raw(), o.raw())); }
    FieldElement operator/(const FieldElement& o) const { return *this * o.inverse(); }       // field.rs:139-142
    FieldElement operator-() const { return wrap(starkf::fe_neg(raw())); }
    bool operator==(const FieldElement& o) const { return value == o.value; }
    bool operator!=(const FieldElement& o) const { return value != o.value; }

    std::string to_string() const                                                            // Display, field.rs:149-153
    {
        // three chunks of at most 19 digits: two 128-bit divisions instead of one per digit (a FRI proof
        // prints some nine hundred of these between its last device call and its return)
        const u128 CH = (u128)10000000000000000000ull;                       // 10^19
        uint64_t part[3];
        u128 v = value;
        int parts = 0;
        do {
            part[parts++] = (uint64_t)(v % CH);
            v /= CH;
        } while (v);
        char buf[64];
        char* end = buf + sizeof(buf);
        char* p = end;
        for (int k = 0; k < parts; k++) {
            uint64_t x = part[k];
            char* chunk_end = p;
            do {
                *--p = (char)('0' + (int)(x % 10));
                x /= 10;
            } while (x);
            if (k + 1 < parts)
                while (chunk_end - p < 19) *--p = '0';                       // inner chunks keep their leading zeros
        }
        return std::string(p, (size_t)(end - p));
    }
    static FieldElement from_decimal(const std::string& s)
    {
        const starkf::HostField& F = host_field();
        starkf::fe acc{0, 0};
        for (char c : s) {
            if (c < '0' || c > '9') throw Panic("BigInt parse err");
            acc = starkf::fe_add(F.mul(acc, starkf::fe{10, 0}), starkf::fe{(uint64_t)(c - '0'), 0});
        }
        return wrap(acc);
    }

    starkf::fe raw() const { return starkf::fe{(uint64_t)value, (uint64_t)(value >> 64)}; }
    static FieldElement wrap(starkf::fe x) { return FieldElement{starkf::to_u128(x)}; }
};

// packed {lo,hi} image of a vector (the layout of the C ABI)
inline std::vector<uint64_t> pack(const std::vector<FieldElement>& v)
{
    std::vector<uint64_t> out(2 * v.size());
    for (size_t i = 0; i < v.size(); i++) {
        out[2 * i] = (uint64_t)v[i].value;
        out[2 * i + 1] = (uint64_t)(v[i].value >> 64);
    }
    return out;
}
inline std::vector<FieldElement> unpack(const std::vector<uint64_t>& p)
{
    std::vector<FieldElement> v(p.size() / 2);
    for (size_t i = 0; i < v.size(); i++) v[i].value = ((u128)p[2 * i + 1] << 64) | p[2 * i];
    return v;
}

// ---------------------------------------------------------------------------
// wire formats (bincode 1.3.3 / num-bigint serde / serde_json / hex)
// ---------------------------------------------------------------------------
namespace wire {

inline std::vector<uint8_t> bincode_field_element(const FieldElement& x)        // leaf bytes, fri.rs:119
{
    uint32_t d[4] = {(uint32_t)x.value, (uint32_t)(x.value >> 32), (uint32_t)(x.value >> 64), (uint32_t)(x.value >> 96)};
    uint64_t nd = 4;
    while (nd > 0 && d[nd - 1] == 0) nd--;
    std::vector<uint8_t> out(9 + 4 * nd);
    out[0] = nd ? 1 : 0;
    std::memcpy(out.data() + 1, &nd, 8);
    std::memcpy(out.data() + 9, d, 4 * nd);
    return out;
}

inline std::string hex_encode(const uint8_t* p, size_t n)
{
    static const char* k = "0123456789abcdef";
    std::string s(2 * n, '0');
    for (size_t i = 0; i < n; i++) {
        s[2 * i] = k[p[i] >> 4];
        s[2 * i + 1] = k[p[i] & 15];
    }
    return s;
}
inline std::vector<uint8_t> hex_decode(const std::string& s)
{
    auto nib = [](char c) -> int {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        throw Panic("invalid hex");
    };
    std::vector<uint8_t> out(s.size() / 2);
    for (size_t i = 0; i < out.size(); i++) out[i] = (uint8_t)(nib(s[2 * i]) * 16 + nib(s[2 * i + 1]));
    return out;
}

inline std::string json_field_element(const FieldElement& x)                    // {"value":[sign,[digits]]}
{
    uint32_t d[4] = {(uint32_t)x.value, (uint32_t)(x.value >> 32), (uint32_t)(x.value >> 64), (uint32_t)(x.value >> 96)};
    int nd = 4;
    while (nd > 0 && d[nd - 1] == 0) nd--;
    std::string s = std::string("{\"value\":[") + (nd ? "1" : "0") + ",[";
    for (int i = 0; i < nd; i++) {
        if (i) s += ",";
        s += std::to_string(d[i]);
    }
    return s + "]]}";
}
inline std::string json_field_vec(const std::vector<FieldElement>& v)           // fri.rs:146
{
    std::string s = "[";
    for (size_t i = 0; i < v.size(); i++) {
        if (i) s += ",";
        s += json_field_element(v[i]);
    }
    return s + "]";
}

// minimal scanner for the three JSON shapes on this path
struct Scanner {
    const std::string& s;
    size_t i = 0;
    explicit Scanner(const std::string& str) : s(str) {}
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
    bool peek(char c) { ws(); return i < s.size() && s[i] == c; }
    void expect(char c)
    {
        ws();
        if (i >= s.size() || s[i] != c) throw Panic("JSON parse error");
        i++;
    }
    bool accept(char c)
    {
        if (peek(c)) { i++; return true; }
        return false;
    }
    int64_t integer()
    {
        ws();
        bool neg = false;
        if (i < s.size() && s[i] == '-') { neg = true; i++; }
        if (i >= s.size() || s[i] < '0' || s[i] > '9') throw Panic("JSON parse error");
        int64_t v = 0;
        while (i < s.size() && s[i] >= '0' && s[i] <= '9') v = v * 10 + (s[i++] - '0');
        return neg ? -v : v;
    }
    std::string string()
    {
        expect('"');
        std::string out;
        while (i < s.size() && s[i] != '"') out.push_back(s[i++]);
        expect('"');
        return out;
    }
    void key(const char* k)
    {
        if (string() != k) throw Panic("JSON parse error");
        expect(':');
    }
};

inline FieldElement scan_field_element(Scanner& sc)                            // {"value":[sign,[digits]]}
{
    sc.expect('{');
    sc.key("value");
    sc.expect('[');
    int64_t sign = sc.integer();
    sc.expect(',');
    sc.expect('[');
    u128 v = 0;
    int k = 0;
    if (!sc.accept(']')) {
        do {
            v |= (u128)(uint32_t)sc.integer() << (32 * k++);
        } while (sc.accept(','));
        sc.expect(']');
    }
    sc.expect(']');
    sc.expect('}');
    FieldElement e = FieldElement::from(v);
    return sign < 0 ? -e : e;
}
inline FieldElement json_field_element_parse(const std::string& str)           // stark.rs:538,566
{
    Scanner sc(str);
    return scan_field_element(sc);
}
inline std::vector<FieldElement> json_field_vec_parse(const std::string& str)   // fri.rs:271
{
    Scanner sc(str);
    std::vector<FieldElement> out;
    sc.expect('[');
    if (!sc.accept(']')) {
        do {
            out.push_back(scan_field_element(sc));
        } while (sc.accept(','));
        sc.expect(']');
    }
    return out;
}

typedef std::array<uint8_t, 32> HashOutput;

inline std::string json_hash_path_bytes(const uint8_t* digests, size_t count)    // fri.rs:188, from `count` digests back to back
{
    // a proof carries hundreds of thousands of these bytes: a table of "0," .. "255," and one buffer
    struct Table {
        char text[256][4];
        uint8_t len[256];
        Table()
        {
            for (unsigned v = 0; v < 256; v++) {
                unsigned k = 0;
                if (v >= 100) text[v][k++] = (char)('0' + v / 100);
                if (v >= 10) text[v][k++] = (char)('0' + v / 10 % 10);
                text[v][k++] = (char)('0' + v % 10);
                text[v][k++] = ',';
                len[v] = (uint8_t)k;
            }
        }
    };
    static const Table T;
    std::string s(count * (32 * 4 + 2) + 2, '\0');
    char* p = &s[0];
    *p++ = '[';
    for (size_t i = 0; i < count; i++) {
        *p++ = '[';
        for (int b = 0; b < 32; b++) {
            const unsigned v = digests[32 * i + b];
            std::memcpy(p, T.text[v], 4);
            p += T.len[v];
        }
        p[-1] = ']';                      // over the last comma
        *p++ = ',';
    }
    if (count) p--;                       // drop the comma after the last digest
    *p++ = ']';
    s.resize((size_t)(p - &s[0]));
    return s;
}
inline std::string json_hash_path(const std::vector<HashOutput>& path)
{
    return json_hash_path_bytes(path.empty() ? nullptr : path[0].data(), path.size());   // arrays of 32 bytes lie back to back
}
inline std::vector<HashOutput> json_hash_path_parse(const std::string& str)
{
    Scanner sc(str);
    std::vector<HashOutput> out;
    sc.expect('[');
    if (!sc.accept(']')) {
        do {
            HashOutput h{};
            sc.expect('[');
            for (int b = 0; b < 32; b++) {
                if (b) sc.expect(',');
                h[b] = (uint8_t)sc.integer();
            }
            sc.expect(']');
            out.push_back(h);
        } while (sc.accept(','));
        sc.expect(']');
    }
    return out;
}

inline std::string json_string_triple(const std::string& a, const std::string& b, const std::string& c)
{
    return "[\"" + a + "\",\"" + b + "\",\"" + c + "\"]";                        // fri.rs:169-178
}
inline std::array<std::string, 3> json_string_triple_parse(const std::string& str)
{
    Scanner sc(str);
    std::array<std::string, 3> out;
    sc.expect('[');
    out[0] = sc.string();
    sc.expect(',');
    out[1] = sc.string();
    sc.expect(',');
    out[2] = sc.string();
    sc.expect(']');
    return out;
}

}  // namespace wire

using wire::HashOutput;

// ---------------------------------------------------------------------------
// proof_stream.rs
// ---------------------------------------------------------------------------
struct ProofStream {
    std::vector<std::string> objects;
    size_t read_idx = 0;

    void push(std::string obj) { objects.push_back(std::move(obj)); }                  // :23-25
    std::string pull()                                                                 // :28-33
    {
        if (!(read_idx < objects.size())) throw Panic("ProofStream: cannot pull object; queue empty.");
        return objects[read_idx++];
    }
    static std::vector<uint8_t> bincode(const std::vector<std::string>& objs, size_t upto)
    {
        size_t total = 8;
        for (size_t i = 0; i < upto; i++) total += 8 + objs[i].size();
        std::vector<uint8_t> out(total);
        uint8_t* p = out.data();
        auto put64 = [&](uint64_t v) {
            for (int i = 0; i < 8; i++) *p++ = (uint8_t)(v >> (8 * i));
        };
        put64(upto);
        for (size_t i = 0; i < upto; i++) {
            put64(objs[i].size());
            std::memcpy(p, objs[i].data(), objs[i].size());
            p += objs[i].size();
        }
        return out;
    }
    std::vector<uint8_t> serialize() const { return bincode(objects, objects.size()); }   // :36-38
    static ProofStream deserialize(const std::vector<uint8_t>& bb)                          // :41-47
    {
        ProofStream ps;
        auto get64 = [&](size_t off) {
            if (off + 8 > bb.size()) throw Panic("ProofStream: truncated input");
            uint64_t v = 0;
            for (int i = 0; i < 8; i++) v |= (uint64_t)bb[off + i] << (8 * i);
            return v;
        };
        uint64_t n = get64(0);
        size_t off = 8;
        for (uint64_t k = 0; k < n; k++) {
            uint64_t len = get64(off);
            off += 8;
            if (off + len > bb.size()) throw Panic("ProofStream: truncated input");
            ps.objects.emplace_back(bb.begin() + off, bb.begin() + off + len);
            off += len;
        }
        return ps;
    }
    std::vector<uint8_t> prover_fiat_shamir(size_t num_bytes) const                         // :50-58
    {
        std::vector<uint8_t> data = serialize();
        return hash::shake256(data.data(), data.size(), num_bytes);
    }
    std::vector<uint8_t> verifier_fiat_shamir(size_t num_bytes) const                       // :61-69
    {
        std::vector<uint8_t> data = bincode(objects, read_idx);
        return hash::shake256(data.data(), data.size(), num_bytes);
    }
};

// ---------------------------------------------------------------------------
// ntt.rs
// ---------------------------------------------------------------------------
struct NTT {
    size_t n;
    FieldElement omega, omega_inv;

    explicit NTT(size_t n_) : n(n_)                                                         // ntt.rs:14-22
    {
        if (n == 0 || (n & (n - 1))) throw Panic("NTT size must be power of 2");
        if (n > ((size_t)1 << 30)) throw Panic("NTT size too large for practical use");
        omega = FieldElement::primitive_nth_root(n);
        omega_inv = omega.inverse();
    }
    void forward(std::vector<FieldElement>& coeffs) const { run(coeffs, 0); }               // :25-28
    void inverse(std::vector<FieldElement>& evals) const { run(evals, 1); }                 // :31-40
    std::vector<FieldElement> interpolate(std::vector<FieldElement> evaluations) const      // :92-97
    {
        if (evaluations.size() != n) throw Panic("Evaluations size must match NTT size");
        inverse(evaluations);
        return evaluations;
    }
    std::vector<FieldElement> evaluate(std::vector<FieldElement> coefficients) const        // :101-107
    {
        coefficients.resize(n, FieldElement::zero());      // pads, or truncates when longer
        forward(coefficients);
        return coefficients;
    }
    std::vector<FieldElement> multiply(const std::vector<FieldElement>& a, const std::vector<FieldElement>& b) const
    {                                                                                       // :110-135
        size_t result_size = a.size() + b.size() - 1;
        size_t ntt_size = 1;
        while (ntt_size < result_size) ntt_size <<= 1;
        NTT ntt(ntt_size);
        std::vector<FieldElement> ap = a, bp = b;
        ap.resize(ntt_size, FieldElement::zero());
        bp.resize(ntt_size, FieldElement::zero());
        ntt.forward(ap);
        ntt.forward(bp);
        std::vector<uint64_t> pa = pack(ap), pb = pack(bp), pc(pa.size());
        check(stark_vec_mul(context(), pa.data(), pb.data(), pc.data(), ntt_size));
        std::vector<FieldElement> c = unpack(pc);
        ntt.inverse(c);
        c.resize(result_size);
        return c;
    }

private:
    void run(std::vector<FieldElement>& v, int inv) const
    {
        if (v.size() != n) throw Panic("Input size must match NTT size");
        std::vector<uint64_t> p = pack(v);
        check(stark_ntt(context(), p.data(), n, inv));
        v = unpack(p);
    }
};

// Evaluate a polynomial (lowest-degree coefficient first, at most n_out of them)
// on offset * omega_n^i: the device path behind Polynomial::eval_domain.
inline std::vector<FieldElement> coset_evaluate(const std::vector<FieldElement>& coeffs_low_first, size_t n_out,
                                                const FieldElement& offset)
{
    std::vector<uint64_t> c = pack(coeffs_low_first), out(2 * n_out);
    uint64_t off[2] = {(uint64_t)offset.value, (uint64_t)(offset.value >> 64)};
    check(stark_coset_lde_batch(context(), c.data(), coeffs_low_first.size(), coeffs_low_first.size(), 1, n_out, off,
                                out.data(), n_out));
    return unpack(out);
}

// ---------------------------------------------------------------------------
// univariate_poly.rs (the evaluate / interpolate front door, plus the small
// host algebra the FRI verifier needs).  Coefficients highest -> lowest degree.
// ---------------------------------------------------------------------------
struct Polynomial {
    std::vector<FieldElement> coeffs;

    FieldElement eval(const FieldElement& x) const                                          // :33-41
    {
        FieldElement val = FieldElement::zero();
        for (const FieldElement& c : coeffs) val = val * x + c;
        return val;
    }
    // x[i] == x[0] * omega_n^i for all i (one running product)
    static bool is_coset_of_unity(const std::vector<FieldElement>& x)
    {
        size_t n = x.size();
        if (n <= 1 || (n & (n - 1))) return false;
        FieldElement omega = FieldElement::primitive_nth_root(n);
        FieldElement cur = x[0];
        for (size_t i = 0; i < n; i++) {
            if (x[i] != cur) return false;
            cur = cur * omega;
        }
        return true;
    }
    static bool is_consecutive_powers_of_unity(const std::vector<FieldElement>& x)          // :180-194
    {
        return x.size() > 1 && x[0] == FieldElement::one() && is_coset_of_unity(x);
    }
    // :44-54.  The reference takes its NTT path only for the unshifted domain
    // and evaluates any other domain point by point; a shifted power-of-two
    // domain offset*omega^i gives the same values through the coset transform
    // (field values are canonical, so equal values are equal bits).  Horner
    // does not truncate, the NTT path does, so the coset path is only taken
    // when no coefficient would be dropped.
    std::vector<FieldElement> eval_domain(const std::vector<FieldElement>& domain) const
    {
        size_t n = domain.size();
        bool pow2 = n != 0 && (n & (n - 1)) == 0;
        if (pow2 && n >= 8 && is_consecutive_powers_of_unity(domain)) return eval_domain_ntt(n);
        if (pow2 && n >= 8 && coeffs.size() <= n && !domain[0].is_zero() && is_coset_of_unity(domain)) {
            std::vector<FieldElement> low(coeffs.rbegin(), coeffs.rend());
            return coset_evaluate(low, n, domain[0]);
        }
        std::vector<FieldElement> out;
        out.reserve(n);
        for (const FieldElement& x : domain) out.push_back(eval(x));
        return out;
    }
    std::vector<FieldElement> eval_domain_ntt(size_t domain_size) const                    // :57-66
    {
        NTT ntt(domain_size);
        std::vector<FieldElement> low(coeffs.rbegin(), coeffs.rend());
        low.resize(domain_size, FieldElement::zero());
        return ntt.evaluate(low);
    }
    size_t degree() const                                                                   // :69-85
    {
        for (size_t i = 0; i < coeffs.size(); i++)
            if (!coeffs[i].is_zero()) return coeffs.size() - 1 - i;
        return 0;
    }
    Polynomial scale(const FieldElement& f) const                                           // :197-200
    {
        Polynomial r;
        for (const FieldElement& c : coeffs) r.coeffs.push_back(c * f);
        return r;
    }
    Polynomial operator+(const Polynomial& o) const                                         // :308-342
    {
        const Polynomial& mx = coeffs.size() > o.coeffs.size() ? *this : o;
        const Polynomial& mn = coeffs.size() > o.coeffs.size() ? o : *this;
        size_t diff = mx.coeffs.size() - mn.coeffs.size();
        Polynomial r;
        for (size_t i = 0; i < diff; i++) r.coeffs.push_back(mx.coeffs[i]);
        for (size_t i = 0; i < mn.coeffs.size(); i++) r.coeffs.push_back(mx.coeffs[diff + i] + mn.coeffs[i]);
        return r;
    }
    Polynomial operator*(const Polynomial& o) const                                         // :403-420
    {
        Polynomial r;
        r.coeffs.assign(coeffs.size() + o.coeffs.size() - 1, FieldElement::zero());
        for (size_t i = 0; i < coeffs.size(); i++)
            for (size_t j = 0; j < o.coeffs.size(); j++) r.coeffs[i + j] = r.coeffs[i + j] + coeffs[i] * o.coeffs[j];
        return r;
    }
    bool is_zero() const                                                                    // :88-94
    {
        for (const FieldElement& c : coeffs)
            if (!c.is_zero()) return false;
        return true;
    }
    Polynomial operator-(const Polynomial& o) const                                         // :345-400
    {
        Polynomial r;
        if (o.coeffs.size() > coeffs.size()) {
            size_t diff = o.coeffs.size() - coeffs.size();
            for (size_t i = 0; i < diff; i++) r.coeffs.push_back(FieldElement::zero() - o.coeffs[i]);
            for (size_t i = 0; i < coeffs.size(); i++) r.coeffs.push_back(coeffs[i] - o.coeffs[diff + i]);
        } else {
            size_t diff = coeffs.size() - o.coeffs.size();
            for (size_t i = 0; i < diff; i++) r.coeffs.push_back(coeffs[i]);
            for (size_t i = 0; i < o.coeffs.size(); i++) r.coeffs.push_back(coeffs[diff + i] - o.coeffs[i]);
        }
        return r;
    }
    Polynomial operator-() const                                                            // :423-434
    {
        Polynomial r;
        for (const FieldElement& c : coeffs) r.coeffs.push_back(FieldElement::zero() - c);
        return r;
    }
    // Long division exactly as written (:437-484): the quotient has deg(self) -
    // deg(rhs) + 1 coefficients, position i of the dividend VECTOR pairs with
    // quotient[i], and the divisor's first stored coefficient is its leading one.
    Polynomial operator/(const Polynomial& rhs) const
    {
        if (rhs.is_zero()) throw Panic("Attempted to divide by zero polynomial.");
        size_t nd = degree(), dd = rhs.degree();
        if (dd > nd) return Polynomial{{FieldElement::zero()}};
        std::vector<FieldElement> dividend = coeffs;
        size_t qd = nd - dd;
        Polynomial q;
        q.coeffs.assign(qd + 1, FieldElement::zero());
        FieldElement lead_inv = rhs.coeffs[0].inverse();
        for (size_t i = 0; i <= qd; i++) {
            if (i < dividend.size() && !dividend[i].is_zero()) {
                FieldElement c = dividend[i] * lead_inv;
                q.coeffs[i] = c;
                for (size_t j = 0; j < rhs.coeffs.size(); j++)
                    if (i + j < dividend.size()) dividend[i + j] = dividend[i + j] - c * rhs.coeffs[j];
            }
        }
        return q;
    }
    // :285-303.  The reference walks the bit positions below the leading zeros of
    // a u128 exponent; squaring [1] gives [1], so for exponents under 2^64 that
    // is the ordinary left-to-right ladder.
    Polynomial pow(u128 exponent) const
    {
        if (is_zero()) return Polynomial{{FieldElement::zero()}};
        if (exponent == 0) return Polynomial{{FieldElement::one()}};
        if (exponent >> 64) throw Panic("Polynomial::pow: exponent out of range");
        Polynomial acc{{FieldElement::one()}};
        int top = 63;
        while (!((exponent >> top) & 1)) top--;
        for (int i = top; i >= 0; i--) {
            acc = acc * acc;
            if ((exponent >> i) & 1) acc = acc * *this;
        }
        return acc;
    }
    Polynomial compose(const Polynomial& other) const                                       // :203-221
    {
        if (is_zero()) return Polynomial{{FieldElement::zero()}};
        Polynomial result{{FieldElement::zero()}}, power{{FieldElement::one()}};
        for (auto it = coeffs.rbegin(); it != coeffs.rend(); ++it) {
            result = result + Polynomial{{*it}} * power;
            power = power * other;
        }
        return result;
    }
    static Polynomial zeroifier_domain(const std::vector<FieldElement>& domain)             // :254-264
    {
        Polynomial x{{FieldElement::one(), FieldElement::zero()}}, acc{{FieldElement::one()}};
        for (const FieldElement& d : domain) acc = acc * (x - Polynomial{{d}});
        return acc;
    }
    // the file format of save / load (:224-238): serde_json of { coeffs: Vec<FieldElement> }
    std::string to_json() const { return "{\"coeffs\":" + wire::json_field_vec(coeffs) + "}"; }
    static Polynomial from_json(const std::string& str)
    {
        size_t a = str.find('['), b = str.rfind(']');
        if (str.compare(0, 10, "{\"coeffs\":") != 0 || a == std::string::npos || b == std::string::npos || b < a)
            throw Panic("invalid polynomial file");
        return Polynomial{wire::json_field_vec_parse(str.substr(a, b - a + 1))};
    }
    void save(const std::string& filename) const
    {
        FILE* f = std::fopen(filename.c_str(), "wb");
        if (!f) throw Panic("cannot create " + filename);
        std::string j = to_json();
        size_t w = std::fwrite(j.data(), 1, j.size(), f);
        std::fclose(f);
        if (w != j.size()) throw Panic("short write to " + filename);
    }
    static Polynomial load(const std::string& filename)
    {
        FILE* f = std::fopen(filename.c_str(), "rb");
        if (!f) throw Panic("cannot open " + filename);
        std::string j;
        char buf[4096];
        size_t r;
        while ((r = std::fread(buf, 1, sizeof buf, f)) > 0) j.append(buf, r);
        std::fclose(f);
        return from_json(j);
    }
    static Polynomial lagrange_traditional(const std::vector<FieldElement>& x, const std::vector<FieldElement>& y)
    {                                                                                       // :104-124,147-164
        Polynomial p;
        for (size_t i = 0; i < x.size(); i++) {
            Polynomial l{{FieldElement::one()}};
            for (size_t j = 0; j < x.size(); j++) {
                if (i == j) continue;
                Polynomial num{{FieldElement::one(), -x[j]}};
                l = (l * num).scale((x[i] - x[j]).inverse());
            }
            p = p + l.scale(y[i]);
        }
        return p;
    }
    static Polynomial lagrange_ntt(const std::vector<FieldElement>& evaluations)            // :167-177
    {
        NTT ntt(evaluations.size());
        std::vector<FieldElement> c = ntt.interpolate(evaluations);
        std::reverse(c.begin(), c.end());
        return Polynomial{c};
    }
    static Polynomial lagrange(const std::vector<FieldElement>& x, const std::vector<FieldElement>& y)   // :127-144
    {
        if (x.size() != y.size()) throw Panic("x and y must be the same length");
        size_t n = x.size();
        if (n <= 8 || (n & (n - 1))) return lagrange_traditional(x, y);
        if (is_consecutive_powers_of_unity(x)) return lagrange_ntt(y);
        return lagrange_traditional(x, y);
    }
    static bool test_colinearity(const std::vector<std::pair<FieldElement, FieldElement>>& points)     // :267-282
    {
        std::vector<FieldElement> xs, ys;
        for (auto& pt : points) {
            xs.push_back(pt.first);
            ys.push_back(pt.second);
        }
        return lagrange(xs, ys).degree() == 1;
    }
};

// ---------------------------------------------------------------------------
// merkle.rs
// ---------------------------------------------------------------------------
struct TreeHandle {
    stark_tree* t = nullptr;
    ~TreeHandle() { if (t) stark_tree_free(t); }
};

struct Merkle {
    static void flatten(const std::vector<std::vector<uint8_t>>& data, std::vector<uint8_t>& flat,
                        std::vector<uint64_t>& offs)
    {
        offs.assign(data.size() + 1, 0);
        for (size_t i = 0; i < data.size(); i++) offs[i + 1] = offs[i] + data[i].size();
        flat.resize(offs.back() ? offs.back() : 1);
        for (size_t i = 0; i < data.size(); i++)
            if (!data[i].empty()) std::memcpy(flat.data() + offs[i], data[i].data(), data[i].size());
    }
    static HashOutput commit(const std::vector<std::vector<uint8_t>>& data_array)           // :38-51
    {
        std::vector<uint8_t> flat;
        std::vector<uint64_t> offs;
        flatten(data_array, flat, offs);
        HashOutput root{};
        check(stark_merkle_commit_bytes(context(), flat.data(), offs.data(), data_array.size(), root.data(), nullptr));
        return root;
    }
    static std::vector<HashOutput> open(size_t index, const std::vector<std::vector<uint8_t>>& data_array)  // :79-93
    {
        std::vector<uint8_t> flat;
        std::vector<uint64_t> offs;
        flatten(data_array, flat, offs);
        HashOutput root{};
        TreeHandle th;
        check(stark_merkle_commit_bytes(context(), flat.data(), offs.data(), data_array.size(), root.data(), &th.t));
        return open_tree(th.t, index);
    }
    static std::vector<std::vector<HashOutput>> open_tree_many(stark_tree* t, const std::vector<uint64_t>& indices)
    {
        std::vector<uint8_t> buf(indices.size() * 64 * 32 + 32);
        size_t depth = 0;
        check(stark_merkle_open_many(t, indices.data(), indices.size(), buf.data(), &depth));
        std::vector<std::vector<HashOutput>> out(indices.size(), std::vector<HashOutput>(depth));
        for (size_t i = 0; i < indices.size(); i++)
            for (size_t k = 0; k < depth; k++) std::memcpy(out[i][k].data(), buf.data() + 32 * (i * depth + k), 32);
        return out;
    }
    static std::vector<HashOutput> open_tree(stark_tree* t, size_t index)
    {
        uint8_t buf[64 * 32];
        size_t depth = 0;
        check(stark_merkle_open(t, index, buf, &depth));
        std::vector<HashOutput> path(depth);
        for (size_t k = 0; k < depth; k++) std::memcpy(path[k].data(), buf + 32 * k, 32);
        return path;
    }
    // Host evaluation of commit for short inputs (the verifier's last codeword, fri.rs:274-276).
    static HashOutput commit_host(const std::vector<std::vector<uint8_t>>& data_array)
    {
        size_t n = data_array.size();
        if (n == 0 || (n & (n - 1))) throw Panic("length must be power of two");
        std::vector<std::vector<uint8_t>> level;
        for (const auto& d : data_array) level.push_back(hash::blake2b(d.data(), d.size(), 32));
        while (level.size() > 1) {
            std::vector<std::vector<uint8_t>> next;
            for (size_t i = 0; i < level.size(); i += 2) {
                hash::Blake2b h(32);
                h.update(level[i]);
                h.update(level[i + 1]);
                next.push_back(h.finalize());
            }
            level.swap(next);
        }
        HashOutput root{};
        std::memcpy(root.data(), level[0].data(), 32);
        return root;
    }
    static bool verify(const HashOutput& root, size_t index, const std::vector<HashOutput>& path,
                       const std::vector<uint8_t>& data_element)                            // :96-132
    {
        if (path.empty() || !(index < ((size_t)1 << path.size()))) throw Panic("cannot verify invalid index");
        std::vector<uint8_t> node = hash::blake2b(data_element.data(), data_element.size(), 32);
        for (const HashOutput& sib : path) {
            hash::Blake2b h(32);
            if (index % 2 == 0) {
                h.update(node);
                h.update(sib.data(), 32);
            } else {
                h.update(sib.data(), 32);
                h.update(node);
            }
            node = h.finalize();
            index >>= 1;
        }
        return std::memcmp(node.data(), root.data(), 32) == 0;
    }
};

// ---------------------------------------------------------------------------
// fri.rs
// ---------------------------------------------------------------------------
struct LayersHandle {
    stark_fri_layers* l = nullptr;
    ~LayersHandle() { if (l) stark_fri_layers_free(l); }
};

struct Fri {
    FieldElement offset, omega;
    size_t domain_length, expansion_factor, num_colinearity_tests;

    Fri(FieldElement offset_, FieldElement omega_, size_t initial_domain_length, size_t expansion_factor_,
        size_t num_colinearity_tests_)                                                      // :31-36
        : offset(offset_), omega(omega_), domain_length(initial_domain_length), expansion_factor(expansion_factor_),
          num_colinearity_tests(num_colinearity_tests_)
    {
        if (!(num_rounds() > 0)) throw Panic("Cannot do FRI with less than 1 round");
    }
    size_t num_rounds() const                                                               // :39-51
    {
        return stark_fri_num_rounds(domain_length, expansion_factor, num_colinearity_tests);
    }
    static size_t sample_index(const std::vector<uint8_t>& bytes, size_t size)             // :81-87
    {
        size_t acc = 0;
        for (uint8_t b : bytes) acc = (acc << 8) ^ (size_t)b;
        return acc % size;
    }
    std::vector<size_t> sample_indices(const std::vector<uint8_t>& seed, size_t size, size_t reduced_size,
                                       size_t number) const                                 // :54-78
    {
        std::vector<size_t> indices, reduced;
        uint64_t counter = 0;
        while (indices.size() < number) {
            hash::Blake2b h(64);
            h.update(seed);
            uint8_t cb[8];
            for (int i = 0; i < 8; i++) cb[i] = (uint8_t)(counter >> (8 * i));
            h.update(cb, 8);
            size_t index = sample_index(h.finalize(), size);
            size_t red = index % reduced_size;
            counter++;
            if (std::find(reduced.begin(), reduced.end(), red) == reduced.end()) {
                indices.push_back(index);
                reduced.push_back(red);
            }
        }
        return indices;
    }
    std::vector<FieldElement> eval_domain() const                                           // :90-97
    {
        std::vector<FieldElement> d;
        d.reserve(domain_length);
        FieldElement w = FieldElement::one();
        for (size_t i = 0; i < domain_length; i++) {
            d.push_back(offset * w);
            w = w * omega;
        }
        return d;
    }

    // commit with everything left on the device (what prove uses)
    std::unique_ptr<LayersHandle> commit_device(const std::vector<FieldElement>& codeword, ProofStream& ps) const
    {
        std::vector<uint64_t> cw = pack(codeword);
        uint64_t off[2] = {(uint64_t)offset.value, (uint64_t)(offset.value >> 64)};
        uint64_t om[2] = {(uint64_t)omega.value, (uint64_t)(omega.value >> 64)};
        std::unique_ptr<LayersHandle> h(new LayersHandle());
        check(stark_fri_commit(context(), cw.data(), codeword.size(), off, om, expansion_factor, num_colinearity_tests,
                               &Fri::on_root, &ps, &h->l));
        size_t last = stark_fri_layers_count(h->l) - 1;
        ps.push(wire::json_field_vec(download(h->l, last)));                                 // :146
        return h;
    }
    // commit a codeword that is already resident on the device (packed, n = domain_length)
    std::unique_ptr<LayersHandle> commit_resident(const uint64_t* codeword_dev, ProofStream& ps) const
    {
        uint64_t off[2] = {(uint64_t)offset.value, (uint64_t)(offset.value >> 64)};
        uint64_t om[2] = {(uint64_t)omega.value, (uint64_t)(omega.value >> 64)};
        std::unique_ptr<LayersHandle> h(new LayersHandle());
        check(stark_fri_commit_dev(context(), codeword_dev, domain_length, off, om, expansion_factor,
                                   num_colinearity_tests, &Fri::on_root, &ps, &h->l));
        size_t last = stark_fri_layers_count(h->l) - 1;
        ps.push(wire::json_field_vec(download(h->l, last)));                                 // :146
        return h;
    }
    std::vector<std::vector<FieldElement>> commit(const std::vector<FieldElement>& codeword, ProofStream& ps) const
    {                                                                                       // :100-152
        std::unique_ptr<LayersHandle> h = commit_device(codeword, ps);
        std::vector<std::vector<FieldElement>> out;
        for (size_t i = 0; i < stark_fri_layers_count(h->l); i++) out.push_back(download(h->l, i));
        return out;
    }

    // query on resident layers: three points per test, then three paths per test (:155-209)
    std::vector<size_t> query_device(stark_fri_layers* l, size_t layer, const std::vector<size_t>& c_indices,
                                     ProofStream& ps) const
    {
        size_t cur_len = stark_fri_layer_len(l, layer);
        std::vector<size_t> a_indices = c_indices, b_indices;
        for (size_t idx : c_indices) b_indices.push_back(idx + cur_len / 2);
        size_t t = num_colinearity_tests;
        // the 3t points and 3t paths of the round in one round trip
        std::vector<uint64_t> ic(c_indices.begin(), c_indices.begin() + t), vals(6 * t);
        std::vector<uint8_t> raw(3 * t * 64 * 32 + 32);
        size_t depth = 0, depth_next = 0;
        check(stark_fri_layer_query(l, layer, ic.data(), t, vals.data(), raw.data(), &depth, &depth_next));
        std::vector<FieldElement> e = unpack(vals);
        for (size_t s = 0; s < t; s++)                                                       // :166-179
            ps.push(wire::json_string_triple(e[s].to_string(), e[t + s].to_string(), e[2 * t + s].to_string()));
        auto path_at = [&](size_t base, size_t d, size_t s) {
            std::vector<HashOutput> p(d);
            for (size_t k = 0; k < d; k++) std::memcpy(p[k].data(), raw.data() + 32 * (base + s * d + k), 32);
            return p;
        };
        for (size_t s = 0; s < t; s++) {                                                     // :182-204
            ps.push(wire::json_hash_path(path_at(0, depth, s)));
            ps.push(wire::json_hash_path(path_at(t * depth, depth, s)));
            ps.push(wire::json_hash_path(path_at(2 * t * depth, depth_next, s)));
        }
        a_indices.insert(a_indices.end(), b_indices.begin(), b_indices.end());
        return a_indices;
    }
    // the reference's signature: codewords passed by value, trees rebuilt (:155-209)
    std::vector<size_t> query(const std::vector<FieldElement>& current_codeword,
                              const std::vector<FieldElement>& next_codeword, const std::vector<size_t>& c_indices,
                              ProofStream& ps) const
    {
        std::vector<size_t> a_indices = c_indices, b_indices;
        for (size_t idx : c_indices) b_indices.push_back(idx + current_codeword.size() / 2);
        for (size_t s = 0; s < num_colinearity_tests; s++)
            ps.push(wire::json_string_triple(current_codeword.at(a_indices[s]).to_string(),
                                             current_codeword.at(b_indices[s]).to_string(),
                                             next_codeword.at(c_indices[s]).to_string()));
        TreeHandle cur, nxt;
        HashOutput root{};
        std::vector<uint64_t> pc = pack(current_codeword), pn = pack(next_codeword);
        check(stark_merkle_commit_fe(context(), pc.data(), current_codeword.size(), root.data(), &cur.t));
        check(stark_merkle_commit_fe(context(), pn.data(), next_codeword.size(), root.data(), &nxt.t));
        for (size_t s = 0; s < num_colinearity_tests; s++) {
            ps.push(wire::json_hash_path(Merkle::open_tree(cur.t, a_indices[s])));
            ps.push(wire::json_hash_path(Merkle::open_tree(cur.t, b_indices[s])));
            ps.push(wire::json_hash_path(Merkle::open_tree(nxt.t, c_indices[s])));
        }
        a_indices.insert(a_indices.end(), b_indices.begin(), b_indices.end());
        return a_indices;
    }

    std::vector<size_t> prove(const std::vector<FieldElement>& codeword, ProofStream& ps) const   // :212-254
    {
        return prove_on(commit_device(codeword, ps), ps);
    }
    // prove for a codeword that already lives on the device (packed, domain_length elements)
    std::vector<size_t> prove_resident(const uint64_t* codeword_dev, ProofStream& ps) const
    {
        return prove_on(commit_resident(codeword_dev, ps), ps);
    }
    std::vector<size_t> prove_on(std::unique_ptr<LayersHandle> h, ProofStream& ps) const
    {
        size_t layers = stark_fri_layers_count(h->l);
        std::vector<size_t> top = sample_indices(ps.prover_fiat_shamir(32), stark_fri_layer_len(h->l, 0) / 2,
                                                 stark_fri_layer_len(h->l, layers - 1), num_colinearity_tests);
        // Fri::query per round (fri.rs:229-238).  query() returns a and b for EVERY index it was given, so the
        // reference's list doubles each round (:160-163,207-208) although only its first num_colinearity_tests
        // entries are ever read (:166-203): round i reads top[s] mod len(i)/2.  All rounds' reads are therefore
        // known now and come back in one round trip; the pushes keep the reference's order.
        const size_t t = num_colinearity_tests;
        if (layers > 1 && t > 0) {
            std::vector<uint64_t> top64(top.begin(), top.begin() + t), vals((layers - 1) * 6 * t);
            size_t digests = 0;
            std::vector<size_t> depth(layers);
            for (size_t i = 0; i < layers; i++) {
                size_t len = stark_fri_layer_len(h->l, i), d = 0;
                while (((size_t)1 << d) < len) d++;
                depth[i] = d;
            }
            for (size_t i = 0; i + 1 < layers; i++) digests += t * (2 * depth[i] + depth[i + 1]);
            std::vector<uint8_t> raw(32 * digests + 32);
            check(stark_fri_query_all(h->l, top64.data(), t, vals.data(), raw.data()));
            std::vector<FieldElement> e = unpack(vals);
            size_t at = 0;
            for (size_t i = 0; i + 1 < layers; i++) {
                const FieldElement* er = e.data() + i * 3 * t;
                for (size_t s = 0; s < t; s++)                                                   // :166-179
                    ps.push(wire::json_string_triple(er[s].to_string(), er[t + s].to_string(), er[2 * t + s].to_string()));
                auto path_at = [&](size_t base, size_t d, size_t s) {                           // the digests as they came back
                    return wire::json_hash_path_bytes(raw.data() + 32 * (at + base + s * d), d);
                };
                for (size_t s = 0; s < t; s++) {                                                 // :182-204
                    ps.push(path_at(0, depth[i], s));
                    ps.push(path_at(t * depth[i], depth[i], s));
                    ps.push(path_at(2 * t * depth[i], depth[i + 1], s));
                }
                at += t * (2 * depth[i] + depth[i + 1]);
            }
        }
        std::vector<size_t> out = top;
        for (size_t idx : top) out.push_back(idx + stark_fri_layer_len(h->l, 0) / 2);
        return out;
    }

    bool verify(ProofStream& ps, std::vector<std::pair<size_t, FieldElement>>& polynomial_values) const   // :256-417
    {
        FieldElement om = omega, off = offset;
        size_t rounds = num_rounds();
        std::vector<std::string> roots;
        std::vector<FieldElement> alphas;
        for (size_t r = 0; r < rounds; r++) {
            roots.push_back(ps.pull());
            alphas.push_back(FieldElement::sample(ps.verifier_fiat_shamir(32)));
        }
        std::vector<FieldElement> last_codeword = wire::json_field_vec_parse(ps.pull());
        std::vector<std::vector<uint8_t>> last_leaves;
        for (const FieldElement& x : last_codeword) last_leaves.push_back(wire::bincode_field_element(x));
        HashOutput lroot = Merkle::commit_host(last_leaves);                                 // :274-276, host: short input
        if (roots.back() != wire::hex_encode(lroot.data(), 32)) return false;
        size_t degree = (last_codeword.size() / expansion_factor) - 1;
        FieldElement last_omega = om, last_offset = off;
        for (size_t r = 0; r + 1 < rounds; r++) {
            last_omega = last_omega * last_omega;
            last_offset = last_offset * last_offset;
        }
        if (!(last_omega.inverse() == last_omega.pow(last_codeword.size() - 1))) throw Panic("assertion failed");
        std::vector<FieldElement> last_domain;
        for (size_t i = 0; i < last_codeword.size(); i++) last_domain.push_back(last_offset * last_omega.pow(i));
        Polynomial poly = Polynomial::lagrange(last_domain, last_codeword);
        for (size_t i = 0; i < last_domain.size(); i++)                                      // :304-305 (Horner: shifted domain)
            if (!(poly.eval(last_domain[i]) == last_codeword[i])) throw Panic("last codeword is not low degree");
        if (poly.degree() > degree) return false;
        std::vector<size_t> top = sample_indices(ps.verifier_fiat_shamir(32), domain_length >> 1,
                                                 domain_length >> (rounds - 1), num_colinearity_tests);
        for (size_t r = 0; r + 1 < rounds; r++) {
            std::vector<size_t> c_indices, a_indices, b_indices;
            for (size_t idx : top) c_indices.push_back(idx % (domain_length >> (r + 1)));
            for (size_t idx : c_indices) {
                a_indices.push_back(idx);
                b_indices.push_back(idx + (domain_length >> (r + 1)));
            }
            std::vector<FieldElement> aa, bb, cc;
            for (size_t s = 0; s < num_colinearity_tests; s++) {
                std::array<std::string, 3> tri = wire::json_string_triple_parse(ps.pull());
                FieldElement ay = FieldElement::from_decimal(tri[0]), by = FieldElement::from_decimal(tri[1]),
                             cy = FieldElement::from_decimal(tri[2]);
                aa.push_back(ay);
                bb.push_back(by);
                cc.push_back(cy);
                if (r == 0) {
                    polynomial_values.emplace_back(a_indices[s], ay);
                    polynomial_values.emplace_back(b_indices[s], by);
                }
                FieldElement ax = off * om.pow(a_indices[s]), bx = off * om.pow(b_indices[s]), cx = alphas[r];
                if (!Polynomial::test_colinearity({{ax, ay}, {bx, by}, {cx, cy}})) return false;
            }
            HashOutput root_r{}, root_n{};
            std::vector<uint8_t> rr = wire::hex_decode(roots[r]), rn = wire::hex_decode(roots[r + 1]);
            std::memcpy(root_r.data(), rr.data(), 32);
            std::memcpy(root_n.data(), rn.data(), 32);
            for (size_t i = 0; i < num_colinearity_tests; i++) {
                std::vector<HashOutput> path = wire::json_hash_path_parse(ps.pull());
                if (!Merkle::verify(root_r, a_indices[i], path, wire::bincode_field_element(aa[i]))) return false;
                path = wire::json_hash_path_parse(ps.pull());
                if (!Merkle::verify(root_r, b_indices[i], path, wire::bincode_field_element(bb[i]))) return false;
                path = wire::json_hash_path_parse(ps.pull());
                if (!Merkle::verify(root_n, c_indices[i], path, wire::bincode_field_element(cc[i]))) return false;
            }
            om = om * om;
            off = off * off;
        }
        return true;
    }

private:
    static int32_t on_root(void* user, uint32_t, int32_t last, const uint8_t root[32], uint64_t alpha_out[2])
    {
        ProofStream* ps = static_cast<ProofStream*>(user);
        ps->push(wire::hex_encode(root, 32));                                                // :120
        if (!last) {
            FieldElement a = FieldElement::sample(ps->prover_fiat_shamir(32));               // :126-127
            alpha_out[0] = (uint64_t)a.value;
            alpha_out[1] = (uint64_t)(a.value >> 64);
        }
        return 0;
    }
    static std::vector<FieldElement> download(stark_fri_layers* l, size_t layer)
    {
        std::vector<uint64_t> buf(2 * stark_fri_layer_len(l, layer));
        check(stark_fri_layer_download(l, layer, buf.data()));
        return unpack(buf);
    }
    static std::vector<std::vector<HashOutput>> paths_from(const std::vector<uint8_t>& buf, size_t count, size_t depth)
    {
        std::vector<std::vector<HashOutput>> out(count, std::vector<HashOutput>(depth));
        for (size_t i = 0; i < count; i++)
            for (size_t k = 0; k < depth; k++) std::memcpy(out[i][k].data(), buf.data() + 32 * (i * depth + k), 32);
        return out;
    }
    static std::vector<std::vector<HashOutput>> open_layer_many(stark_fri_layers* l, size_t layer,
                                                                const std::vector<uint64_t>& indices)
    {
        std::vector<uint8_t> buf(indices.size() * 64 * 32 + 32);
        size_t depth = 0;
        check(stark_fri_layer_open_many(l, layer, indices.data(), indices.size(), buf.data(), &depth));
        return paths_from(buf, indices.size(), depth);
    }
    static std::vector<HashOutput> open_layer(stark_fri_layers* l, size_t layer, size_t index)
    {
        uint8_t buf[64 * 32];
        size_t depth = 0;
        check(stark_fri_layer_open(l, layer, index, buf, &depth));
        std::vector<HashOutput> path(depth);
        for (size_t k = 0; k < depth; k++) std::memcpy(path[k].data(), buf + 32 * k, 32);
        return path;
    }
};

}  // namespace stark
