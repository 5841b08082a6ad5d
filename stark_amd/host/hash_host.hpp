// Host-side hashes used above the C ABI: BLAKE2b (RFC 7693; the reference's
// blake2 crate, merkle.rs:29,43 / fri.rs:60 / stark.rs:210) for Merkle::verify
// and index sampling, and SHAKE256 (FIPS 202; the reference's sha3 crate,
// proof_stream.rs:52,63) for the Fiat-Shamir transcript.  Small inputs only:
// the bulk hashing runs on the device.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

namespace stark {
namespace hash {

class Blake2b {
public:
    explicit Blake2b(size_t outlen) : outlen_(outlen)
    {
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                       0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                       0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
        for (int i = 0; i < 8; i++) h_[i] = iv[i];
        h_[0] ^= 0x01010000ull ^ (uint64_t)outlen;
    }
    void update(const uint8_t* p, size_t len)
    {
        while (len) {
            if (fill_ == 128) {               // a full buffer is only compressed once more data follows
                t_ += 128;
                compress(false);
                fill_ = 0;
            }
            size_t k = 128 - fill_;
            if (k > len) k = len;
            std::memcpy(buf_ + fill_, p, k);
            fill_ += k;
            p += k;
            len -= k;
        }
    }
    void update(const std::vector<uint8_t>& v) { update(v.data(), v.size()); }
    std::vector<uint8_t> finalize()
    {
        t_ += fill_;
        std::memset(buf_ + fill_, 0, 128 - fill_);
        compress(true);
        std::vector<uint8_t> out(outlen_);
        std::memcpy(out.data(), h_, outlen_);
        return out;
    }

private:
    static uint64_t rotr(uint64_t x, int r) { return (x >> r) | (x << (64 - r)); }
    void compress(bool last)
    {
        static const uint8_t sigma[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                       0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                       0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
        uint64_t m[16], v[16];
        std::memcpy(m, buf_, 128);
        for (int i = 0; i < 8; i++) { v[i] = h_[i]; v[i + 8] = iv[i]; }
        v[12] ^= t_;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = sigma[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h_[i] ^= v[i] ^ v[i + 8];
    }
    uint64_t h_[8];
    uint64_t t_ = 0;
    uint8_t buf_[128];
    size_t fill_ = 0;
    size_t outlen_;
};

inline std::vector<uint8_t> blake2b(const uint8_t* p, size_t len, size_t outlen)
{
    Blake2b h(outlen);
    h.update(p, len);
    return h.finalize();
}

// Keccak-f[1600] (FIPS 202, 3.2-3.3).  The lane walk of rho and pi is generated at compile time from the
// specification ((x, y) -> (y, 2x + 3y), rotation (t + 1)(t + 2) / 2), and every loop has a constant trip
// count and is unrolled, so the 25 lanes stay in registers: the Fiat-Shamir transcript hashes the whole
// stream again for every challenge (proof_stream.rs:50-58), some hundred permutations per FRI proof, all
// of them between two device launches that wait for the result.
struct KeccakWalk {
    int lane[24];
    int rot[24];
};
constexpr KeccakWalk keccak_walk()
{
    KeccakWalk w{};
    int x = 1, y = 0;
    for (int t = 0; t < 24; t++) {
        const int nx = y, ny = (2 * x + 3 * y) % 5;     // where lane (x, y) goes
        w.lane[t] = nx + 5 * ny;
        w.rot[t] = ((t + 1) * (t + 2) / 2) % 64;        // by how much lane (x, y) is rotated on the way
        x = nx;
        y = ny;
    }
    return w;
}

#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("O3", "unroll-loops")))
#endif
inline void keccak_f1600(uint64_t (&a)[25])
{
    static constexpr uint64_t rc[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
        0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
        0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
        0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    constexpr KeccakWalk W = keccak_walk();
    uint64_t s[25];
    for (int i = 0; i < 25; i++) s[i] = a[i];
    for (int round = 0; round < 24; round++) {
        uint64_t c[5];
#pragma GCC unroll 5
        for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
#pragma GCC unroll 5
        for (int x = 0; x < 5; x++) {
            const uint64_t n = c[(x + 1) % 5];
            const uint64_t d = c[(x + 4) % 5] ^ ((n << 1) | (n >> 63));
#pragma GCC unroll 5
            for (int y = 0; y < 25; y += 5) s[x + y] ^= d;
        }
        uint64_t carry = s[1];                          // rho and pi along the walk that starts at lane (1, 0)
#pragma GCC unroll 24
        for (int t = 0; t < 24; t++) {
            const uint64_t next = s[W.lane[t]];
            s[W.lane[t]] = (carry << W.rot[t]) | (carry >> (64 - W.rot[t]));
            carry = next;
        }
#pragma GCC unroll 5
        for (int y = 0; y < 25; y += 5) {
            const uint64_t b0 = s[y], b1 = s[y + 1], b2 = s[y + 2], b3 = s[y + 3], b4 = s[y + 4];
            s[y] = b0 ^ (~b1 & b2);
            s[y + 1] = b1 ^ (~b2 & b3);
            s[y + 2] = b2 ^ (~b3 & b4);
            s[y + 3] = b3 ^ (~b4 & b0);
            s[y + 4] = b4 ^ (~b0 & b1);
        }
        s[0] ^= rc[round];
    }
    for (int i = 0; i < 25; i++) a[i] = s[i];
}

inline void shake256_into(const uint8_t* in, size_t len, uint8_t* out, size_t outlen)
{
    const size_t rate = 136;
    uint64_t s[25] = {0};
    uint8_t* sb = reinterpret_cast<uint8_t*>(s);
    while (len >= rate) {
        uint64_t block[17];
        std::memcpy(block, in, rate);
        for (int i = 0; i < 17; i++) s[i] ^= block[i];
        keccak_f1600(s);
        in += rate;
        len -= rate;
    }
    for (size_t i = 0; i < len; i++) sb[i] ^= in[i];
    sb[len] ^= 0x1F;
    sb[rate - 1] ^= 0x80;
    keccak_f1600(s);
    size_t off = 0;
    while (off < outlen) {
        size_t k = outlen - off < rate ? outlen - off : rate;
        std::memcpy(out + off, sb, k);
        off += k;
        if (off < outlen) keccak_f1600(s);
    }
}

inline std::vector<uint8_t> shake256(const uint8_t* in, size_t len, size_t outlen)
{
    std::vector<uint8_t> out(outlen);
    shake256_into(in, len, out.data(), outlen);
    return out;
}

}  // namespace hash
}  // namespace stark
