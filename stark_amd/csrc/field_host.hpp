// Host-side scalar field helpers used to build twiddle tables and scalars.
// (reference: src/modules/field.rs:42-116)
#pragma once
#include "field.hpp"

namespace starkf {

struct HostField {
    fe r2;        // 2^256 mod p  (Montgomery form of 2^128)
    fe one_m;     // Montgomery form of 1
    HostField()
    {
        fe r = fe{1, 0};
        for (int i = 0; i < 256; i++) r = fe_add(r, r);
        r2 = r;
        one_m = to_mont(fe{1, 0});
    }
    fe to_mont(fe a) const { return mont_mul(a, r2); }
    fe from_mont(fe a) const { return mont_mul(a, fe{1, 0}); }
    fe mul(fe a, fe b) const { return mont_mul(mont_mul(a, b), r2); }
    fe pow(fe base, u128 exp) const
    {
        fe result = fe{1, 0};
        while (exp) {
            if (exp & 1) result = mul(result, base);
            base = mul(base, base);
            exp >>= 1;
        }
        return result;
    }
    // field.rs:67-93 (extended Euclid) yields a^(p-2) for a != 0 and 0 for 0.
    fe inverse(fe a) const
    {
        if (fe_is_zero(a)) return fe_zero();
        u128 p = ((u128)P_HI << 64) | P_LO;
        return pow(a, p - 2);
    }
    fe generator() const { return fe{GEN_LO, GEN_HI}; }
    // field.rs:96-107
    fe primitive_nth_root(unsigned log_n) const
    {
        fe root = generator();
        for (unsigned k = 119; k > log_n; k--) root = mul(root, root);
        return root;
    }
};

}  // namespace starkf
