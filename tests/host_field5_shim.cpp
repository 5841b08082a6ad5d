// Host build of the lazy 5-limb field core (stark_amd/csrc/field5.hpp) so the
// CPU test-suite can check it against big-integer arithmetic without a GPU.
#include "../stark_amd/csrc/field5.hpp"
using namespace starkf;
extern "C" {
void t_mont_mul5(const int32_t* a, const int32_t* w, int32_t* out, int plus_p)
{
    fe5 x, y;
    for (int i = 0; i < 5; i++) { x.l[i] = a[i]; y.l[i] = w[i]; }
    fe5 r = plus_p ? mont_mul5<true>(x, y) : mont_mul5<false>(x, y);
    for (int i = 0; i < 5; i++) out[i] = r.l[i];
}
void t_reduce5(const int32_t* a, uint64_t* out)
{
    fe5 x;
    for (int i = 0; i < 5; i++) x.l[i] = a[i];
    fe r = reduce5_slow(x);
    out[0] = r.lo; out[1] = r.hi;
}
void t_pack_canonical(const int32_t* a, uint64_t* out, uint32_t* flagged)
{
    fe5 x;
    for (int i = 0; i < 5; i++) x.l[i] = a[i];
    *flagged = noncanonical5(x);
    fe r = pack_canonical(x);
    out[0] = r.lo; out[1] = r.hi;
}
void t_unpack5(const uint64_t* v, int32_t* out)
{
    fe5 r = unpack5(fe{v[0], v[1]});
    for (int i = 0; i < 5; i++) out[i] = r.l[i];
}
// w25: 25 limbs, [k][i] = limb k of (w * 2^(26 i + 52) mod p)
void t_mont_mul5_c(const int32_t* a, const int32_t* w25, int32_t* out, int plus_p)
{
    fe5 x;
    tw25 w{};
    for (int i = 0; i < 5; i++) x.l[i] = a[i];
    for (int k = 0; k < 5; k++)
        for (int i = 0; i < 5; i++) w.l[k][i] = w25[5 * k + i];
    fe5 r = plus_p ? mont_mul5_c<true>(x, w) : mont_mul5_c<false>(x, w);
    for (int i = 0; i < 5; i++) out[i] = r.l[i];
}
}
