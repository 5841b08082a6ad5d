// Host build of the lazy 5-limb field core (stark_amd/csrc/field5.hpp) so the
// CPU test-suite can check it against big-integer arithmetic without a GPU.
#include "../stark_amd/csrc/field5.hpp"
using namespace starkf;
extern "C" {
void t_mont_mul5(const int32_t* a, const int32_t* w, int32_t* out)
{
    fe5 x, y;
    for (int i = 0; i < 5; i++) { x.l[i] = a[i]; y.l[i] = w[i]; }
    fe5 r = mont_mul5(x, y);
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
}
