// Host twin (TEST ONLY): compiles the product's __host__ __device__ arithmetic for x86 so that
// logic can be checked against the oracle in the GPU-less container.  Never loaded by the product.
#include "../../bbs_sign_amd/csrc/g1.hpp"
#include <cstring>
using namespace bbs;

template <class P> static Fe<P> ld(const uint32_t* p) { Fe<P> r; for (int i = 0; i < P::N; i++) r.v[i] = p[i]; return fe_from_limbs<P>(r.v); }
template <class P> static void st(uint32_t* p, const Fe<P>& a) { Fe<P> c = fe_to_canonical<P>(a); for (int i = 0; i < P::N; i++) p[i] = c.v[i]; }

template <class P> static void fieldop(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    Fe<P> x = ld<P>(a), y = ld<P>(b), r;
    switch (op) {
        case 0: r = fe_mul<P>(x, y); break;
        case 1: r = fe_add<P>(x, y); break;
        case 2: r = fe_sub<P>(x, y); break;
        case 3: r = fe_inv<P>(x); break;
        case 4: r = fe_neg<P>(x); break;
        default: r = fe_sqr<P>(x);
    }
    st<P>(out, r);
}

template <class C> static void g1mul(const uint32_t* xy, const uint32_t* k, uint32_t* out) {
    constexpr int N = C::FpP::N;
    G1Aff<C> p = {ld<typename C::FpP>(xy), ld<typename C::FpP>(xy + N)};
    G1Aff<C> r = g1j_to_aff<C>(g1_mul_aff<C>(p, k));
    st<typename C::FpP>(out, r.x); st<typename C::FpP>(out + N, r.y);
}

extern "C" {
// which: 0 bls fp, 1 bls fr, 2 bn fp, 3 bn fr
void twin_fieldop(int which, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    switch (which) {
        case 0: fieldop<BlsFpParams>(op, a, b, out); break;
        case 1: fieldop<BlsFrParams>(op, a, b, out); break;
        case 2: fieldop<BnFpParams>(op, a, b, out); break;
        default: fieldop<BnFrParams>(op, a, b, out);
    }
}
void twin_g1_mul(int curve, const uint32_t* xy, const uint32_t* k, uint32_t* out) {
    if (curve == 0) g1mul<BlsCurve>(xy, k, out); else g1mul<BnCurve>(xy, k, out);
}
}
