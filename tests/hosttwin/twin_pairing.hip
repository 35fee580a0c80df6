// Host twin (TEST ONLY) for the pairing primitive.
#include "../../bbs_sign_amd/csrc/host_g2.hpp"
#include <cstring>
using namespace bbs;
template <class P> static Fe<P> ld(const uint32_t* p) { return fe_from_limbs<P>(p); }

template <class C> static int pairing2(const uint32_t* pa, const uint32_t* qa, int qa_inf, const uint32_t* pb, const uint32_t* qb, int qb_inf) {
    constexpr int N = C::FpP::N; using P = typename C::FpP;
    G1Aff<C> Pa = {ld<P>(pa), ld<P>(pa + N)}, Pb = {ld<P>(pb), ld<P>(pb + N)};
    G2Aff<C> Qa = {{ld<P>(qa), ld<P>(qa + N)}, {ld<P>(qa + 2 * N), ld<P>(qa + 3 * N)}, qa_inf != 0};
    G2Aff<C> Qb = {{ld<P>(qb), ld<P>(qb + N)}, {ld<P>(qb + 2 * N), ld<P>(qb + 3 * N)}, qb_inf != 0};
    if (!g2_on_curve<C>(Qa) || !g2_on_curve<C>(Qb)) return -2;
    static LineTable<C> ta, tb; static MillerSchedule s;
    build_schedule<C>(s);
    if (!build_line_table<C>(Qa, ta) || !build_line_table<C>(Qb, tb)) return -1;
    return pairing_product2_is_one<C>(&s, &ta, Pa, &tb, Pb) ? 1 : 0;
}
extern "C" int twin_pairing2(int curve, const uint32_t* pa, const uint32_t* qa, int qa_inf, const uint32_t* pb, const uint32_t* qb, int qb_inf) {
    return curve == 0 ? pairing2<BlsCurve>(pa, qa, qa_inf, pb, qb, qb_inf) : pairing2<BnCurve>(pa, qa, qa_inf, pb, qb, qb_inf);
}
