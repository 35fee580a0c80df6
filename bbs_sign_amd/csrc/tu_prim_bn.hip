// explicit instantiation: prim for BnCurve
#include "op_prim.hpp"
template int Ctx<BnCurve>::set_generators(const uint8_t*, size_t, const uint8_t*, size_t);
template int h2s_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, size_t, uint8_t*);
template int msm_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, size_t, const uint8_t*, const uint8_t*, size_t, uint8_t*, int8_t*);
template int pairing_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, int8_t*);
template int selftest_f12<BnCurve>(Ctx<BnCurve>*, int, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
template int msm_pippenger<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, uint8_t*, int*, int8_t*);
template int proofs_from_octets_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint64_t*, uint8_t*, uint8_t*, uint64_t*, int8_t*);
template int g1_decompress_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, uint8_t*, int8_t*);
template int signatures_from_octets_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, uint8_t*, int8_t*);
