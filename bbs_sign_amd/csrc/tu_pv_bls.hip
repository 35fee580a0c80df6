// explicit instantiation: pv for BlsCurve
#include "op_pv.hpp"
template int pv_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*);
