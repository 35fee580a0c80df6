// explicit instantiation: pv for BnCurve
#include "op_pv.hpp"
template int pv_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*);
