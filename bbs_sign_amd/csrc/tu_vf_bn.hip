// explicit instantiation: vf for BnCurve
#include "op_vf.hpp"
template int vf_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint8_t*, const uint64_t*);
