// explicit instantiation: vf for BlsCurve
#include "op_vf.hpp"
template int vf_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint8_t*, const uint64_t*);
