// explicit instantiation: sg for BlsCurve
#include "op_sg.hpp"
template int sg_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint64_t*);
