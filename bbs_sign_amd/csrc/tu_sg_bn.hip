// explicit instantiation: sg for BnCurve
#include "op_sg.hpp"
template int sg_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint64_t*);
