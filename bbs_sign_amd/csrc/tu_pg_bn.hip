// explicit instantiation: pg for BnCurve
#include "op_pg.hpp"
template int pg_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint8_t*, const uint64_t*);
