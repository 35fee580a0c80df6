// explicit instantiation: pg for BlsCurve
#include "op_pg.hpp"
template int pg_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**, const uint8_t*, const uint8_t*, const uint64_t*);
