#pragma once
#include "runtime.hpp"
// declarations of the per-operation templates (defined in op_*.hpp, instantiated in tu_*.hip)
template <class C> int pv_upload(Ctx<C>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
template <class C> int vf_upload(Ctx<C>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
template <class C> int sg_upload(Ctx<C>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
template <class C> int pg_upload(Ctx<C>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
template <class C> int h2s_batch(Ctx<C>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, size_t, uint8_t*);
template <class C> int msm_batch(Ctx<C>*, size_t, const uint8_t*, size_t, const uint8_t*, const uint8_t*, size_t, uint8_t*, int8_t*);
template <class C> int selftest_f12(Ctx<C>*, int, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
template <class C> int pairing_batch(Ctx<C>*, size_t, const uint8_t*, const uint8_t*, int8_t*);
template <class C> int msm_pippenger(Ctx<C>*, size_t, const uint8_t*, const uint8_t*, uint8_t*, int*, int8_t*);
template <class C> int g1_decompress_batch(Ctx<C>*, size_t, const uint8_t*, uint8_t*, int8_t*);
template <class C> int proofs_from_octets_batch(Ctx<C>*, size_t, const uint8_t*, const uint64_t*, uint8_t*, uint8_t*, uint64_t*, int8_t*);
extern template int pv_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int pv_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int vf_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int vf_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int sg_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int sg_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int pg_upload<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int pg_upload<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint64_t*, bbs_job**);
extern template int Ctx<BlsCurve>::set_generators(const uint8_t*, size_t, const uint8_t*, size_t);
extern template int h2s_batch<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, size_t, uint8_t*);
extern template int msm_batch<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, size_t, const uint8_t*, const uint8_t*, size_t, uint8_t*, int8_t*);
extern template int pairing_batch<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, int8_t*);
extern template int msm_pippenger<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint8_t*, uint8_t*, int*, int8_t*);
extern template int g1_decompress_batch<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, uint8_t*, int8_t*);
extern template int proofs_from_octets_batch<BlsCurve>(Ctx<BlsCurve>*, size_t, const uint8_t*, const uint64_t*, uint8_t*, uint8_t*, uint64_t*, int8_t*);
extern template int Ctx<BnCurve>::set_generators(const uint8_t*, size_t, const uint8_t*, size_t);
extern template int h2s_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint64_t*, const uint8_t*, size_t, uint8_t*);
extern template int msm_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, size_t, const uint8_t*, const uint8_t*, size_t, uint8_t*, int8_t*);
extern template int pairing_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, int8_t*);
extern template int msm_pippenger<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint8_t*, uint8_t*, int*, int8_t*);
extern template int g1_decompress_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, uint8_t*, int8_t*);
extern template int proofs_from_octets_batch<BnCurve>(Ctx<BnCurve>*, size_t, const uint8_t*, const uint64_t*, uint8_t*, uint8_t*, uint64_t*, int8_t*);
extern template int selftest_f12<BlsCurve>(Ctx<BlsCurve>*, int, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
extern template int selftest_f12<BnCurve>(Ctx<BnCurve>*, int, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*);
