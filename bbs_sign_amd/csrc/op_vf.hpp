// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

// ---- verify ------------------------------------------------------------------------------------
template <class C>
struct VfJob : JobBase<C> {
    using JobBase<C>::JobBase;
    VfArgs<C> a{};
    PairArgs<C> pa{};
    BvState<C> bv{};                  // batch verification only
};

template <class C>
int vf_upload(Ctx<C>* ctx, size_t n, const uint8_t* sigs, const uint8_t* msgs, const uint64_t* msg_off,
                     const uint8_t* headers, const uint64_t* hdr_off, bbs_job** out) {
    constexpr int N = C::FpP::N;
    constexpr int NC = C::FpP::NC;
    constexpr int FPB = 4 * NC;
    using R = typename C::FrP;
    if (!ctx->gens_set || !ctx->pk_set) return BBS_E_STATE;
    if (!out || (n && (!sigs || !msg_off))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    const size_t rec = 2 * FPB + 32;
    auto job = std::unique_ptr<VfJob<C>>(new VfJob<C>(ctx));
    job->n = n;
    job->status0.assign(n, ST_PENDING);
    Soa sa, se, sm;
    sa.init(2 * NC, n); se.init(8, n); sm.init((size_t)std::max(L, 1) * 8, n);
    for (size_t i = 0; i < n; i++) {
        int8_t& st = job->status0[i];
        const size_t l = (size_t)(msg_off[i + 1] - msg_off[i]);
        if (l != (size_t)L) { st = BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH; continue; }   // verify.rs:69-71
        if (ctx->dst_too_long) { st = BBS_ST_PANIC_DST_TOO_LONG; continue; }
        bool ok = pack_g1<C>(sa, 0, i, sigs + i * rec);
        ok &= pack_fe<R>(se, 0, i, sigs + i * rec + 2 * FPB);
        for (size_t j = 0; j < l; j++) ok &= pack_fe<R>(sm, j * 8, i, msgs + (msg_off[i] + j) * 32);
        if (!ok) st = BBS_ST_NONCANONICAL;
    }
    BytePool hp;
    if (!hp.build(n, headers, hdr_off)) return BBS_E_ARG;
    int rc = BBS_OK;
    VfArgs<C>& a = job->a;
    a.n = n; a.L = L; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    a.glv = (C::K::HAS_GLV && (C::K::GLV_ALWAYS || ctx->points_in_subgroup)) ? 1 : 0;
    a.sig_a = job->up(sa.soa(), rc); a.sig_e = job->up(se.soa(), rc); a.msgs = job->up(sm.soa(), rc);
    a.hdr_off = job->up(hp.off, rc); a.hdr_len = job->up(hp.len, rc); a.hdr_bytes = job->up(hp.bytes, rc);
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)VF_NPARTS * 3 * N * n, rc);
    a.aff = job->template scratch<uint32_t>((size_t)2 * 2 * N * n, rc);
    a.fmiller = job->template scratch<uint32_t>((size_t)2 * 12 * N * n, rc);
    if (rc) return rc;
    if ((rc = job->finish_setup())) return rc;
    a.status = job->d_status.template as<int8_t>();
    PairArgs<C>& pa = job->pa;
    pa.n = n; pa.cc = a.cc; pa.pa = a.aff; pa.pb = a.aff + (size_t)2 * N * n; pa.negate_b = 0;
    pa.canonical = 0; pa.gate_arr = a.status; pa.gate = ST_PAIRING; pa.out = a.status; pa.fmiller = a.fmiller;
    VfJob<C>* j = job.get();
    j->stages.push_back({"vf_scalars", [j]() { return rt::launch<VfScalars<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"vf_msm_parts", [j]() { return rt::launch<VfMsmPart<C>>(j->stream(), j->a, j->n * VF_NPARTS); }});
    j->stages.push_back({"vf_combine", [j]() { return rt::launch<VfCombine<C>>(j->stream(), j->a, j->n); }});
    if (!ctx->batch_verify) {
        add_pairing_stages<C>(j, &j->pa, 0, "pair_miller", "pair_final_exp", "pairing_6lane");
    } else {
        // e(A, W) e(e A - B, BP2) == 1 for all pending items at once: A and e A - B are in a.aff (Montgomery)
        if ((rc = add_batch_verification<C>(j, &j->bv, ctx, n, a.cc, a.status, a.aff, a.aff + (size_t)2 * N * n, 0, &j->pa))) return rc;
    }
    *out = job.release();
    return BBS_OK;
}

