// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

// ---- sign --------------------------------------------------------------------------------------
template <class C>
struct SgJob : JobBase<C> {
    using JobBase<C>::JobBase;
    SgArgs<C> a{};
    int fetch_signatures(uint8_t* out) override {
        constexpr int N = C::FpP::NC;       // canonical words
        if (this->use() || rt::sync(this->stream())) return BBS_E_HIP;
        const size_t n = this->n;
        std::vector<uint32_t> A((size_t)2 * N * n), E((size_t)8 * n);
        if (this->down(A, a.out_a) || this->down(E, a.out_e)) return BBS_E_HIP;
        std::vector<int8_t> st(n);
        if (rt::d2h(st.data(), this->d_status.p, n, this->stream())) return BBS_E_HIP;
        const size_t rec = 8 * N + 32;
        for (size_t i = 0; i < n; i++) {
            if (st[i] != 1) { std::memset(out + i * rec, 0, rec); continue; }
            unpack_words_le(A, n, 0, i, 2 * N, out + i * rec);
            unpack_words_le(E, n, 0, i, 8, out + i * rec + 8 * N);
        }
        return BBS_OK;
    }
};

template <class C>
int sg_upload(Ctx<C>* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* headers,
                     const uint64_t* hdr_off, bbs_job** out) {
    constexpr int N = C::FpP::N;
    using R = typename C::FrP;
    if (!ctx->gens_set || !ctx->sk_set) return BBS_E_STATE;
    if (!out || (n && !msg_off)) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    auto job = std::unique_ptr<SgJob<C>>(new SgJob<C>(ctx));
    job->n = n;
    job->status0.assign(n, ST_PENDING);
    Soa sm;
    sm.init((size_t)std::max(L, 1) * 8, n);
    for (size_t i = 0; i < n; i++) {
        int8_t& st = job->status0[i];
        const size_t l = (size_t)(msg_off[i + 1] - msg_off[i]);
        if (l != (size_t)L) { st = BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH; continue; }   // sign.rs:77-79
        if (ctx->dst_too_long) { st = BBS_ST_PANIC_DST_TOO_LONG; continue; }
        bool ok = true;
        for (size_t j = 0; j < l; j++) ok &= pack_fe<R>(sm, j * 8, i, msgs + (msg_off[i] + j) * 32);
        if (!ok) st = BBS_ST_NONCANONICAL;
    }
    BytePool hp;
    if (!hp.build(n, headers, hdr_off)) return BBS_E_ARG;
    int rc = BBS_OK;
    SgArgs<C>& a = job->a;
    a.n = n; a.L = L; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    std::memcpy(a.sk, ctx->sk, sizeof(a.sk));
    a.msgs = job->up(sm.soa(), rc);
    a.hdr_off = job->up(hp.off, rc); a.hdr_len = job->up(hp.len, rc); a.hdr_bytes = job->up(hp.bytes, rc);
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)NFIX * 3 * N * n, rc);
    a.out_a = job->template scratch<uint32_t>((size_t)2 * C::FpP::NC * n, rc);
    a.out_e = job->template scratch<uint32_t>((size_t)8 * n, rc);
    if (rc) return rc;
    if ((rc = job->finish_setup())) return rc;
    a.status = job->d_status.template as<int8_t>();
    SgJob<C>* j = job.get();
    j->stages.push_back({"sg_scalars", [j]() { return rt::launch<SgScalars<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"sg_msm_parts", [j]() { return rt::launch<SgMsmPart<C>>(j->stream(), j->a, j->n * NFIX); }});
    j->stages.push_back({"sg_combine", [j]() { return rt::launch<SgCombine<C>>(j->stream(), j->a, j->n); }});
    *out = job.release();
    return BBS_OK;
}

