// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

// ---- sign --------------------------------------------------------------------------------------
template <class C>
struct SgJob : JobBase<C> {
    using JobBase<C>::JobBase;
    SgArgs<C> a{};
    VfIngestArgs<C> ingest{};
    MsgHashArgs mh{};                 // raw-message form only
    // the records come off the device in the caller's layout (stage SgEmit): delivery is one copy
    size_t rec_bytes() const { return a.oct_form ? (size_t)(4 * C::FpP::NC + 32) : (size_t)(8 * C::FpP::NC + 32); }
    int set_octet_form() override { a.oct_form = 1; return BBS_OK; }
    int fetch_signatures(uint8_t* out) override {
        if (this->use() || rt::sync(this->stream())) return BBS_E_HIP;
        if (int rc = this->require_decided()) return rc;        // a job that never ran holds no records (fail closed)
        return rt::d2h(out, a.out_rec, this->n * rec_bytes(), this->stream()) ? BBS_E_HIP : BBS_OK;
    }
    // submit form
    uint8_t* sigs_to = nullptr;
    HostBuf h_out;
    void set_result_targets(uint8_t* sigs, uint8_t*, uint64_t*) override { sigs_to = sigs; }
    int enqueue_result_fetch() override {
        if (!this->n) return BBS_OK;
        if (!h_out.p && h_out.alloc(this->n * rec_bytes())) return BBS_E_NOMEM;
        return rt::d2h_async(h_out.p, a.out_rec, this->n * rec_bytes(), this->stream()) ? BBS_E_HIP : BBS_OK;
    }
    int deliver() override {
        if (int rc = JobBase<C>::deliver()) return rc;
        if (sigs_to && this->n) {
            if (!h_out.p) return BBS_E_STATE;
            std::memcpy(sigs_to, h_out.p, this->n * rec_bytes());
        }
        return BBS_OK;
    }
};

template <class C>
int sg_upload(Ctx<C>* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* headers,
                     const uint64_t* hdr_off, bbs_job** out, const uint8_t* msg_bytes, const uint64_t* msg_byte_off) {
    // msg_byte_off != nullptr: raw messages, hashed to scalars on the device (see vf_upload)
    constexpr int N = C::FpP::N;
    if (!ctx->gens_set || !ctx->sk_set) return BBS_E_STATE;
    if (!out || (n && !msg_off)) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    auto job = std::unique_ptr<SgJob<C>>(new SgJob<C>(ctx));
    job->n = n;
    // staging image + device-side checks (stage VfIngest without a signature record: sign.rs:77-79's length check,
    // range checks of the messages, SoA transposition)
    const bool raw = msg_byte_off != nullptr;
    RaggedIn ms{msg_off, msgs, 32}, hb{hdr_off, headers, 1};
    ms.offsets_only = raw;
    if (!ms.measure(n) || !hb.measure(n) || hb.total > 0xF0000000ull) return BBS_E_ARG;
    const size_t nm = raw ? (size_t)ms.total : 0;
    // (message t of the batch is entry msg_off[0] + t of msg_byte_off: item offsets need not start at zero)
    RaggedIn mb{raw ? (nm ? msg_byte_off + msg_off[0] : zero_off1()) : nullptr, msg_bytes, 1};   // nm == 0: msg_byte_off is never indexed
    if (raw && (!mb.measure(nm) || mb.total > 0xF0000000ull)) return BBS_E_ARG;
    if (int rc0 = stage_image(job.get(), n, nullptr, 0, {&ms, &hb}, raw ? &mb : nullptr, nm)) return rc0;
    const uint8_t* dimg = job->d_raw.template as<uint8_t>();
    int rc = BBS_OK;
    const size_t Lw = (size_t)std::max(L, 1), nn = std::max<size_t>(n, 1);
    SgArgs<C>& a = job->a;
    a.n = n; a.L = L; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    std::memcpy(a.sk, ctx->sk, sizeof(a.sk));
    uint32_t* smsgs = job->template scratch<uint32_t>(Lw * 8 * nn, rc);
    uint32_t* offs = job->template scratch<uint32_t>(2 * nn, rc);
    if (rc) return rc;
    a.msgs = smsgs;
    a.hdr_off = offs; a.hdr_len = offs + nn; a.hdr_bytes = dimg + hb.at_data;
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)NFIX * 3 * N * n, rc);
    a.out_a = job->template scratch<uint32_t>((size_t)2 * C::FpP::NC * n, rc);
    a.out_e = job->template scratch<uint32_t>((size_t)8 * n, rc);
    a.out_rec = job->template scratch<uint32_t>((size_t)(2 * C::FpP::NC + 8) * nn, rc);
    a.oct_form = 0;
    if (rc) return rc;
    if ((rc = job->finish_setup_device())) return rc;
    a.status = job->d_status.template as<int8_t>();
    VfIngestArgs<C>& ia = job->ingest;
    ia.n = n; ia.L = L; ia.dst_too_long = ctx->dst_too_long ? 1 : 0; ia.has_sig = 0;
    ia.rec = nullptr; ia.oct = nullptr; ia.pcode = nullptr; ia.msg_dst_too_long = 0;
    ia.m_off = reinterpret_cast<const uint64_t*>(dimg + ms.at_off); ia.hdr_off64 = reinterpret_cast<const uint64_t*>(dimg + hb.at_off);
    ia.m = reinterpret_cast<const uint32_t*>(dimg + ms.at_data);
    if (raw) {
        ia.m = hash_raw_messages<C>(job.get(), ctx, mb, nm, job->mh, ia.msg_dst_too_long, rc);
        if (rc) return rc;
    }
    ia.sig_a = nullptr; ia.sig_e = nullptr; ia.msgs = smsgs; ia.hdr_off = offs; ia.hdr_len = offs + nn;
    ia.status0 = job->d_status0.template as<int8_t>();
    if (rt::launch<VfIngest<C>>(job->stream(), ia, n)) return BBS_E_HIP;
    SgJob<C>* j = job.get();
    j->stages.push_back({"sg_scalars", [j]() { return rt::launch<SgScalars<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"sg_msm_parts", [j]() { return rt::launch<SgMsmPart<C>>(j->stream(), j->a, j->n * NFIX); }});
    j->stages.push_back({"sg_combine", [j]() { return rt::launch<SgCombine<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"sg_emit", [j]() { return rt::launch<SgEmit<C>>(j->stream(), j->a, j->n); }});
    *out = job.release();
    return BBS_OK;
}

