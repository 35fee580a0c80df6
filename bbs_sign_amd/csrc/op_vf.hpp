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
    VfIngestArgs<C> ingest{};
    VfOctArgs<C> oct{};               // wire form only
    MsgHashArgs mh{};                 // raw-message form only
    BvState<C> bv{};                  // batch verification only
};

template <class C>
int vf_upload(Ctx<C>* ctx, size_t n, const uint8_t* sigs, const uint8_t* msgs, const uint64_t* msg_off,
                     const uint8_t* headers, const uint64_t* hdr_off, bbs_job** out, const uint8_t* octets,
                     const uint8_t* msg_bytes, const uint64_t* msg_byte_off) {
    // msg_byte_off != nullptr: the messages arrive as RAW BYTES (message t of the batch = msg_bytes[msg_byte_off[t] ..
    // msg_byte_off[t + 1]), msg_off counts messages per item, msgs is ignored) and are hashed to scalars on the device
    // octets != nullptr: the wire form -- n strings compress(A) || e instead of the records `sigs`
    constexpr int N = C::FpP::N;
    constexpr int NC = C::FpP::NC;
    constexpr int FPB = 4 * NC;
    if (!ctx->gens_set || !ctx->pk_set) return BBS_E_STATE;
    if (!out || (n && ((!sigs && !octets) || !msg_off))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    const bool wire = octets != nullptr;
    const size_t rec = wire ? (size_t)FPB + 32 : (size_t)2 * FPB + 32;
    if (wire) sigs = octets;
    auto job = std::unique_ptr<VfJob<C>>(new VfJob<C>(ctx));
    job->n = n;
    // the batch as one staging image, one asynchronous copy; checks, range checks and the SoA transposition on the
    // device (stage VfIngest), as for proof_verify
    const bool raw = msg_byte_off != nullptr;
    RaggedIn ms{msg_off, msgs, 32}, hb{hdr_off, headers, 1};
    ms.offsets_only = raw;
    if (!ms.measure(n) || !hb.measure(n) || hb.total > 0xF0000000ull) return BBS_E_ARG;
    const size_t nm = raw ? (size_t)ms.total : 0;
    // (message t of the batch is entry msg_off[0] + t of msg_byte_off: item offsets need not start at zero)
    RaggedIn mb{raw ? (nm ? msg_byte_off + msg_off[0] : zero_off1()) : nullptr, msg_bytes, 1};   // nm == 0: msg_byte_off is never indexed
    if (raw && (!mb.measure(nm) || mb.total > 0xF0000000ull)) return BBS_E_ARG;
    if (int rc0 = stage_image(job.get(), n, sigs, rec, {&ms, &hb}, raw ? &mb : nullptr, nm)) return rc0;
    const uint8_t* dimg = job->d_raw.template as<uint8_t>();
    int rc = BBS_OK;
    const size_t Lw = (size_t)std::max(L, 1), nn = std::max<size_t>(n, 1);
    VfArgs<C>& a = job->a;
    a.n = n; a.L = L; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    // wire form: the decoder has checked that A is in G1, so the GLV split is sound without the caller's word
    a.glv = (C::K::HAS_GLV && (C::K::GLV_ALWAYS || ctx->points_in_subgroup || wire)) ? 1 : 0;
    uint32_t* sig_a = job->template scratch<uint32_t>((size_t)2 * NC * nn, rc);
    uint32_t* sig_e = job->template scratch<uint32_t>((size_t)8 * nn, rc);
    uint32_t* smsgs = job->template scratch<uint32_t>(Lw * 8 * nn, rc);
    uint32_t* offs = job->template scratch<uint32_t>(2 * nn, rc);
    if (rc) return rc;
    a.sig_a = sig_a; a.sig_e = sig_e; a.msgs = smsgs;
    a.hdr_off = offs; a.hdr_len = offs + nn; a.hdr_bytes = dimg + hb.at_data;
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)VF_NPARTS * 3 * N * n, rc);
    a.aff = job->template scratch<uint32_t>((size_t)2 * 2 * N * n, rc);
    a.fmiller = job->template scratch<uint32_t>((size_t)2 * 12 * N * n, rc);
    a.vtab = job->template scratch<uint32_t>((size_t)G1_TAB * 2 * N * nn, rc);
    if (rc) return rc;
    if ((rc = job->finish_setup_device())) return rc;
    a.status = job->d_status.template as<int8_t>();
    VfIngestArgs<C>& ia = job->ingest;
    ia.n = n; ia.L = L; ia.dst_too_long = ctx->dst_too_long ? 1 : 0; ia.has_sig = 1;
    ia.rec = wire ? nullptr : reinterpret_cast<const uint32_t*>(dimg);
    ia.oct = nullptr; ia.pcode = nullptr; ia.msg_dst_too_long = 0;
    if (wire) {
        int8_t* pcode = job->template scratch<int8_t>(nn, rc);
        if (rc) return rc;
        VfOctArgs<C>& oa = job->oct;
        oa.n = n; oa.oct = dimg; oa.sig_a = sig_a; oa.pcode = pcode;
        if (rt::launch<VfOctDecode<C>>(job->stream(), oa, n)) return BBS_E_HIP;
        ia.oct = dimg; ia.pcode = pcode;
    }
    ia.m_off = reinterpret_cast<const uint64_t*>(dimg + ms.at_off); ia.hdr_off64 = reinterpret_cast<const uint64_t*>(dimg + hb.at_off);
    ia.m = reinterpret_cast<const uint32_t*>(dimg + ms.at_data);
    if (raw) {
        ia.m = hash_raw_messages<C>(job.get(), ctx, mb, nm, job->mh, ia.msg_dst_too_long, rc);
        if (rc) return rc;
    }
    ia.sig_a = sig_a; ia.sig_e = sig_e; ia.msgs = smsgs; ia.hdr_off = offs; ia.hdr_len = offs + nn;
    ia.status0 = job->d_status0.template as<int8_t>();
    if (rt::launch<VfIngest<C>>(job->stream(), ia, n)) return BBS_E_HIP;
    PairArgs<C>& pa = job->pa;
    pa.n = n; pa.cc = a.cc; pa.pa = a.aff; pa.pb = a.aff + (size_t)2 * N * n; pa.negate_b = 0;
    pa.canonical = 0; pa.gate_arr = a.status; pa.gate = ST_PAIRING; pa.out = a.status; pa.fmiller = a.fmiller;
    VfJob<C>* j = job.get();
    // e * A needs only what the ingest stage left: it runs on the job's side stream beside the scalars and the fixed-base
    // chunks (critical path 3.0 instead of 0.13 + 0.9 + 3.0 ms).  In both forms: with 6 / 8 / 12 resident jobs in flight
    // 1.75 / 1.79 / 1.81 M verify/s against 1.64 / 1.71 / 1.81 with the job kept to one stream
    // (profiles/r05_e_verify_var_mul_side_stream.log).  Except batch verification's throughput form, whose jobs are kept alive
    // by the dozen and must own ONE hardware queue each (32 in flight: 4.2 M/s on one stream, 3.1 M/s on two).
    static const int side_forced = []() { const char* v = getenv("BBS_VF_SIDE"); return v ? atoi(v) : -1; }();      // A/B: 0 never, 1 always
    const int side = side_forced >= 0 ? (side_forced ? 1 : 0) : ((ctx->batch_verify && !job->latency_form) ? 0 : 1);
    j->stages.push_back({"vf_var_mul", [j, side]() { return rt::launch<VfVarMul<C>>(side ? j->stream_aux(1) : j->stream(), j->a, j->n); }, side, 0});
    j->stages.push_back({"vf_scalars", [j]() { return rt::launch<VfScalars<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"vf_fixed_chunks", [j]() { return rt::launch<VfFixedChunk<C>>(j->stream(), j->a, j->n * (size_t)NFIX); }});
    j->stages.push_back({"vf_combine", [j]() { return rt::launch<VfCombine<C>>(j->stream(), j->a, j->n); }, 0, 1});
    if (!ctx->batch_verify) {
        add_pairing_stages<C>(j, &j->pa, 0, "pair_miller", "pair_final_exp", "pairing_6lane");
    } else {
        // e(A, W) e(e A - B, BP2) == 1 for all pending items at once: A and e A - B are in a.aff (Montgomery)
        // (the second point is computed by vf_combine, so here the combination follows the MSM chain on the main stream)
        if ((rc = add_batch_combination<C>(j, &j->bv, ctx, n, a.cc, a.aff, a.aff + (size_t)2 * N * n, 0, a.status, ST_PAIRING, 0, 0))) return rc;
        add_batch_decision<C>(j, &j->bv, &j->pa, 0);
    }
    *out = job.release();
    return BBS_OK;
}

