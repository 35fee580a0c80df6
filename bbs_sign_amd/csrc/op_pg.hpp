// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

// ---- proof_gen ---------------------------------------------------------------------------------
template <class C>
struct PgJob : JobBase<C> {
    using JobBase<C>::JobBase;
    PgArgs<C> a{};
    PgIngestArgs<C> ingest{};
    VfOctArgs<C> oct{};               // signature octets in (wire form) only
    MsgHashArgs mh{};                 // raw-message form only
    // The proofs come off the device in the caller's layout (stage PgEmit): records, the m^ of the undisclosed messages
    // in ascending index order (L slots per item, the first ucount[i] used) and those counts.  Host side of a delivery:
    // one copy of the records, one contiguous copy per item of its commitments, the running offsets.
    size_t rec_bytes() const { return (size_t)(24 * C::FpP::NC + 128); }
    size_t mh_bytes() const { return (size_t)std::max(a.L, 1) * 32; }              // per item
    void unpack(const uint8_t* rec, const uint8_t* mh, const uint32_t* ucount, uint8_t* pf_out, uint8_t* commit_out,
                uint64_t* commit_off) const {
        const size_t n = this->n;
        if (pf_out && n) std::memcpy(pf_out, rec, n * rec_bytes());
        uint64_t off = 0;
        for (size_t i = 0; i < n; i++) {
            if (commit_off) commit_off[i] = off;
            const size_t u = ucount[i];
            if (commit_out && u) std::memcpy(commit_out + off * 32, mh + i * mh_bytes(), u * 32);
            off += u;
        }
        if (commit_off) commit_off[n] = off;
    }
    size_t out_bytes() const { return this->n * (rec_bytes() + mh_bytes() + 4); }
    // wire form (PgEmit, oct_form): the octet strings sit at a fixed stride at the start of the block, the counts where
    // they always are; ucount = 0xFFFFFFFF marks a failed item (no string)
    int set_octet_form() override { a.oct_form = 1; return BBS_OK; }
    size_t oct_stride() const { return (size_t)(12 * C::FpP::NC) + 32 * (size_t)(4 + std::max(a.L, 1)); }
    void unpack_octets(const uint8_t* blk, const uint32_t* ucount, uint8_t* oct_out, uint64_t* oct_off) const {
        uint64_t off = 0;
        for (size_t i = 0; i < this->n; i++) {
            if (oct_off) oct_off[i] = off;
            if (ucount[i] == 0xFFFFFFFFu) continue;
            const size_t len = (size_t)(12 * C::FpP::NC) + 32 * (size_t)(4 + ucount[i]);
            if (oct_out) std::memcpy(oct_out + off, blk + i * oct_stride(), len);
            off += len;
        }
        if (oct_off) oct_off[this->n] = off;
    }
    // one device block [records | m^ | counts] so that a fetch is one copy
    int fetch_proofs(uint8_t* pf_out, uint8_t* commit_out, uint64_t* commit_off) override {
        if (this->use() || rt::sync(this->stream())) return BBS_E_HIP;
        const size_t n = this->n;
        if (a.oct_form) return BBS_E_STATE;                     // the wire form is delivered by bbs_job_wait only
        if (!n) { if (commit_off) commit_off[0] = 0; return BBS_OK; }
        if (int rc = this->require_decided()) return rc;        // a job that never ran holds no records (fail closed)
        std::vector<uint8_t> h(out_bytes());
        if (rt::d2h(h.data(), a.out_rec, h.size(), this->stream())) return BBS_E_HIP;
        unpack(h.data(), h.data() + n * rec_bytes(), reinterpret_cast<const uint32_t*>(h.data() + n * (rec_bytes() + mh_bytes())),
               pf_out, commit_out, commit_off);
        return BBS_OK;
    }
    // submit form
    uint8_t* pf_to = nullptr;
    uint8_t* cm_to = nullptr;
    uint64_t* cmo_to = nullptr;
    HostBuf h_out;
    void set_result_targets(uint8_t* pf, uint8_t* cm, uint64_t* cmo) override { pf_to = pf; cm_to = cm; cmo_to = cmo; }
    int enqueue_result_fetch() override {
        if (!this->n) return BBS_OK;
        if (!h_out.p && h_out.alloc(out_bytes())) return BBS_E_NOMEM;
        return rt::d2h_async(h_out.p, a.out_rec, out_bytes(), this->stream()) ? BBS_E_HIP : BBS_OK;
    }
    int deliver() override {
        if (int rc = JobBase<C>::deliver()) return rc;
        if (!pf_to && !cm_to && !cmo_to) return BBS_OK;
        const size_t n = this->n;
        if (!n) { if (cmo_to) cmo_to[0] = 0; return BBS_OK; }
        if (!h_out.p) return BBS_E_STATE;
        const uint8_t* h = h_out.template as<uint8_t>();
        if (a.oct_form) {
            unpack_octets(h, reinterpret_cast<const uint32_t*>(h + n * (rec_bytes() + mh_bytes())), pf_to, cmo_to);
            return BBS_OK;
        }
        unpack(h, h + n * rec_bytes(), reinterpret_cast<const uint32_t*>(h + n * (rec_bytes() + mh_bytes())), pf_to, cm_to, cmo_to);
        return BBS_OK;
    }
};

template <class C>
int pg_upload(Ctx<C>* ctx, size_t n, const uint8_t* sigs, const uint8_t* msgs, const uint64_t* msg_off,
                     const uint64_t* didx, const uint64_t* didx_off, const uint8_t* rnd, const uint64_t* rnd_off,
                     const uint8_t* headers, const uint64_t* hdr_off, const uint8_t* ph, const uint64_t* ph_off,
                     bbs_job** out, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off) {
    // sig_octets != nullptr: the signatures arrive as octet strings compress(A) || e (decoded and subgroup-checked on the
    // device, `sigs` ignored); msg_byte_off != nullptr: the messages arrive as raw bytes (see vf_upload)
    constexpr int N = C::FpP::N;
    constexpr int NC = C::FpP::NC;
    constexpr int FPB = 4 * NC;
    if (!ctx->gens_set || !ctx->pk_set) return BBS_E_STATE;
    const bool wire = sig_octets != nullptr, raw = msg_byte_off != nullptr;
    if (!out || (n && ((!sigs && !wire) || !msg_off || !didx_off || !rnd_off || !rnd))) return BBS_E_ARG;
    if (wire) sigs = sig_octets;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    const size_t rec = wire ? (size_t)FPB + 32 : (size_t)2 * FPB + 32;
    auto job = std::unique_ptr<PgJob<C>>(new PgJob<C>(ctx));
    job->n = n;
    RaggedIn ms{msg_off, msgs, 32}, di{didx_off, reinterpret_cast<const uint8_t*>(didx), 8}, rs{rnd_off, rnd, 32},
             hb{hdr_off, headers, 1}, pb{ph_off, ph, 1};
    ms.offsets_only = raw;
    if (!ms.measure(n) || !di.measure(n) || !rs.measure(n) || !hb.measure(n) || !pb.measure(n)) return BBS_E_ARG;
    const size_t nm = raw ? (size_t)ms.total : 0;
    // (message t of the batch is entry msg_off[0] + t of msg_byte_off: item offsets need not start at zero)
    RaggedIn mb{raw ? (nm ? msg_byte_off + msg_off[0] : zero_off1()) : nullptr, msg_bytes, 1};   // nm == 0: msg_byte_off is never indexed
    if (raw && (!mb.measure(nm) || mb.total > 0xF0000000ull)) return BBS_E_ARG;
    if (hb.total > 0xF0000000ull || pb.total > 0xF0000000ull) return BBS_E_ARG;
    // Host side, one comparison per item (no field data is touched): the contract of this ABI on the number of random
    // scalars (proof_gen.rs:145-149; checked where the reference would have got that far).  Everything else -- the
    // reference's checks, range checks, deduplication, unpacking, SoA transposition, and the layout of the results
    // (stage PgEmit) -- happens on the device.
    for (size_t i = 0; i < n; i++) {
        const size_t l = (size_t)(msg_off[i + 1] - msg_off[i]);
        const size_t r = (size_t)(didx_off[i + 1] - didx_off[i]);
        const size_t nr = (size_t)(rnd_off[i + 1] - rnd_off[i]);
        const uint64_t* idx = didx + didx_off[i];
        if (r > l) continue;                                          // -> InvalidDisclosedIndicesLength on the device
        bool bad = false;
        for (size_t k = 0; k < r; k++) if (idx[k] >= l) bad = true;
        if (bad) continue;                                            // -> InvalidDisclosedIndex
        if (nr != 5 + l - r) return BBS_E_ARG;
    }
    if (int rc0 = stage_image(job.get(), n, sigs, rec, {&ms, &di, &rs, &hb, &pb}, raw ? &mb : nullptr, nm)) return rc0;
    const uint8_t* dimg = job->d_raw.template as<uint8_t>();
    auto d64 = [&](size_t at) { return reinterpret_cast<const uint64_t*>(dimg + at); };
    auto d32 = [&](size_t at) { return reinterpret_cast<const uint32_t*>(dimg + at); };
    int rc = BBS_OK;
    const size_t Lw = (size_t)std::max(L, 1), nn = std::max<size_t>(n, 1);
    PgArgs<C>& a = job->a;
    a.n = n; a.L = L; a.Rmax = (int)Lw; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    a.glv = (C::K::HAS_GLV && (C::K::GLV_ALWAYS || ctx->points_in_subgroup || wire)) ? 1 : 0;   // wire: the decoder checked A
    PgIngestArgs<C>& ia = job->ingest;
    ia.n = n; ia.L = L; ia.dst_too_long = ctx->dst_too_long ? 1 : 0;
    ia.rec = wire ? nullptr : d32(0);
    ia.oct = nullptr; ia.pcode = nullptr; ia.msg_dst_too_long = 0;
    ia.m_off = d64(ms.at_off); ia.di_off = d64(di.at_off); ia.rnd_off = d64(rs.at_off); ia.hdr_off64 = d64(hb.at_off); ia.ph_off64 = d64(pb.at_off);
    ia.m = d32(ms.at_data); ia.di = d64(di.at_data); ia.rnd = d32(rs.at_data);
    ia.sig_a = job->template scratch<uint32_t>((size_t)2 * NC * nn, rc);
    ia.sig_e = job->template scratch<uint32_t>((size_t)8 * nn, rc);
    ia.msgs = job->template scratch<uint32_t>(Lw * 8 * nn, rc);
    ia.dmask = job->template scratch<uint32_t>(((Lw + 31) / 32) * nn, rc);
    ia.didx = job->template scratch<uint32_t>(Lw * nn, rc);
    ia.rcount = job->template scratch<uint32_t>(nn, rc);
    ia.rnd5 = job->template scratch<uint32_t>((size_t)5 * 8 * nn, rc);
    ia.mtilde = job->template scratch<uint32_t>(Lw * 8 * nn, rc);
    uint32_t* offs = job->template scratch<uint32_t>(4 * nn, rc);
    if (rc) return rc;
    if (wire) {
        int8_t* pcode = job->template scratch<int8_t>(nn, rc);
        if (rc) return rc;
        VfOctArgs<C>& oa = job->oct;
        oa.n = n; oa.oct = dimg; oa.sig_a = ia.sig_a; oa.pcode = pcode;
        if (rt::launch<VfOctDecode<C>>(job->stream(), oa, n)) return BBS_E_HIP;
        ia.oct = dimg; ia.pcode = pcode;
    }
    if (raw) {
        ia.m = hash_raw_messages<C>(job.get(), ctx, mb, nm, job->mh, ia.msg_dst_too_long, rc);
        if (rc) return rc;
    }
    ia.hdr_off = offs; ia.hdr_len = offs + nn; ia.ph_off = offs + 2 * nn; ia.ph_len = offs + 3 * nn;
    a.sig_a = ia.sig_a; a.sig_e = ia.sig_e; a.msgs = ia.msgs; a.dmask = ia.dmask; a.didx = ia.didx; a.rcount = ia.rcount;
    a.rnd5 = ia.rnd5; a.mtilde = ia.mtilde;
    a.hdr_off = ia.hdr_off; a.hdr_len = ia.hdr_len; a.hdr_bytes = dimg + hb.at_data;
    a.ph_off = ia.ph_off; a.ph_len = ia.ph_len; a.ph_bytes = dimg + pb.at_data;
    a.dom = job->template scratch<uint32_t>(8 * n, rc);
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.fscal2 = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.vscal = job->template scratch<uint32_t>((size_t)PG_NVAR * 8 * n, rc);
    a.bpart = job->template scratch<uint32_t>((size_t)NFIX * 3 * N * n, rc);
    a.baff = job->template scratch<uint32_t>((size_t)2 * 2 * N * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)PG_NPARTS * 3 * N * n, rc);
    // the variable-base parts: one lane per multiplication for a job that is alone (shortest longest lane), Bbar and T1 as
    // joint chains otherwise (14 % fewer instructions per proof)
    a.nvar = job->latency_form ? PG_NVAR : PG_NVAR_JOINT;
    a.vtab = job->template scratch<uint32_t>((size_t)PG_NVAR * G1_TAB * 2 * N * std::max<size_t>(n, 1), rc);
    a.ctab = nullptr; a.comb_ok = nullptr;
    static const bool use_comb = [] { const char* v = getenv("BBS_PG_COMB"); return !v || atoi(v) != 0; }();     // A/B: BBS_PG_COMB=0
    // (where the GLV split is on -- BN254 always, BLS12-381 for vouched or decoded points -- the comb runs over the 64-bit halves
    // of the split scalars: 64 doublings of preparation per point instead of 192; the plain comb there measured -12 % on BN254
    // against the 126-doubling joint chains)
    if (a.nvar == PG_NVAR_JOINT && use_comb) {
        a.ctab = job->template scratch<uint32_t>((size_t)2 * comb_table_words(N) * std::max<size_t>(n, 1), rc);
        a.comb_ok = job->template scratch<int8_t>(2 * std::max<size_t>(n, 1), rc);
    }
    a.out_pts = job->template scratch<uint32_t>((size_t)3 * 2 * NC * n, rc);
    a.out_sc = job->template scratch<uint32_t>((size_t)4 * 8 * n, rc);
    a.out_mhat = job->template scratch<uint32_t>((size_t)std::max(L, 1) * 8 * n, rc);
    {   // [records | m^ | counts] in one block (PgJob::out_bytes)
        const size_t wrec = (size_t)(6 * NC + 32) * nn, wmh = Lw * 8 * nn;
        a.out_rec = job->template scratch<uint32_t>(wrec + wmh + nn, rc);
        if (rc) return rc;
        a.out_mh = a.out_rec + wrec;
        a.ucount = a.out_mh + wmh;
        a.oct_form = 0;
    }
    if (rc) return rc;
    if ((rc = job->finish_setup_device())) return rc;
    a.status = job->d_status.template as<int8_t>();
    ia.status0 = job->d_status0.template as<int8_t>();
    if (rt::launch<PgIngest<C>>(job->stream(), ia, n)) return BBS_E_HIP;
    PgJob<C>* j = job.get();
    j->stages.push_back({"pg_scalars", [j]() { return rt::launch<PgScalars<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"pg_b_parts", [j]() { return rt::launch<PgBPart<C>>(j->stream(), j->a, j->n * (size_t)(2 * NFIX)); }});
    j->stages.push_back({"pg_b_combine", [j]() { return rt::launch<PgBCombine<C>>(j->stream(), j->a, j->n); }});
    if (j->a.ctab) j->stages.push_back({"pg_tables", [j]() { return rt::launch<PgTables<C>>(j->stream(), j->a, j->n * 2); }});
    j->stages.push_back({"pg_var_parts", [j]() { return rt::launch<PgVarPart<C>>(j->stream(), j->a, j->n * (size_t)j->a.nvar); }});
    j->stages.push_back({"pg_finalize", [j]() { return rt::launch<PgFinalize<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"pg_emit", [j]() { return rt::launch<PgEmit<C>>(j->stream(), j->a, j->n); }});
    *out = job.release();
    return BBS_OK;
}

