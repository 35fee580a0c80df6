// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

// ---- proof_gen ---------------------------------------------------------------------------------
template <class C>
struct PgJob : JobBase<C> {
    using JobBase<C>::JobBase;
    PgArgs<C> a{};
    std::vector<std::vector<uint32_t>> undisclosed;     // per item, sorted
    int fetch_proofs(uint8_t* pf_out, uint8_t* commit_out, uint64_t* commit_off) override {
        constexpr int N = C::FpP::NC;       // canonical words
        constexpr int FPB = 4 * N;
        if (this->use() || rt::sync(this->stream())) return BBS_E_HIP;
        const size_t n = this->n;
        const int L = a.L;
        std::vector<uint32_t> P((size_t)3 * 2 * N * n), S((size_t)4 * 8 * n), M((size_t)std::max(L, 1) * 8 * n);
        if (this->down(P, a.out_pts) || this->down(S, a.out_sc) || this->down(M, a.out_mhat)) return BBS_E_HIP;
        std::vector<int8_t> st(n);
        if (rt::d2h(st.data(), this->d_status.p, n, this->stream())) return BBS_E_HIP;
        const size_t rec = 6 * FPB + 128;
        uint64_t off = 0;
        for (size_t i = 0; i < n; i++) {
            if (commit_off) commit_off[i] = off;
            if (st[i] != 1) { if (pf_out) std::memset(pf_out + i * rec, 0, rec); continue; }
            if (pf_out) {
                for (int p = 0; p < 3; p++) unpack_words_le(P, n, (size_t)p * 2 * N, i, 2 * N, pf_out + i * rec + (size_t)p * 2 * FPB);
                for (int k = 0; k < 4; k++) unpack_words_le(S, n, (size_t)k * 8, i, 8, pf_out + i * rec + 6 * FPB + 32 * k);
            }
            for (uint32_t j : undisclosed[i]) {
                if (commit_out) unpack_words_le(M, n, (size_t)j * 8, i, 8, commit_out + off * 32);
                off++;
            }
        }
        if (commit_off) commit_off[n] = off;
        return BBS_OK;
    }
};

template <class C>
int pg_upload(Ctx<C>* ctx, size_t n, const uint8_t* sigs, const uint8_t* msgs, const uint64_t* msg_off,
                     const uint64_t* didx, const uint64_t* didx_off, const uint8_t* rnd, const uint64_t* rnd_off,
                     const uint8_t* headers, const uint64_t* hdr_off, const uint8_t* ph, const uint64_t* ph_off,
                     bbs_job** out) {
    constexpr int N = C::FpP::N;
    constexpr int NC = C::FpP::NC;
    constexpr int FPB = 4 * NC;
    using R = typename C::FrP;
    if (!ctx->gens_set || !ctx->pk_set) return BBS_E_STATE;
    if (!out || (n && (!sigs || !msg_off || !didx_off || !rnd_off || !rnd))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    const size_t rec = 2 * FPB + 32;
    auto job = std::unique_ptr<PgJob<C>>(new PgJob<C>(ctx));
    job->n = n;
    job->status0.assign(n, ST_PENDING);
    job->undisclosed.resize(n);
    size_t rmax = 1;
    for (size_t i = 0; i < n; i++) rmax = std::max<size_t>(rmax, (size_t)(didx_off[i + 1] - didx_off[i]));
    Soa sa, se, sm, dmask, didx_s, rcount, rnd5, mt;
    sa.init(2 * NC, n); se.init(8, n); sm.init((size_t)std::max(L, 1) * 8, n);
    dmask.init((size_t)(std::max(L, 1) + 31) / 32, n); didx_s.init(rmax, n); rcount.init(1, n);
    rnd5.init(5 * 8, n); mt.init((size_t)std::max(L, 1) * 8, n);
    std::vector<uint8_t> seen;
    for (size_t i = 0; i < n; i++) {
        int8_t& st = job->status0[i];
        const size_t l = (size_t)(msg_off[i + 1] - msg_off[i]);
        const size_t r = (size_t)(didx_off[i + 1] - didx_off[i]);
        const size_t nr = (size_t)(rnd_off[i + 1] - rnd_off[i]);
        const uint64_t* idx = didx + didx_off[i];
        // proof_gen.rs:133-143
        if (r > l) { st = BBS_ST_INVALID_DISCLOSED_INDICES_LENGTH; continue; }
        bool bad = false;
        for (size_t k = 0; k < r; k++) if (idx[k] >= l) bad = true;
        if (bad) { st = BBS_ST_INVALID_DISCLOSED_INDEX; continue; }
        if (nr != 5 + l - r) return BBS_E_ARG;                       // contract of this ABI (:145-149)
        // proof_init, proof_gen.rs:229-239
        if (l != (size_t)L) { st = BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH; continue; }
        seen.assign(l, 0);
        size_t distinct = 0;
        for (size_t k = 0; k < r; k++) if (!seen[idx[k]]) { seen[idx[k]] = 1; distinct++; }
        if (distinct != r) { st = BBS_ST_INVALID_RANDOM_SCALARS_AND_UNDISCLOSED_INDICES_LENGTH; continue; }
        if (ctx->dst_too_long) { st = BBS_ST_PANIC_DST_TOO_LONG; continue; }
        bool ok = pack_g1<C>(sa, 0, i, sigs + i * rec);
        ok &= pack_fe<R>(se, 0, i, sigs + i * rec + 2 * FPB);
        for (size_t j = 0; j < l; j++) ok &= pack_fe<R>(sm, j * 8, i, msgs + (msg_off[i] + j) * 32);
        const uint8_t* rs = rnd + rnd_off[i] * 32;
        for (int k = 0; k < 5; k++) ok &= pack_fe<R>(rnd5, (size_t)k * 8, i, rs + 32 * k);
        size_t ku = 0, kd = 0;
        for (size_t j = 0; j < l; j++) {
            if (seen[j]) {
                dmask.at(j >> 5, i) |= 1u << (j & 31);
                didx_s.at(kd++, i) = (uint32_t)j;
            } else {
                ok &= pack_fe<R>(mt, j * 8, i, rs + 32 * (5 + ku));
                ku++;
                job->undisclosed[i].push_back((uint32_t)j);
            }
        }
        rcount.at(0, i) = (uint32_t)kd;
        if (!ok) st = BBS_ST_NONCANONICAL;
    }
    BytePool hp, pp;
    if (!hp.build(n, headers, hdr_off) || !pp.build(n, ph, ph_off)) return BBS_E_ARG;
    int rc = BBS_OK;
    PgArgs<C>& a = job->a;
    a.n = n; a.L = L; a.Rmax = (int)rmax; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    a.glv = (C::K::HAS_GLV && (C::K::GLV_ALWAYS || ctx->points_in_subgroup)) ? 1 : 0;
    a.sig_a = job->up(sa.soa(), rc); a.sig_e = job->up(se.soa(), rc); a.msgs = job->up(sm.soa(), rc);
    a.dmask = job->up(dmask.soa(), rc); a.didx = job->up(didx_s.soa(), rc); a.rcount = job->up(rcount.soa(), rc);
    a.rnd5 = job->up(rnd5.soa(), rc); a.mtilde = job->up(mt.soa(), rc);
    a.hdr_off = job->up(hp.off, rc); a.hdr_len = job->up(hp.len, rc); a.hdr_bytes = job->up(hp.bytes, rc);
    a.ph_off = job->up(pp.off, rc); a.ph_len = job->up(pp.len, rc); a.ph_bytes = job->up(pp.bytes, rc);
    a.dom = job->template scratch<uint32_t>(8 * n, rc);
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.fscal2 = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.vscal = job->template scratch<uint32_t>((size_t)PG_NVAR * 8 * n, rc);
    a.bpart = job->template scratch<uint32_t>((size_t)NFIX * 3 * N * n, rc);
    a.baff = job->template scratch<uint32_t>((size_t)2 * 2 * N * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)PG_NPARTS * 3 * N * n, rc);
    a.out_pts = job->template scratch<uint32_t>((size_t)3 * 2 * NC * n, rc);
    a.out_sc = job->template scratch<uint32_t>((size_t)4 * 8 * n, rc);
    a.out_mhat = job->template scratch<uint32_t>((size_t)std::max(L, 1) * 8 * n, rc);
    if (rc) return rc;
    if ((rc = job->finish_setup())) return rc;
    a.status = job->d_status.template as<int8_t>();
    PgJob<C>* j = job.get();
    j->stages.push_back({"pg_scalars", [j]() { return rt::launch<PgScalars<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"pg_b_parts", [j]() { return rt::launch<PgBPart<C>>(j->stream(), j->a, j->n * NFIX); }});
    j->stages.push_back({"pg_b_combine", [j]() { return rt::launch<PgBCombine<C>>(j->stream(), j->a, j->n); }});
    j->stages.push_back({"pg_msm_parts", [j]() { return rt::launch<PgMsmPart<C>>(j->stream(), j->a, j->n * PG_NPARTS); }});
    j->stages.push_back({"pg_finalize", [j]() { return rt::launch<PgFinalize<C>>(j->stream(), j->a, j->n); }});
    *out = job.release();
    return BBS_OK;
}

