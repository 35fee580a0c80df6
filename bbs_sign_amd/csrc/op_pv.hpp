// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

// ---- proof_verify ----------------------------------------------------------------------------
template <class C>
struct PvJob : JobBase<C> {
    using JobBase<C>::JobBase;
    PvArgs<C> a{};
    PairArgs<C> pa{};
    PvFinishArgs fin{};
    BvState<C> bv{};                  // batch verification only
};

template <class C>
int pv_upload(Ctx<C>* ctx, size_t n, const uint8_t* proofs_fixed, const uint8_t* commitments,
                     const uint64_t* commit_off, const uint8_t* dmsgs, const uint64_t* dmsg_off,
                     const uint64_t* didx, const uint64_t* didx_off, const uint8_t* headers,
                     const uint64_t* hdr_off, const uint8_t* ph, const uint64_t* ph_off, bbs_job** out) {
    constexpr int N = C::FpP::N;        // internal limbs
    constexpr int NC = C::FpP::NC;      // canonical 32-bit words
    constexpr int FPB = 4 * NC;
    using R = typename C::FrP;
    if (!ctx->gens_set || !ctx->pk_set) return BBS_E_STATE;
    if (!out || (n && (!proofs_fixed || !commit_off || !dmsg_off || !didx_off))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    const size_t rec = 6 * FPB + 128;
    auto job = std::unique_ptr<PvJob<C>>(new PvJob<C>(ctx));
    job->n = n;
    job->status0.assign(n, 1);
    size_t rmax = 1;
    for (size_t i = 0; i < n; i++) rmax = std::max<size_t>(rmax, (size_t)(didx_off[i + 1] - didx_off[i]));
    if (rmax > 0xFFFFFF) return BBS_E_ARG;
    Soa pts, sc, slots, dmask, didx_s, rcount;
    pts.init(3 * 2 * NC, n); sc.init(4 * 8, n); slots.init((size_t)std::max(L, 1) * 8, n);
    dmask.init((size_t)(std::max(L, 1) + 31) / 32, n); didx_s.init(rmax, n); rcount.init(1, n);
    std::vector<uint8_t> seen;
    for (size_t i = 0; i < n; i++) {
        int8_t& st = job->status0[i];
        const size_t u = (size_t)(commit_off[i + 1] - commit_off[i]);
        const size_t r = (size_t)(didx_off[i + 1] - didx_off[i]);
        const size_t rm = (size_t)(dmsg_off[i + 1] - dmsg_off[i]);
        const size_t l = u + r;
        const uint64_t* idx = didx + didx_off[i];
        // proof_verify.rs:139-150 (order of checks)
        bool bad = false;
        for (size_t k = 0; k < r; k++) if (idx[k] >= l) bad = true;
        if (bad) { st = BBS_ST_INVALID_DISCLOSED_INDEX; continue; }
        if (rm != r) { st = BBS_ST_INVALID_INDICES_AND_MESSAGES_LENGTH; continue; }
        if ((size_t)L != l) { st = BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH; continue; }
        if (ctx->dst_too_long) { st = BBS_ST_PANIC_DST_TOO_LONG; continue; }
        // duplicates make the undisclosed set larger than `commitments`: the reference indexes
        // proof.commitments[i] out of bounds (proof_verify.rs:177-179) and panics
        seen.assign(l, 0);
        size_t distinct = 0;
        for (size_t k = 0; k < r; k++) if (!seen[idx[k]]) { seen[idx[k]] = 1; distinct++; }
        if (distinct != r) { st = BBS_ST_PANIC_INDEX_OUT_OF_BOUNDS; continue; }
        const uint8_t* pf = proofs_fixed + i * rec;
        bool ok = true;
        for (int p = 0; p < 3; p++) ok &= pack_g1<C>(pts, (size_t)p * 2 * NC, i, pf + (size_t)p * 2 * FPB);
        for (int k = 0; k < 4; k++) ok &= pack_fe<R>(sc, (size_t)k * 8, i, pf + 6 * FPB + 32 * k);
        // slots: disclosed messages at their index, commitments at the sorted undisclosed indexes
        for (size_t k = 0; k < r; k++) {
            const size_t j = (size_t)idx[k];
            ok &= pack_fe<R>(slots, j * 8, i, dmsgs + (dmsg_off[i] + k) * 32);
            dmask.at(j >> 5, i) |= 1u << (j & 31);
            didx_s.at(k, i) = (uint32_t)j;
        }
        size_t cu = 0;
        for (size_t j = 0; j < l; j++) {
            if (seen[j]) continue;
            ok &= pack_fe<R>(slots, j * 8, i, commitments + (commit_off[i] + cu) * 32);
            cu++;
        }
        rcount.at(0, i) = (uint32_t)r;
        if (!ok) st = BBS_ST_NONCANONICAL;
    }
    BytePool hp, pp;
    if (!hp.build(n, headers, hdr_off) || !pp.build(n, ph, ph_off)) return BBS_E_ARG;
    int rc = BBS_OK;
    PvArgs<C>& a = job->a;
    a.n = n; a.L = L; a.Rmax = (int)rmax; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    a.glv = (C::K::HAS_GLV && (C::K::GLV_ALWAYS || ctx->points_in_subgroup)) ? 1 : 0;
    a.pts = job->up(pts.soa(), rc); a.sc = job->up(sc.soa(), rc); a.slots = job->up(slots.soa(), rc);
    a.dmask = job->up(dmask.soa(), rc); a.didx = job->up(didx_s.soa(), rc); a.rcount = job->up(rcount.soa(), rc);
    a.hdr_off = job->up(hp.off, rc); a.hdr_len = job->up(hp.len, rc); a.hdr_bytes = job->up(hp.bytes, rc);
    a.ph_off = job->up(pp.off, rc); a.ph_len = job->up(pp.len, rc); a.ph_bytes = job->up(pp.bytes, rc);
    a.dom = job->template scratch<uint32_t>(8 * n, rc);
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)PV_NPARTS * 3 * N * n, rc);
    a.aff = job->template scratch<uint32_t>((size_t)5 * 2 * N * n, rc);
    a.fmiller = job->template scratch<uint32_t>((size_t)2 * 12 * N * n, rc);
    a.vtab = job->template scratch<uint32_t>((size_t)4 * G1_TAB * 2 * N * std::max<size_t>(n, 1), rc);   // T1's three tables + the one of D * r3^
    if (rc) return rc;
    if ((rc = job->finish_setup())) return rc;
    a.status = job->d_status.template as<int8_t>();
    int8_t* pair_ok = job->template scratch<int8_t>(n ? n : 1, rc);
    if (rc) return rc;
    PairArgs<C>& pa = job->pa;
    pa.n = n; pa.cc = a.cc; pa.pa = a.pts; pa.pb = a.pts + (size_t)2 * NC * n; pa.negate_b = 1;
    pa.canonical = 1; pa.gate_arr = job->d_status0.template as<int8_t>(); pa.gate = 1; pa.out = pair_ok;
    pa.fmiller = a.fmiller;
    job->fin.n = n; job->fin.status = a.status; job->fin.pair_ok = pair_ok;
    PvJob<C>* j = job.get();
    j->stages.push_back({"pv_scalars", [j]() { return rt::launch<PvScalars<C>>(j->stream(), j->a, j->n); }});
    if (!ctx->batch_verify) {
        // every item its own pairing product, on the job's second stream concurrently with the MSM / challenge
        // stages: it needs only the proof's own points (canonical, converted in the kernel) and the
        // host-validated flag
        add_pairing_stages<C>(j, &j->pa, 1, "pair_miller", "pair_final_exp", "pairing_6lane");
        j->stages.push_back({"pv_msm_parts", [j]() { return rt::launch<PvMsmPart<C>>(j->stream(), j->a, j->n * PV_NPARTS); }});
        j->stages.push_back({"pv_challenge", [j]() { return rt::launch<PvChallenge<C>>(j->stream(), j->a, j->n); }});
        j->stages.push_back({"pv_finish", [j]() { return rt::launch<PvFinish>(j->stream(), j->fin, j->n); }, 0, 1});
    } else {
        // batch verification (pippenger.hpp): one combined pairing check over the items whose challenge matched;
        // if it fails, the per-item kernel decides (its lanes return at once when the combined check passed)
        pa.gate_arr = a.status; pa.gate = 2;                                     // fallback: items still pending
        j->stages.push_back({"pv_msm_parts", [j]() { return rt::launch<PvMsmPart<C>>(j->stream(), j->a, j->n * PV_NPARTS); }});
        j->stages.push_back({"pv_challenge", [j]() { return rt::launch<PvChallenge<C>>(j->stream(), j->a, j->n); }});
        // a_bar, b_bar in Montgomery form, stored by PvMsmPart
        if ((rc = add_batch_verification<C>(j, &j->bv, ctx, n, a.cc, a.status, a.aff, a.aff + (size_t)2 * N * n, 1, &j->pa))) return rc;
        j->stages.push_back({"pv_finish", [j]() { return rt::launch<PvFinish>(j->stream(), j->fin, j->n); }});
    }
    *out = job.release();
    return BBS_OK;
}

