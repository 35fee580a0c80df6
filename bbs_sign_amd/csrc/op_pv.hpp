// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

// ---- proof_verify ----------------------------------------------------------------------------
// where a job's doubling chains run (see pv_upload): 3 / 2 / 1; BBS_PV_MSM_LAYOUT overrides (A/B), read once.
// one_queue: the job is meant to own ONE hardware queue (batch verification's throughput form keeps many jobs alive).
// latency_form: T1's three terms are single multiplications (PvVarMul parts 0 .. 2): all four multiplications go to the side
// stream as one launch, so that the main stream's scalars -> fixed-base chunks run BESIDE them -- on the main stream in front
// of the scalars they made the MSM chain (2.85 + 0.13 + 0.9 + 0.6 ms) longer than the pairing (4.2 ms) it runs beside:
// one batch at a time 4.7 against 4.35 ms (profiles/r05_a_ab_split_msm_layouts.log, new3 vs new2).
inline int pv_msm_layout(bool one_queue, bool latency_form) {
    static const int forced = []() { const char* v = getenv("BBS_PV_MSM_LAYOUT"); const int k = v ? atoi(v) : 0; return (k >= 1 && k <= 3) ? k : 0; }();
    if (forced) return forced;
    return one_queue ? 1 : (latency_form ? 2 : 3);
}
template <class C>
struct PvJob : JobBase<C> {
    using JobBase<C>::JobBase;
    PvArgs<C> a{};
    PairArgs<C> pa{};
    PvFinishArgs fin{};
    PvIngestArgs<C> ingest{};
    PvOctArgs<C> oct{};               // wire form only
    MsgHashArgs mh{};                 // wire form with raw messages only
    BvState<C> bv{};                  // batch verification only
};

// octets / oct_off != nullptr: the wire form (bbs_proof_verify_octets_*): proofs_fixed / commitments / commit_off are
// ignored, the proofs come as octet strings and are decoded on the device (codec_dev.hpp PvOctDecode / PvOctIngest)
template <class C>
int pv_upload(Ctx<C>* ctx, size_t n, const uint8_t* proofs_fixed, const uint8_t* commitments,
                     const uint64_t* commit_off, const uint8_t* dmsgs, const uint64_t* dmsg_off,
                     const uint64_t* didx, const uint64_t* didx_off, const uint8_t* headers,
                     const uint64_t* hdr_off, const uint8_t* ph, const uint64_t* ph_off, bbs_job** out,
                     const uint8_t* octets, const uint64_t* oct_off, const uint8_t* msg_bytes, const uint64_t* msg_byte_off) {
    // msg_byte_off != nullptr (wire form only): the disclosed messages arrive as RAW BYTES -- message t of the batch is
    // msg_bytes[msg_byte_off[t] .. msg_byte_off[t + 1]), dmsg_off counts messages per item as before, dmsgs is ignored --
    // and are mapped to scalars on the device (msg_to_scalars, interface_utilities.rs:76-88)
    const bool wire = oct_off != nullptr;
    const bool raw = msg_byte_off != nullptr;
    if (raw && !wire) return BBS_E_ARG;
    constexpr int N = C::FpP::N;        // internal limbs
    constexpr int NC = C::FpP::NC;      // canonical 32-bit words
    constexpr int FPB = 4 * NC;
    if (!ctx->gens_set || !ctx->pk_set) return BBS_E_STATE;
    if (!out || (n && (!dmsg_off || !didx_off))) return BBS_E_ARG;
    if (n && !wire && (!proofs_fixed || !commit_off)) return BBS_E_ARG;
    if (n && wire && !octets) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const int L = ctx->L;
    const size_t rec = 6 * FPB + 128;
    auto job = std::unique_ptr<PvJob<C>>(new PvJob<C>(ctx));
    job->n = n;
    // ---- the batch as one staging image in page-locked memory, one asynchronous copy; everything else (the
    // reference's checks, range checks, unpacking, the SoA transposition) happens on the device: stage PvIngest
    // first ragged section: the commitments (core form) or the proof octet strings (wire form)
    RaggedIn cm{wire ? oct_off : commit_off, wire ? octets : commitments, wire ? (size_t)1 : (size_t)32}, dm{dmsg_off, dmsgs, 32},
             di{didx_off, reinterpret_cast<const uint8_t*>(didx), 8}, hb{hdr_off, headers, 1}, pb{ph_off, ph, 1};
    dm.offsets_only = raw;
    if (!cm.measure(n) || !dm.measure(n) || !di.measure(n) || !hb.measure(n) || !pb.measure(n)) return BBS_E_ARG;
    if (hb.total > 0xF0000000ull || pb.total > 0xF0000000ull) return BBS_E_ARG;
    const size_t nm = raw ? (size_t)dm.total : 0;                   // disclosed messages of the whole batch
    // (message t of the batch is entry dmsg_off[0] + t of msg_byte_off: item offsets need not start at zero)
    RaggedIn mb{raw ? (nm ? msg_byte_off + dmsg_off[0] : zero_off1()) : nullptr, msg_bytes, 1};   // nm == 0: msg_byte_off is never indexed
    if (raw && (!mb.measure(nm) || mb.total > 0xF0000000ull)) return BBS_E_ARG;
    // (the message section is ragged over MESSAGES, not items: stage_image places and fills it with nm as its count)
    if (int rc0 = stage_image(job.get(), n, wire ? nullptr : proofs_fixed, wire ? 0 : rec, {&cm, &dm, &di, &hb, &pb}, raw ? &mb : nullptr, nm)) return rc0;

    const uint8_t* dimg = job->d_raw.template as<uint8_t>();
    auto d64 = [&](size_t at) { return reinterpret_cast<const uint64_t*>(dimg + at); };
    auto d32 = [&](size_t at) { return reinterpret_cast<const uint32_t*>(dimg + at); };
    int rc = BBS_OK;
    const size_t Lw = (size_t)std::max(L, 1), nn = std::max<size_t>(n, 1);
    PvArgs<C>& a = job->a;
    a.n = n; a.L = L; a.Rmax = (int)Lw; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    // wire form: every point has passed the subgroup check of the decoder, the GLV split needs no vouching
    a.glv = (C::K::HAS_GLV && (C::K::GLV_ALWAYS || ctx->points_in_subgroup || wire)) ? 1 : 0;
    a.nvar = job->latency_form ? PV_NVAR_SPLIT : PV_NVAR;
    uint32_t* pts = job->template scratch<uint32_t>((size_t)3 * 2 * NC * nn, rc);
    uint32_t* sc = job->template scratch<uint32_t>((size_t)4 * 8 * nn, rc);
    uint32_t* slots = job->template scratch<uint32_t>(Lw * 8 * nn, rc);
    uint32_t* dmask = job->template scratch<uint32_t>(((Lw + 31) / 32) * nn, rc);
    uint32_t* didx_s = job->template scratch<uint32_t>(Lw * nn, rc);
    uint32_t* rcount = job->template scratch<uint32_t>(nn, rc);
    uint32_t* offs = job->template scratch<uint32_t>(4 * nn, rc);
    if (rc) return rc;
    a.pts = pts; a.sc = sc; a.slots = slots; a.dmask = dmask; a.didx = didx_s; a.rcount = rcount;
    a.hdr_off = offs; a.hdr_len = offs + nn; a.ph_off = offs + 2 * nn; a.ph_len = offs + 3 * nn;
    a.hdr_bytes = dimg + hb.at_data; a.ph_bytes = dimg + pb.at_data;
    a.dom = job->template scratch<uint32_t>(8 * n, rc);
    a.fscal = job->template scratch<uint32_t>((size_t)(L + 2) * 8 * n, rc);
    a.partials = job->template scratch<uint32_t>((size_t)PV_NPARTS_MAX * 3 * N * n, rc);
    a.aff = job->template scratch<uint32_t>((size_t)5 * 2 * N * n, rc);
    a.fmiller = job->template scratch<uint32_t>((size_t)2 * 12 * N * n, rc);
    a.vtab = job->template scratch<uint32_t>((size_t)4 * G1_TAB * 2 * N * std::max<size_t>(n, 1), rc);   // T1's three tables + the one of D * r3^
    a.fixwk = FixTreeWork<C>{nullptr, nullptr, nullptr};
    if (ctx->fix_tree) {
        const size_t T = (size_t)(L + 2) * (size_t)ctx->hc.n_windows, nn1 = std::max<size_t>(n, 1);
        a.fixwk.pts0 = job->template scratch<uint32_t>(T * 2 * N * nn1, rc);
        a.fixwk.pts1 = job->template scratch<uint32_t>(((T + 1) / 2) * 2 * N * nn1, rc);
        a.fixwk.pre = job->template scratch<uint32_t>(std::max<size_t>(T / 2, 1) * N * nn1, rc);
    }
    if (rc) return rc;
    if ((rc = job->finish_setup_device())) return rc;
    a.status = job->d_status.template as<int8_t>();
    int8_t* pair_ok = job->template scratch<int8_t>(nn, rc);
    if (rc) return rc;
    job->zero_on_reset.push_back({pair_ok, nn});          // a pairing lane that never ran reads as "product != 1"
    if (!wire) {
        PvIngestArgs<C>& ia = job->ingest;
        ia.n = n; ia.L = L; ia.dst_too_long = ctx->dst_too_long ? 1 : 0;
        ia.rec = d32(0);
        ia.cm_off = d64(cm.at_off); ia.dm_off = d64(dm.at_off); ia.di_off = d64(di.at_off);
        ia.hdr_off64 = d64(hb.at_off); ia.ph_off64 = d64(pb.at_off);
        ia.cm = d32(cm.at_data); ia.dm = d32(dm.at_data); ia.di = d64(di.at_data);
        ia.pts = pts; ia.sc = sc; ia.slots = slots; ia.dmask = dmask; ia.didx = didx_s; ia.rcount = rcount;
        ia.hdr_off = offs; ia.hdr_len = offs + nn; ia.ph_off = offs + 2 * nn; ia.ph_len = offs + 3 * nn;
        ia.status0 = job->d_status0.template as<int8_t>();
        if (rt::launch<PvIngest<C>>(job->stream(), ia, n)) return BBS_E_HIP;
    } else {
        PvOctArgs<C>& oa = job->oct;
        oa.n = n; oa.L = L; oa.dst_too_long = ctx->dst_too_long ? 1 : 0;
        oa.oct = dimg + cm.at_data; oa.oct_off = d64(cm.at_off);
        oa.dm_off = d64(dm.at_off); oa.di_off = d64(di.at_off); oa.hdr_off64 = d64(hb.at_off); oa.ph_off64 = d64(pb.at_off);
        oa.dm = d32(dm.at_data); oa.di = d64(di.at_data);
        oa.msg_dst_too_long = 0;
        if (raw) {
            uint32_t* dmsc = hash_raw_messages<C>(job.get(), ctx, mb, nm, job->mh, oa.msg_dst_too_long, rc);
            if (rc) return rc;
            oa.dm = dmsc;
        }
        oa.pts = pts; oa.sc = sc; oa.slots = slots; oa.dmask = dmask; oa.didx = didx_s; oa.rcount = rcount;
        oa.hdr_off = offs; oa.hdr_len = offs + nn; oa.ph_off = offs + 2 * nn; oa.ph_len = offs + 3 * nn;
        oa.pcode = job->template scratch<int8_t>(3 * nn, rc);
        if (rc) return rc;
        oa.status0 = job->d_status0.template as<int8_t>();
        if (rt::launch<PvOctDecode<C>>(job->stream(), oa, 3 * n) || rt::launch<PvOctIngest<C>>(job->stream(), oa, n)) return BBS_E_HIP;
    }
    PairArgs<C>& pa = job->pa;
    pa.n = n; pa.cc = a.cc; pa.pa = a.pts; pa.pb = a.pts + (size_t)2 * NC * n; pa.negate_b = 1;
    pa.canonical = 1; pa.gate_arr = job->d_status0.template as<int8_t>(); pa.gate = ST_PENDING; pa.out = pair_ok;
    pa.fmiller = a.fmiller;
    job->fin.n = n; job->fin.status = a.status; job->fin.pair_ok = pair_ok;
    PvJob<C>* j = job.get();
    // (measured: letting the fixed-base lanes compute their own scalars and dropping the pv_scalars launch costs more than
    // the launch -- the hash and the Fr products land in the register-critical MSM kernel: 5.14 -> 5.33 ms, scratch 2.7 -> 3.1 KB)
    a.bv_dig = nullptr; a.bv_ppts = nullptr; a.bv_n_pad = 0;
    // The multi-scalar multiplication as separate kernels (stages.hpp, round 5).  The fixed-base chunks follow the scalars on
    // the main stream; where the doubling chains go is `layout`:
    //   3 : T1's chain on the job's second side stream, the single multiplications in front of the scalars on the main stream
    //       (two launches, each with its own budget; the main stream joins the side stream before the challenge stage)
    //   2 : all chains as ONE launch on the second side stream
    //   1 : all chains as one launch on the main stream behind the fixed-base chunks (a job that owns one hardware queue:
    //       batch verification's throughput form)
    // chains_behind_scalars (layout 2 only): the side stream forks behind the scalar stage instead of in front of it.  Batch
    // verification's latency form runs its bucket kernel (k_pip_window: workgroups of four wavefronts that need room on ONE
    // compute unit together) on the first side stream from the start; chains that start at the same moment spread a long
    // wavefront over every compute unit first and the bucket workgroups then wait for whole compute units (measured: that
    // kernel 1.3 -> 4.1 ms, the batch 4.8 -> 7.6 ms, profiles/r05_g_single_batch_forms.log).  0.14 ms later they come second.
    auto msm_chain = [j](int layout, bool chains_behind_scalars = false) {
        auto chains2 = [j]() { j->stages.push_back({"pv_chains", [j]() { return rt::launch<PvChains<C>>(j->stream_aux(2), j->a, j->n * PvChains<C>::units(j->a)); }, 2, 0}); };
        if (layout == 3) j->stages.push_back({"pv_t1_chain", [j]() { return rt::launch<PvT1Chain<C>>(j->stream_aux(2), j->a, j->n); }, 2, 0});
        if (layout == 2 && !chains_behind_scalars) chains2();
        if (layout == 3) j->stages.push_back({"pv_var_mul", [j]() { return rt::launch<PvVarMul<C>>(j->stream(), j->a, j->n * (size_t)(j->a.nvar - PvVarMul<C>::first_part(j->a))); }});
        j->stages.push_back({"pv_scalars", [j]() { return rt::launch<PvScalars<C>>(j->stream(), j->a, j->n); }});
        if (layout == 2 && chains_behind_scalars) chains2();
        if (j->a.fixwk.pts0) j->stages.push_back({"pv_fixed_tree", [j]() { return rt::launch<PvFixedTree<C>>(j->stream(), j->a, j->n); }});
        else j->stages.push_back({"pv_fixed_chunks", [j]() { return rt::launch<PvFixedChunk<C>>(j->stream(), j->a, j->n * (size_t)NFIX); }});
        if (layout == 1) j->stages.push_back({"pv_chains", [j]() { return rt::launch<PvChains<C>>(j->stream(), j->a, j->n * PvChains<C>::units(j->a)); }});
    };
    const int join_chains = 2;      // Stage::join bit of the second side stream (ignored where nothing was forked onto it)
    if (!ctx->batch_verify) {
        // every item its own pairing product, on the job's second stream concurrently with the MSM / challenge
        // stages: it needs only the proof's own points (canonical, converted in the kernel) and the flag the ingest stage
        // left.  First in the list: the second stream forks where its first stage stands, i.e. before the MSM chain.
        // One fused kernel in BOTH forms: the split of the two Miller loops (820 + 410 wavefronts) pays where the pairing
        // follows the MSM chain (verify: 7.4 -> 6.9 ms), but beside the 768 wavefronts of this operation's latency-form MSM
        // chain it oversubscribes the 1024 SIMDs and the queued wavefronts cost more than the split saves (measured
        // 5.1 ms split vs 4.4 ms fused, profiles/r03_j_latency_form_split.log)
        static const bool lat_split = []() { const char* v = getenv("BBS_PV_LAT_SPLIT"); return v && atoi(v) != 0; }();      // A/B knob
        add_pairing_stages<C>(j, &j->pa, 1, "pair_miller", "pair_final_exp", "pairing_6lane", false, 0, lat_split);
        msm_chain(pv_msm_layout(false, job->latency_form));
        j->stages.push_back({"pv_challenge", [j]() { return rt::launch<PvChallenge<C>>(j->stream(), j->a, j->n); }, 0, join_chains});
        j->stages.push_back({"pv_finish", [j]() { return rt::launch<PvFinish>(j->stream(), j->fin, j->n); }, 0, 1});
    } else {
        // batch verification (pippenger.hpp): combined pairing checks instead of n products; if one fails, the per-item
        // kernel decides the items still pending (its lanes write Ok(true) and return at once otherwise).
        //  * latency form: the combination runs on the job's second stream BESIDE the MSM / challenge stages, over the
        //    structurally valid items (it needs only the proof's own points) -- the shortest path for a job that is alone;
        //  * throughput form: the combination FOLLOWS the challenge stage on the one stream, over the items whose challenge
        //    matched, and the challenge stage itself prepares it (PvChallengeBv).  Six launches per job, one hardware queue:
        //    small kernels queue behind the long wavefronts of other jobs, so every launch saved is latency saved.
        const int aux = job->latency_form ? 1 : 0;
        if (aux && (rc = add_batch_combination<C>(j, &j->bv, ctx, n, a.cc, a.pts, a.pts + (size_t)2 * NC * n, 1,
                                                  job->d_status0.template as<int8_t>(), ST_PENDING, 1, 1, true))) return rc;
        msm_chain(pv_msm_layout(!aux, job->latency_form), aux != 0);
        if (aux) {
            j->stages.push_back({"pv_challenge", [j]() { return rt::launch<PvChallenge<C>>(j->stream(), j->a, j->n); }, 0, join_chains});
        } else {
            // (a_bar, b_bar in Montgomery form were stored by PvT1Chain; the challenge stage writes digits and points)
            j->stages.push_back({"pv_challenge", [j]() { return rt::launch<PvChallengeBv<C>>(j->stream(), j->a, j->n); }, 0, join_chains});
            if ((rc = add_batch_combination<C>(j, &j->bv, ctx, n, a.cc, a.aff, a.aff + (size_t)2 * N * n, 0, a.status, ST_PAIRING, 1, 0, false))) return rc;
            a.bv_dig = j->bv.prep.dig; a.bv_ppts = j->bv.prep.ppts; a.bv_n_pad = j->bv.prep.n_pad;
            std::memcpy(a.bv_seed, j->bv.prep.seed, sizeof(a.bv_seed));
        }
        pa.gate_arr = a.status; pa.gate = ST_PAIRING; pa.out = a.status;          // fallback: the items still pending, verdict in place
        add_batch_decision<C>(j, &j->bv, &j->pa, aux);
    }
    *out = job.release();
    return BBS_OK;
}

