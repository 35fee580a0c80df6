// Host orchestration of one batched operation (template over the curve); instantiated by the
// per-curve translation units tu_*.hip so the library builds in parallel.
#pragma once
#include "runtime.hpp"

template <class C>
int Ctx<C>::set_generators(const uint8_t* g, size_t count, const uint8_t* aid, size_t aid_len) {
    if (count < 1 || (!g) || (aid_len && !aid)) return BBS_E_ARG;
    if (use()) return BBS_E_HIP;
    std::vector<G1Aff<C>> tmp(count);
    for (size_t k = 0; k < count; k++) {
        if (!fe_from_le_bytes<typename C::FpP>(g + k * 2 * FPB, tmp[k].x)) return BBS_E_ARG;
        if (!fe_from_le_bytes<typename C::FpP>(g + k * 2 * FPB + FPB, tmp[k].y)) return BBS_E_ARG;
        if (!g1a_on_curve<C>(tmp[k])) return BBS_E_ARG;
    }
    // Nothing of the context changes until the new tables have been BUILT: a failure below (no device memory, a HIP error)
    // leaves the previous generator set, its tables and gens_set as they were.
    const int L_new = (int)count - 1;
    const int nb = L_new + 2;                    // fixed-base tables over [P1, Q1, H_1..H_L]
    auto table_bytes_at = [&](int c) { return (size_t)nb * ((256 + c - 1) / c) * ((size_t)1 << (c - 1)) * fix_tab_stride<C>() * 4; };
    // candidate widths, widest first.  A requested width is the only candidate.  Automatic: the widest of {20, 16, 12, 8}
    // whose tables fit 1/32 of the memory now free on the device and 8 GiB -- at 32 messages that is 16 bits (2 GB, 16
    // additions per scalar); 20 bits (26 GB, 13 additions, +3.5 % proof_verify/s) is a deliberate choice for a verifier with
    // ONE issuer and memory to spare (bench.py asks for it), not a default: a verifier of many issuers or message counts
    // keeps many table sets -- and, because the free figure is a snapshot (other ranks or processes on the device,
    // fragmentation), every narrower width after it as a fallback when the allocation itself fails.
    std::vector<int> widths;
    if (win_bits_requested) widths.push_back(win_bits_requested);
    else {
        const size_t free_b = rt::mem_free_bytes();
        const size_t budget = std::min<size_t>(free_b / 32, (size_t)8 << 30);
        for (int c : {20, 16, 12, 8}) if (c == 8 || table_bytes_at(c) <= budget) widths.push_back(c);
    }
    std::vector<uint32_t> bases((size_t)nb * 2 * N);
    auto put = [&](size_t k, const G1Aff<C>& p) {
        for (int j = 0; j < N; j++) { bases[k * 2 * N + j] = p.x.v[j]; bases[k * 2 * N + N + j] = p.y.v[j]; }
    };
    put(0, hc.p1);
    for (size_t k = 0; k < tmp.size(); k++) put(1 + k, tmp[k]);
    DevBuf n_bases, n_winbase, n_tables;
    if (n_bases.alloc(bases.size() * 4)) return BBS_E_NOMEM;
    int wb = 0, W = 0;
    size_t per_win = 0;
    for (int c : widths) {
        W = (256 + c - 1) / c;
        per_win = (size_t)1 << (c - 1);                         // signed digits: |digit| = 1 .. 2^(c-1)
        if (n_winbase.alloc((size_t)nb * W * 2 * N * 4) == 0 && n_tables.alloc((size_t)nb * W * per_win * fix_tab_stride<C>() * 4) == 0) { wb = c; break; }
        n_winbase.release(); n_tables.release();
    }
    if (!wb) return BBS_E_NOMEM;
    if (rt::h2d(n_bases.p, bases.data(), bases.size() * 4, stream)) return BBS_E_HIP;
    TabArgs<C> ta;
    ta.n_bases = nb; ta.win_bits = wb; ta.n_windows = W;
    ta.bases = n_bases.as<uint32_t>(); ta.winbase = n_winbase.as<uint32_t>(); ta.tables = n_tables.as<uint32_t>();
    if (rt::launch<TabWinBase<C>>(stream, ta, (size_t)nb)) return BBS_E_HIP;
    if (rt::launch<TabEntry<C>>(stream, ta, (size_t)nb * W * per_win)) return BBS_E_HIP;
    if (rt::sync(stream)) return BBS_E_HIP;
    // ---- commit
    d_bases.swap(n_bases); d_winbase.swap(n_winbase); d_tables.swap(n_tables);      // (the previous buffers are released with the locals)
    gens.swap(tmp);
    L = L_new;
    api_id.assign(aid, aid + aid_len);
    gens_set = true;
    win_bits = wb;
    hc.L = L; hc.n_bases = nb; hc.win_bits = win_bits; hc.n_windows = W;
    for (int j = 0; j < 8; j++) hc.fix_bias[j] = 0;
    for (int w = 0; w + 1 < W; w++) { const int b = win_bits * w + win_bits - 1; hc.fix_bias[b >> 5] |= 1u << (b & 31); }
    rebuild_hash();
    return BBS_OK;
}


template <class C>
int h2s_batch(Ctx<C>* ctx, size_t n, const uint8_t* msgs, const uint64_t* off, const uint8_t* dst, size_t dst_len, uint8_t* out) {
    if (dst_len > 255 || !out || (n && !off)) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    BytePool bp;
    if (!bp.build(n, msgs, off)) return BBS_E_ARG;
    DevBuf d_off, d_len, d_bytes, d_out;
    if (d_off.alloc(n * 4 + 4) || d_len.alloc(n * 4 + 4) || d_bytes.alloc(bp.bytes.size()) || d_out.alloc(n * 32 + 4)) return BBS_E_NOMEM;
    if (rt::h2d(d_off.p, bp.off.data(), n * 4, ctx->stream) || rt::h2d(d_len.p, bp.len.data(), n * 4, ctx->stream) ||
        rt::h2d(d_bytes.p, bp.bytes.data(), bp.bytes.size(), ctx->stream)) return BBS_E_HIP;
    H2sArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = n; a.off = d_off.as<uint32_t>(); a.len = d_len.as<uint32_t>(); a.bytes = d_bytes.as<uint8_t>();
    if (dst_len) std::memcpy(a.dst, dst, dst_len);
    a.dst_len = (uint32_t)dst_len; a.out = d_out.as<uint32_t>();
    if (rt::launch<H2sItem<C>>(ctx->stream, a, n) || rt::sync(ctx->stream)) return BBS_E_HIP;
    std::vector<uint32_t> w(n * 8);
    if (rt::d2h(w.data(), d_out.p, n * 32, ctx->stream)) return BBS_E_HIP;
    for (size_t i = 0; i < n; i++) unpack_words_le(w, n, 0, i, 8, out + i * 32);
    return BBS_OK;
}
template <class C>
int msm_batch(Ctx<C>* ctx, size_t n, const uint8_t* fs, size_t nf, const uint8_t* vp, const uint8_t* vs, size_t nv,
                     uint8_t* out, int8_t* status) {
    constexpr int N = C::FpP::N;
    constexpr int NC = C::FpP::NC;
    constexpr int FPB = 4 * NC;
    using R = typename C::FrP;
    if (!ctx->gens_set) return BBS_E_STATE;
    if (nf > (size_t)ctx->L + 2 || !out || !status || (nf && !fs) || (nv && (!vp || !vs))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    Soa F, VP, VS;
    F.init(std::max<size_t>(nf, 1) * 8, n); VP.init(std::max<size_t>(nv, 1) * 2 * NC, n); VS.init(std::max<size_t>(nv, 1) * 8, n);
    std::vector<int8_t> st0(n, ST_PENDING);
    for (size_t i = 0; i < n; i++) {
        bool ok = true;
        for (size_t k = 0; k < nf; k++) ok &= pack_fe<R>(F, k * 8, i, fs + (i * nf + k) * 32);
        for (size_t k = 0; k < nv; k++) {
            ok &= pack_g1<C>(VP, k * 2 * NC, i, vp + (i * nv + k) * 2 * FPB);
            ok &= pack_fe<R>(VS, k * 8, i, vs + (i * nv + k) * 32);
        }
        if (!ok) st0[i] = BBS_ST_NONCANONICAL;
    }
    DevBuf dF, dVP, dVS, dSt, dPart, dOut;
    if (dF.alloc(F.bytes()) || dVP.alloc(VP.bytes()) || dVS.alloc(VS.bytes()) || dSt.alloc(n + 4) ||
        dPart.alloc((nv + NFIX) * 3 * N * n * 4 + 4) || dOut.alloc((size_t)2 * NC * n * 4 + 4)) return BBS_E_NOMEM;
    if (rt::h2d(dF.p, F.soa().data(), F.bytes(), ctx->stream) || rt::h2d(dVP.p, VP.soa().data(), VP.bytes(), ctx->stream) ||
        rt::h2d(dVS.p, VS.soa().data(), VS.bytes(), ctx->stream) || rt::h2d(dSt.p, st0.data(), n, ctx->stream)) return BBS_E_HIP;
    int rc = ctx->sync_consts();
    if (rc) return rc;
    MsmArgs<C> a;
    a.n = n; a.n_fixed = (int)nf; a.n_var = (int)nv; a.cc = ctx->d_consts.template as<CtxConsts<C>>();
    a.glv = (C::K::HAS_GLV && (C::K::GLV_ALWAYS || ctx->points_in_subgroup)) ? 1 : 0;
    a.fscal = dF.as<uint32_t>(); a.vpts = dVP.as<uint32_t>(); a.vscal = dVS.as<uint32_t>();
    a.status = dSt.as<int8_t>(); a.partials = dPart.as<uint32_t>(); a.out = dOut.as<uint32_t>();
    a.fixwk = FixTreeWork<C>{nullptr, nullptr, nullptr};
    DevBuf dW0, dW1, dW2;
    if (ctx->fix_tree && n) {
        const size_t T = nf * (size_t)ctx->hc.n_windows;
        if (dW0.alloc(std::max<size_t>(T, 1) * 2 * N * n * 4) || dW1.alloc(std::max<size_t>((T + 1) / 2, 1) * 2 * N * n * 4) ||
            dW2.alloc(std::max<size_t>(T / 2, 1) * N * n * 4)) return BBS_E_NOMEM;
        a.fixwk = FixTreeWork<C>{dW0.as<uint32_t>(), dW1.as<uint32_t>(), dW2.as<uint32_t>()};
    }
    DevBuf dTab;
    if (dTab.alloc(std::max<size_t>(nv, 1) * G1_TAB * 2 * N * std::max<size_t>(n, 1) * 4)) return BBS_E_NOMEM;
    a.vtab = dTab.as<uint32_t>();
    if (rt::launch<MsmVarMul<C>>(ctx->stream, a, n * nv)) return BBS_E_HIP;
    if (a.fixwk.pts0 ? rt::launch<MsmFixedTree<C>>(ctx->stream, a, n) : rt::launch<MsmFixedChunk<C>>(ctx->stream, a, n * NFIX)) return BBS_E_HIP;
    if (rt::launch<MsmCombine<C>>(ctx->stream, a, n) || rt::sync(ctx->stream)) return BBS_E_HIP;
    std::vector<uint32_t> w((size_t)2 * NC * n);
    if (rt::d2h(w.data(), dOut.p, w.size() * 4, ctx->stream) || rt::d2h(status, dSt.p, n, ctx->stream)) return BBS_E_HIP;
    if (!statuses_final(status, n)) return BBS_E_STATE;
    for (size_t i = 0; i < n; i++) {
        if (status[i] == 1) unpack_words_le(w, n, 0, i, 2 * NC, out + i * 2 * FPB);
        else std::memset(out + i * 2 * FPB, 0, 2 * FPB);
    }
    return BBS_OK;
}
// S = sum_i k_i * P_i over n per-item points by the bucket method (pippenger.hpp); items whose encoding is not
// canonical (-40) or whose point is not on the curve (-41) are reported in status and contribute nothing
template <class C>
int msm_pippenger(Ctx<C>* ctx, size_t n, const uint8_t* pts, const uint8_t* scal, uint8_t* out, int* out_inf, int8_t* status) {
    constexpr int N = C::FpP::N;
    constexpr int NC = C::FpP::NC;
    constexpr int FPB = 4 * NC;
    constexpr int NW = 32;
    using R = typename C::FrP;
    if (!out || !out_inf || !status || (n && (!pts || !scal))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    Soa P, S;
    P.init(2 * NC, std::max<size_t>(n, 1)); S.init(8, std::max<size_t>(n, 1));
    std::vector<int8_t> st0(std::max<size_t>(n, 1), ST_PENDING);
    for (size_t i = 0; i < n; i++) {
        bool ok = pack_g1<C>(P, 0, i, pts + i * 2 * FPB);
        ok &= pack_fe<R>(S, 0, i, scal + i * 32);
        if (!ok) st0[i] = BBS_ST_NONCANONICAL;
    }
    const size_t n_pad = (n + 3) & ~(size_t)3;
    DevBuf dP, dS, dSt, dPm, dDig, dW, dOut;
    if (dP.alloc(P.bytes()) || dS.alloc(S.bytes()) || dSt.alloc(n + 4) || dPm.alloc((size_t)2 * N * n * 4 + 16) ||
        dDig.alloc((size_t)NW * n_pad + 4) || dW.alloc((size_t)3 * N * NW * 4) || dOut.alloc((size_t)2 * N * 4))
        return BBS_E_NOMEM;
    if (rt::h2d(dP.p, P.soa().data(), P.bytes(), ctx->stream) || rt::h2d(dS.p, S.soa().data(), S.bytes(), ctx->stream) ||
        rt::h2d(dSt.p, st0.data(), n, ctx->stream) || rt::dmemset(dDig.p, 0, (size_t)NW * n_pad, ctx->stream)) return BBS_E_HIP;
    PipPrep<C> prep{dP.as<uint32_t>(), dPm.as<uint32_t>(), dSt.as<int8_t>(), n};
    PipDigitArgs da{n, n_pad, dS.as<uint32_t>(), dSt.as<int8_t>(), dDig.as<uint8_t>()};
    PipArgs<C> a{};
    a.n = n; a.n_pad = n_pad; a.M = 1; a.NW = NW; a.ppts = dPm.as<uint32_t>(); a.dig = dDig.as<uint8_t>();
    a.wins = dW.as<uint32_t>(); a.out = dOut.as<uint32_t>();
    if (rt::launch<PipPrep<C>>(ctx->stream, prep, n) || rt::launch<PipDigits>(ctx->stream, da, n)) return BBS_E_HIP;
#ifdef BBS_HOST_TWIN
    const size_t T = (size_t)NW * PIP_NB;
    DevBuf dList, dB, dSeg;
    if (dList.alloc((size_t)NW * n * 4 + 4) || dB.alloc((size_t)3 * N * T * 4) || dSeg.alloc((size_t)3 * N * NW * (PIP_NB / PIP_SEG) * 4)) return BBS_E_NOMEM;
    a.list = dList.as<uint32_t>(); a.buckets = dB.as<uint32_t>(); a.segs = dSeg.as<uint32_t>();
    if (rt::launch<PipBuckets<C>>(ctx->stream, a, T) || rt::launch<PipSegments<C>>(ctx->stream, a, (size_t)NW * (PIP_NB / PIP_SEG)) ||
        rt::launch<PipWindows<C>>(ctx->stream, a, (size_t)NW) || rt::launch<PipFinal<C>>(ctx->stream, a, 1) || rt::sync(ctx->stream))
        return BBS_E_HIP;
#else
    // the workgroup-cooperative kernel (pippenger.hpp): one workgroup per (window, tile of 4096 items), then the tiles'
    // window sums shifted by 2^(8 w) and added
    PipCoopArgs<C> co{};
    co.n = n; co.n_pad = n_pad; co.M = 1; co.NW = NW; co.n_tiles = (int)((std::max<size_t>(n, 1) + PIP_TILE - 1) / PIP_TILE);
    co.ppts = dPm.as<uint32_t>(); co.dig = dDig.as<uint8_t>();
    DevBuf dTile;
    if (dTile.alloc((size_t)3 * N * NW * co.n_tiles * 4)) return BBS_E_NOMEM;
    co.tile_sums = dTile.as<uint32_t>();
    co.out_aff = nullptr;
    PipTileSumArgs<C> ts{1, NW, n ? co.n_tiles : 0, 1, co.tile_sums, nullptr, a.wins};
    if (rt::launch_pip_windows<C>(ctx->stream, co) || rt::launch<PipTileSums<C>>(ctx->stream, ts, (size_t)NW) ||
        rt::launch<PipFinal<C>>(ctx->stream, a, 1) || rt::sync(ctx->stream))
        return BBS_E_HIP;
#endif
    std::vector<uint32_t> w((size_t)2 * N);
    if (rt::d2h(w.data(), dOut.p, w.size() * 4, ctx->stream) || rt::d2h(status, dSt.p, n, ctx->stream)) return BBS_E_HIP;
    if (!statuses_final(status, n)) return BBS_E_STATE;
    G1Aff<C> r;
    for (int j = 0; j < N; j++) { r.x.v[j] = w[j]; r.y.v[j] = w[N + j]; }
    *out_inf = g1a_is_inf<C>(r) ? 1 : 0;
    if (*out_inf) std::memset(out, 0, 2 * FPB);
    else { fe_to_le_bytes<typename C::FpP>(r.x, out); fe_to_le_bytes<typename C::FpP>(r.y, out + FPB); }
    return BBS_OK;
}

// n compressed G1 points -> affine records (identity = zeros) with the device's decode stage; code[i]: 0 ok,
// 1 ok and the identity, -40 malformed / non-canonical, -41 not on the curve / not in the prime-order subgroup
template <class C>
int g1_decompress_batch(Ctx<C>* ctx, size_t n, const uint8_t* in, uint8_t* out, int8_t* code) {
    constexpr int NC = C::FpP::NC;
    constexpr size_t NB = 4 * NC;
    if (!code || (n && (!in || !out))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    DevBuf dIn, dOut, dCode;
    if (dIn.alloc(std::max<size_t>(n, 1) * NB) || dOut.alloc((size_t)2 * NC * std::max<size_t>(n, 1) * 4) || dCode.alloc(n + 4)) return BBS_E_NOMEM;
    if (rt::h2d(dIn.p, in, n * NB, ctx->stream)) return BBS_E_HIP;
    G1DecodeArgs<C> a{n, dIn.as<uint8_t>(), dOut.as<uint32_t>(), dCode.as<int8_t>()};
    if (rt::launch<G1Decode<C>>(ctx->stream, a, n) || rt::sync(ctx->stream)) return BBS_E_HIP;
    std::vector<uint32_t> w((size_t)2 * NC * std::max<size_t>(n, 1));
    if (n && (rt::d2h(w.data(), dOut.p, (size_t)2 * NC * n * 4, ctx->stream) || rt::d2h(code, dCode.p, n, ctx->stream))) return BBS_E_HIP;
    for (size_t i = 0; i < n; i++) {
        if (code[i] == 0) unpack_words_le(w, n, 0, i, 2 * NC, out + i * 2 * NB);
        else std::memset(out + i * 2 * NB, 0, 2 * NB);
    }
    return BBS_OK;
}

// n signatures as octet strings (compress(A) || e, fp_bytes + 32 each) -> records A || e; the point work on the device.
// status[i] = 1 or the code bbs_signature_from_octets gives for item i (same order of checks)
template <class C>
int signatures_from_octets_batch(Ctx<C>* ctx, size_t n, const uint8_t* oct, uint8_t* rec_out, int8_t* status) {
    constexpr size_t NB = 4 * C::FpP::NC, olen = NB + 32, rlen = 2 * NB + 32;
    using R = typename C::FrP;
    if (!status || (n && (!oct || !rec_out))) return BBS_E_ARG;
    std::vector<uint8_t> cp(std::max<size_t>(n, 1) * NB), aff(std::max<size_t>(n, 1) * 2 * NB);
    for (size_t i = 0; i < n; i++) std::memcpy(cp.data() + i * NB, oct + i * olen, NB);
    std::vector<int8_t> code(n + 1);
    const int rc = g1_decompress_batch<C>(ctx, n, cp.data(), aff.data(), code.data());
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) {
        uint8_t* r = rec_out + i * rlen;
        std::memset(r, 0, rlen);
        if (code[i] < 0) { status[i] = code[i]; continue; }
        if (code[i] == 1) { status[i] = BBS_ST_INVALID_ENCODING; continue; }        // A must not be the identity
        uint32_t w[8];
        uint8_t* e = r + 2 * NB;
        for (int k = 0; k < 32; k++) e[k] = oct[i * olen + NB + 31 - k];
        bool zero = true;
        for (int k = 0; k < 8; k++) { w[k] = le32(e + 4 * k); zero &= w[k] == 0; }
        if (!limbs_lt_mod<R>(w)) { status[i] = BBS_ST_NONCANONICAL; std::memset(e, 0, 32); continue; }
        if (zero) { status[i] = BBS_ST_INVALID_ENCODING; continue; }                // e = 0 is rejected
        std::memcpy(r, aff.data() + i * 2 * NB, 2 * NB);
        status[i] = 1;
    }
    return BBS_OK;
}

// n proofs as octet strings -> the records of bbs_core_proof_verify_*: lengths and scalars on the host, the 3 n
// compressed points on the device (codec_dev.hpp).  status[i] = 1 or the code bbs_proof_from_octets gives for item i
// (same order of checks).  commit_off_out: n + 1 entries; commitments_out holds sum_i U_i scalars.
template <class C>
int proofs_from_octets_batch(Ctx<C>* ctx, size_t n, const uint8_t* oct, const uint64_t* off, uint8_t* pf_out, uint8_t* cm_out,
                             uint64_t* cm_off_out, int8_t* status) {
    constexpr int NC = C::FpP::NC;
    constexpr size_t NB = 4 * NC;
    using R = typename C::FrP;
    if (!status || !cm_off_out || (n && (!oct || !off || !pf_out))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    const size_t fixed = 3 * NB + 4 * 32, rec = 6 * NB + 128;
    std::vector<uint8_t> cpts(std::max<size_t>(3 * n, 1) * NB, 0);
    cm_off_out[0] = 0;
    auto be_to_le = [](const uint8_t* be, uint8_t* le) {
        uint32_t w[8];
        for (int i = 0; i < 32; i++) le[i] = be[31 - i];
        for (int i = 0; i < 8; i++) w[i] = le32(le + 4 * i);
        return limbs_lt_mod<R>(w);
    };
    for (size_t i = 0; i < n; i++) {
        const size_t len = (size_t)(off[i + 1] - off[i]);
        const uint8_t* o = oct + off[i];
        status[i] = 1;
        cm_off_out[i + 1] = cm_off_out[i];
        std::memset(pf_out + i * rec, 0, rec);
        if (off[i + 1] < off[i] || len < fixed || (len - fixed) % 32) { status[i] = BBS_ST_INVALID_ENCODING; continue; }
        const size_t u = (len - fixed) / 32;
        for (int p = 0; p < 3; p++) std::memcpy(cpts.data() + (i * 3 + p) * NB, o + (size_t)p * NB, NB);
        // scalars (their verdict is applied after the points', as in bbs_proof_from_octets)
        const uint8_t* s = o + 3 * NB;
        bool ok = true;
        for (int k = 0; k < 3; k++) ok &= be_to_le(s + 32 * k, pf_out + i * rec + 6 * NB + 32 * k);
        if (cm_out) for (size_t k = 0; k < u; k++) ok &= be_to_le(s + 96 + 32 * k, cm_out + (cm_off_out[i] + k) * 32);
        ok &= be_to_le(s + 96 + 32 * u, pf_out + i * rec + 6 * NB + 96);
        cm_off_out[i + 1] = cm_off_out[i] + u;
        if (!ok) status[i] = BBS_ST_NONCANONICAL;       // provisional: a point failure reported first
    }
    const size_t np = 3 * n;
    DevBuf dIn, dOut, dCode;
    if (dIn.alloc(cpts.size()) || dOut.alloc((size_t)2 * NC * std::max<size_t>(np, 1) * 4) || dCode.alloc(np + 4)) return BBS_E_NOMEM;
    if (rt::h2d(dIn.p, cpts.data(), cpts.size(), ctx->stream)) return BBS_E_HIP;
    G1DecodeArgs<C> a{np, dIn.as<uint8_t>(), dOut.as<uint32_t>(), dCode.as<int8_t>()};
    if (rt::launch<G1Decode<C>>(ctx->stream, a, np) || rt::sync(ctx->stream)) return BBS_E_HIP;
    std::vector<uint32_t> w((size_t)2 * NC * std::max<size_t>(np, 1));
    std::vector<int8_t> code(np + 4);
    if (np && (rt::d2h(w.data(), dOut.p, (size_t)2 * NC * np * 4, ctx->stream) || rt::d2h(code.data(), dCode.p, np, ctx->stream))) return BBS_E_HIP;
    for (size_t i = 0; i < n; i++) {
        if (status[i] == BBS_ST_INVALID_ENCODING) continue;
        int8_t verdict = status[i];
        for (int p = 2; p >= 0; p--) {                   // the first failing point (lowest p) decides
            const int8_t c = code[i * 3 + p];
            if (c == 1) verdict = BBS_ST_INVALID_ENCODING;              // identity points are rejected
            else if (c < 0) verdict = c;
        }
        status[i] = verdict;
        if (verdict == 1)
            for (int p = 0; p < 3; p++) unpack_words_le(w, np, 0, i * 3 + p, 2 * NC, pf_out + i * rec + (size_t)p * 2 * NB);
        else std::memset(pf_out + i * rec, 0, rec);
    }
    return BBS_OK;
}

template <class C>
int pairing_batch(Ctx<C>* ctx, size_t n, const uint8_t* pa, const uint8_t* pb, int8_t* status) {
    constexpr int N = C::FpP::N;
    constexpr int NC = C::FpP::NC;
    constexpr int FPB = 4 * NC;
    if (!ctx->pk_set) return BBS_E_STATE;
    if (!status || (n && (!pa || !pb))) return BBS_E_ARG;
    if (ctx->use()) return BBS_E_HIP;
    Soa A, B;
    A.init(2 * NC, n); B.init(2 * NC, n);
    std::vector<int8_t> st0(n, ST_PENDING);
    for (size_t i = 0; i < n; i++) {
        bool ok = pack_g1<C>(A, 0, i, pa + i * 2 * FPB) & pack_g1<C>(B, 0, i, pb + i * 2 * FPB);
        if (!ok) st0[i] = BBS_ST_NONCANONICAL;
    }
    DevBuf dA, dB, dAm, dBm, dSt, dF;
    if (dA.alloc(A.bytes()) || dB.alloc(B.bytes()) || dAm.alloc((size_t)2 * N * n * 4 + 4) || dBm.alloc((size_t)2 * N * n * 4 + 4) || dSt.alloc(n + 4) ||
        dF.alloc((size_t)2 * 12 * N * n * 4 + 4)) return BBS_E_NOMEM;
    if (rt::h2d(dA.p, A.soa().data(), A.bytes(), ctx->stream) || rt::h2d(dB.p, B.soa().data(), B.bytes(), ctx->stream) ||
        rt::h2d(dSt.p, st0.data(), n, ctx->stream)) return BBS_E_HIP;
    int rc = ctx->sync_consts();
    if (rc) return rc;
    PairPrep<C> pp{dA.as<uint32_t>(), dB.as<uint32_t>(), dAm.as<uint32_t>(), dBm.as<uint32_t>(), dSt.as<int8_t>(), n};
    PairArgs<C> a{};
    a.n = n; a.cc = ctx->d_consts.template as<CtxConsts<C>>(); a.pa = dAm.as<uint32_t>(); a.pb = dBm.as<uint32_t>();
    a.negate_b = 0; a.canonical = 0; a.gate_arr = dSt.as<int8_t>(); a.gate = ST_PAIRING; a.out = dSt.as<int8_t>(); a.fmiller = dF.as<uint32_t>();
#ifdef BBS_HOST_TWIN
    if (rt::launch<PairPrep<C>>(ctx->stream, pp, n) || rt::launch<PairMiller<C>>(ctx->stream, a, n * 2) ||
        rt::launch<PairFinal<C>>(ctx->stream, a, n) || rt::sync(ctx->stream)) return BBS_E_HIP;
#else
    if (rt::launch<PairPrep<C>>(ctx->stream, pp, n) ||
        rt::launch<PairDist<C>>(ctx->stream, a, ((n + GRP_PER_WAVE - 1) / GRP_PER_WAVE) * 64) || rt::sync(ctx->stream)) return BBS_E_HIP;
#endif
    if (rt::d2h(status, dSt.p, n, ctx->stream)) return BBS_E_HIP;
    return statuses_final(status, n) ? BBS_OK : BBS_E_STATE;
}

// GPU self-test of the lane-sliced Fp12: the reference result is computed by the SAME one-lane
// tower code compiled for the host (tower.hpp / pairing.hpp, validated against the oracle by the
// host-twin tests); only the six-lane version runs on the device.
template <class C, int OP>
static Fp12<C> selftest_ref(const Fp12<C>& x, const Fp12<C>& y, const CtxConsts<C>& hc, const G1Aff<C>& P) {
    if constexpr (OP == 0) return f12_mul<C>(x, y);
    else if constexpr (OP == 1) return f12_frob<C, 1>(x);
    else if constexpr (OP == 2) return f12_frob<C, 2>(x);
    else if constexpr (OP == 3) return f12_frob<C, 3>(x);
    else if constexpr (OP == 4) return f12_inv<C>(x);
    else if constexpr (OP == 5) return f12_conj<C>(x);
    else if constexpr (OP == 6) return f12_mul_line<C>(x, hc.tab_bp2.e[3], P);
    else if constexpr (OP == 7) return final_exponentiation<C>(x);
    else if constexpr (OP == 10) return f12_sqr<C>(x);
    else if constexpr (OP == 8) return f12_sqr<C>(x);
    else if constexpr (OP == 11) return f12_pow_x<C>(x);
    else return x;
}

template <class C>
int selftest_f12(Ctx<C>* ctx, int op, const uint8_t* a_le, const uint8_t* b_le, uint8_t* out_single, uint8_t* out_dist) {
#ifdef BBS_HOST_TWIN
    (void)ctx; (void)op; (void)a_le; (void)b_le; (void)out_single; (void)out_dist;
    return BBS_E_ARG;
#else
    constexpr int N = C::FpP::N;
    constexpr int FPB = 4 * C::FpP::NC;
    using P = typename C::FpP;
    if (ctx->use()) return BBS_E_HIP;
    Fp<C> xe[12], ye[12];
    for (int k = 0; k < 12; k++)
        if (!fe_from_le_bytes<P>(a_le + k * FPB, xe[k]) || !fe_from_le_bytes<P>(b_le + k * FPB, ye[k])) return BBS_E_ARG;
    Fp12<C> x = f12_from_array<C>(xe), y = f12_from_array<C>(ye);
    if (op >= 10) {   // cyclotomic input
        x = f12_mul<C>(f12_conj<C>(x), f12_inv<C>(x));
        x = f12_mul<C>(f12_frob<C, 2>(x), x);
    }
    G1Aff<C> Pt = {ye[0], ye[1]};
    Fp12<C> rs;
    int lrc = -1;
    std::vector<uint32_t> X(12 * N), B(12 * N);
    f12_to_array<C>(x, xe);
    for (int k = 0; k < 12; k++) for (int j = 0; j < N; j++) { X[k * N + j] = xe[k].v[j]; B[k * N + j] = ye[k].v[j]; }
    DevBuf dB, dD;
    if (dB.alloc(B.size() * 4) || dD.alloc(X.size() * 4)) return BBS_E_NOMEM;
    if (rt::h2d(dB.p, B.data(), B.size() * 4, ctx->stream) || rt::h2d(dD.p, X.data(), X.size() * 4, ctx->stream)) return BBS_E_HIP;
    int rc = ctx->sync_consts();
    if (rc) return rc;
    SelfTestArgs<C> a{op, ctx->d_consts.template as<CtxConsts<C>>(), nullptr, dB.as<uint32_t>(), nullptr, dD.as<uint32_t>()};
#define BBS_ST_CASE(K) case K: rs = selftest_ref<C, K>(x, y, ctx->hc, Pt); lrc = rt::launch<SelfTestDist<C, K>>(ctx->stream, a, 64); break;
    switch (op) {
        BBS_ST_CASE(0) BBS_ST_CASE(1) BBS_ST_CASE(2) BBS_ST_CASE(3) BBS_ST_CASE(4) BBS_ST_CASE(5)
        BBS_ST_CASE(6) BBS_ST_CASE(7) BBS_ST_CASE(8) BBS_ST_CASE(10) BBS_ST_CASE(11)
        default: return BBS_E_ARG;
    }
#undef BBS_ST_CASE
    if (lrc || rt::sync(ctx->stream)) return BBS_E_HIP;
    std::vector<uint32_t> D(12 * N);
    if (rt::d2h(D.data(), dD.p, D.size() * 4, ctx->stream)) return BBS_E_HIP;
    Fp<C> se[12];
    f12_to_array<C>(rs, se);
    for (int k = 0; k < 12; k++) {
        Fe<P> yv;
        for (int j = 0; j < N; j++) yv.v[j] = D[k * N + j];
        fe_to_le_bytes<P>(se[k], out_single + k * FPB);
        fe_to_le_bytes<P>(yv, out_dist + k * FPB);
    }
    return BBS_OK;
#endif
}
