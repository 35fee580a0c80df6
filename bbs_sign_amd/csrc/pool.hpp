// core_proof_verify over a LIST of proofs, fanned out over SEVERAL GPUs behind the C ABI ("multi-GPU
// fan-out is internal"; 8(e): shard by curve, then contiguous ranges per GPU, tables replicated, no data-path collective).
//
// The reference verifies one proof per call (src/proof_verify.rs:19-61, core form :64-116); a caller with a list loops.
// A bbs_pool owns one context per (curve, device) -- the same generators and issuer key on every device -- and
// bbs_pool_proof_verify takes the list as one section per curve in the layout of bbs_core_proof_verify_batch, cuts every
// section into contiguous shares (ceil(n / devices) items per device: sharding.shard_plan's rule), every share into jobs of
// at most `max_batch` items, and runs the jobs of a device from ONE submitting thread per device through
// bbs_core_proof_verify_submit (completion-order retire, curves alternating so that BN254 and BLS12-381 jobs overlap on the
// chip).  The statuses land straight in the caller's array, in the caller's order: one process drives every GPU, so the
// exchange that the one-process-per-GPU launcher (bbs_sign_amd/mixed.py, RCCL all_gather) needs does not exist here.
//
// Plain host code on top of the C ABI of include/bbs_sign_amd.h (no field arithmetic, no HIP calls here).
#pragma once
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bbs_sign_amd.h"

struct bbs_pool {
    std::vector<int> devices;                         // device ids, one entry per member (an id may repeat: two context sets on one GPU)
    // ctx[curve][member]; created on the first configuration call that names the curve
    std::vector<bbs_ctx*> ctx[2];
    int inflight = 6;                                 // jobs outstanding per device (the serving loop's plateau: DESIGN.md 5)
    std::mutex mu;                                    // one routed call at a time (a context takes one submitting thread)
    ~bbs_pool() {
        for (auto& v : ctx) for (bbs_ctx* c : v) if (c) bbs_ctx_destroy(c);
    }
    static int curve_slot(int curve) { return curve == BBS_CURVE_BLS12_381 ? 0 : (curve == BBS_CURVE_BN254 ? 1 : -1); }
    int ensure(int curve) {
        const int s = curve_slot(curve);
        if (s < 0) return BBS_E_ARG;
        if (!ctx[s].empty()) return BBS_OK;
        std::vector<bbs_ctx*> v(devices.size(), nullptr);
        for (size_t d = 0; d < devices.size(); d++) {
            const int rc = bbs_ctx_create(curve, devices[d], &v[d]);
            if (rc) { for (bbs_ctx* c : v) if (c) bbs_ctx_destroy(c); return rc; }
        }
        ctx[s] = v;
        return BBS_OK;
    }
    // fn(ctx) on every member's context of the curve; the first failure is reported (the others are still attempted, so
    // that the members do not drift apart more than they must)
    template <class F>
    int each(int curve, F fn) {
        std::lock_guard<std::mutex> g(mu);
        int rc = ensure(curve);
        if (rc) return rc;
        for (bbs_ctx* c : ctx[curve_slot(curve)]) { const int r = fn(c); if (r && !rc) rc = r; }
        return rc;
    }
};

namespace pool_detail {
// contiguous share of member d of D over n items: [lo, hi) with ceil(n / D) items per member (sharding.shard_plan)
inline void share_of(size_t n, size_t D, size_t d, size_t& lo, size_t& hi) {
    const size_t per = D ? (n + D - 1) / D : n;
    lo = std::min(n, d * per);
    hi = std::min(n, lo + per);
}
struct Piece {                 // one job: items [lo, hi) of list `li`
    size_t li, lo, hi;
};
// the jobs of one member, curves alternating: round robin over the lists' piece queues
inline std::vector<Piece> pieces_of_member(const bbs_pv_list* lists, size_t n_lists, size_t D, size_t d, size_t max_batch) {
    std::vector<std::vector<Piece>> per_list(n_lists);
    for (size_t li = 0; li < n_lists; li++) {
        size_t lo, hi;
        share_of(lists[li].n, D, d, lo, hi);
        for (size_t a = lo; a < hi; a += max_batch) per_list[li].push_back(Piece{li, a, std::min(hi, a + max_batch)});
    }
    std::vector<Piece> out;
    for (size_t k = 0;; k++) {
        bool any = false;
        for (size_t li = 0; li < n_lists; li++) if (k < per_list[li].size()) { out.push_back(per_list[li][k]); any = true; }
        if (!any) break;
    }
    return out;
}
// one member's submitting loop: at most `inflight` jobs outstanding, retired in completion order; statuses are delivered by
// bbs_job_wait into `dst` (the caller's array, or this call's staging array when the list has a global index)
inline int run_member(bbs_pool* p, size_t d, const bbs_pv_list* lists, size_t n_lists, const std::vector<int8_t*>& dst, size_t max_batch) {
    const std::vector<Piece> todo = pieces_of_member(lists, n_lists, p->devices.size(), d, max_batch);
    std::vector<bbs_job*> live;
    int rc = BBS_OK;
    auto retire_one = [&]() {
        size_t k = 0;
        int r = bbs_jobs_wait_any(live.data(), live.size(), &k);
        if (r == BBS_E_STATE && !live.empty()) { k = 0; r = bbs_job_wait(live[0]); }     // (cannot happen: every live job was run)
        if (r && !rc) rc = r;
        if (k < live.size()) { bbs_job_free(live[k]); live.erase(live.begin() + (long)k); }
    };
    for (const Piece& pc : todo) {
        if (rc) break;
        while (live.size() >= (size_t)std::max(1, p->inflight)) retire_one();
        const bbs_pv_list& L = lists[pc.li];
        const size_t rec = 6 * bbs_fp_bytes(L.curve) + 128;
        bbs_job* job = nullptr;
        // sub-ranges of ragged sections are plain pointer arithmetic: item k's elements are data[off[k] .. off[k + 1]), and the
        // staging code rebases the offsets it is given (runtime.hpp RaggedIn)
        const int r = bbs_core_proof_verify_submit(
            p->ctx[bbs_pool::curve_slot(L.curve)][d], pc.hi - pc.lo, L.proofs_fixed + pc.lo * rec,
            L.commitments, L.commit_off ? L.commit_off + pc.lo : nullptr, L.disclosed_msgs, L.dmsg_off ? L.dmsg_off + pc.lo : nullptr,
            L.disclosed_idx, L.didx_off ? L.didx_off + pc.lo : nullptr, L.headers, L.hdr_off ? L.hdr_off + pc.lo : nullptr,
            L.ph, L.ph_off ? L.ph_off + pc.lo : nullptr, dst[pc.li] + pc.lo, &job);
        if (r) { if (!rc) rc = r; break; }
        live.push_back(job);
    }
    while (!live.empty()) retire_one();
    return rc;
}
}  // namespace pool_detail

inline int pool_proof_verify(bbs_pool* p, const bbs_pv_list* lists, size_t n_lists, size_t max_batch) {
    if (!p || (n_lists && !lists)) return BBS_E_ARG;
    if (max_batch == 0) max_batch = 4096;
    std::lock_guard<std::mutex> g(p->mu);
    // arguments first: nothing is submitted unless the whole call is well-formed
    for (size_t li = 0; li < n_lists; li++) {
        const bbs_pv_list& L = lists[li];
        const int s = bbs_pool::curve_slot(L.curve);
        if (s < 0 || (L.n && (!L.status || !L.proofs_fixed || !L.commit_off || !L.dmsg_off || !L.didx_off))) return BBS_E_ARG;
        if (L.n && p->ctx[s].empty()) return BBS_E_STATE;          // no generators / key for this curve yet
    }
    // lists with a global index deliver into a staging array and are scattered at the end
    std::vector<std::vector<int8_t>> staged(n_lists);
    std::vector<int8_t*> dst(n_lists, nullptr);
    for (size_t li = 0; li < n_lists; li++) {
        if (lists[li].global_index) { staged[li].assign(lists[li].n, (int8_t)0); dst[li] = staged[li].data(); }
        else dst[li] = lists[li].status;
    }
    const size_t D = p->devices.size();
    std::vector<int> rcs(D, BBS_OK);
    if (D == 1) rcs[0] = pool_detail::run_member(p, 0, lists, n_lists, dst, max_batch);
    else {
        std::vector<std::thread> th;
        for (size_t d = 0; d < D; d++) th.emplace_back([&, d]() { rcs[d] = pool_detail::run_member(p, d, lists, n_lists, dst, max_batch); });
        for (auto& t : th) t.join();
    }
    int rc = BBS_OK;
    for (int r : rcs) if (r && !rc) rc = r;
    if (rc) return rc;                                              // (statuses of the jobs that did deliver are in place; the call failed)
    for (size_t li = 0; li < n_lists; li++)
        if (lists[li].global_index)
            for (size_t k = 0; k < lists[li].n; k++) lists[li].status[lists[li].global_index[k]] = staged[li][k];
    return BBS_OK;
}
