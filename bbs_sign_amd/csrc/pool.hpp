// core_proof_verify over a LIST of proofs, fanned out over SEVERAL GPUs behind the C ABI ("multi-GPU
// fan-out is internal"; 8(e): shard by curve, then contiguous ranges per GPU, tables replicated, no data-path collective).
//
// The reference verifies one proof per call (src/proof_verify.rs:19-61, core form :64-116); a caller with a list loops.
// A bbs_pool owns one context per (curve, member device) -- the same generators and issuer key on every member -- and one
// SUBMITTING THREAD per member.  bbs_pool_proof_verify_submit takes the list as one section per curve in the layout of
// bbs_core_proof_verify_batch, cuts every section into contiguous shares (ceil(n / members) items per member:
// sharding.shard_plan's rule), every share into jobs of at most `max_batch` items, and queues them to the members' threads,
// which run them through bbs_core_proof_verify_submit -- at most `inflight` jobs outstanding per member, retired in completion
// order, curves alternating so that BN254 and BLS12-381 jobs overlap on the chip.  A member does NOT drain between lists:
// the jobs of the next list go in while the last jobs of this one finish (at 8 GPUs a member owns two jobs per list; run
// list by list the device idles through every ramp and drain -- measured 7.3 against 4.9 ms per list).
// The statuses land straight in the caller's array, in the caller's order: one process drives every GPU, so the exchange
// that the one-process-per-GPU launcher (bbs_sign_amd/mixed.py, RCCL all_gather) needs does not exist here.
//
// Plain host code on top of the C ABI of include/bbs_sign_amd.h (no field arithmetic, no HIP calls here).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/bbs_sign_amd.h"

struct bbs_pool;
namespace pool_detail {
// contiguous share of member d of D over n items: [lo, hi) with ceil(n / D) items per member (sharding.shard_plan)
inline void share_of(size_t n, size_t D, size_t d, size_t& lo, size_t& hi) {
    const size_t per = D ? (n + D - 1) / D : n;
    lo = std::min(n, d * per);
    hi = std::min(n, lo + per);
}
struct Piece { size_t li, lo, hi; };          // one job: items [lo, hi) of section `li`
// a ragged part of a section as the staging code will judge it for every job cut from it (runtime.hpp RaggedIn::measure):
// offsets never decrease, the total is not garbage, and there is data behind a non-empty total.  Checked for the whole
// section BEFORE anything is queued, so that a malformed list is refused by the submitting call and not by a member's
// thread after other jobs of the list have run.
inline bool ragged_ok(const uint64_t* off, const void* data, size_t n) {
    if (!off) return true;
    for (size_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return false;
    const uint64_t total = off[n] - off[0];
    if (total > ((uint64_t)1 << 36)) return false;
    return total == 0 || data != nullptr;
}
}  // namespace pool_detail

// a list in flight on the pool: the sections as the caller gave them (the caller's buffers stay valid until wait), where
// the statuses go, and the bookkeeping of the members' threads
struct bbs_pool_job {
    bbs_pool* pool = nullptr;
    std::vector<bbs_pv_list> lists;
    std::vector<std::vector<int8_t>> staged;  // sections with a global index deliver here and are scattered by wait
    std::vector<int8_t*> dst;
    size_t max_batch = 4096;
    std::mutex mu;
    std::condition_variable cv;
    size_t members_left = 0;                  // members that still have pieces of this list to submit or to retire
    int rc = BBS_OK;                          // first failure of any member
    bool scattered = false;
    void member_done(int r) {
        std::lock_guard<std::mutex> g(mu);
        if (r && !rc) rc = r;
        if (members_left) members_left--;
        cv.notify_all();
    }
};

struct bbs_pool {
    std::vector<int> devices;                         // device ids, one entry per member (an id may repeat: two context sets on one GPU)
    std::vector<bbs_ctx*> ctx[2];                     // ctx[curve][member]; created at the first configuration call that names the curve
    std::atomic<int> inflight{6};                     // jobs outstanding per member (the serving loop's plateau: DESIGN.md 5)
    std::mutex mu;                                    // configuration and submission are serialised
    // ---- one submitting thread per member
    struct Member {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<bbs_pool_job*> todo;
        bool stop = false;
    };
    std::vector<std::unique_ptr<Member>> members;
    std::atomic<int> lists_in_flight{0};

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
    // that the members do not drift apart more than they must).  Refused while a list is in flight: its jobs read the
    // contexts' tables and keys.
    template <class F>
    int each(int curve, F fn) {
        std::lock_guard<std::mutex> g(mu);
        if (lists_in_flight.load() > 0) return BBS_E_STATE;
        int rc = ensure(curve);
        if (rc) return rc;
        for (bbs_ctx* c : ctx[curve_slot(curve)]) { const int r = fn(c); if (r && !rc) rc = r; }
        return rc;
    }

    // the jobs of member d for one list, curves alternating: round robin over the sections' piece queues
    std::vector<pool_detail::Piece> pieces_of(const bbs_pool_job& j, size_t d) const {
        using pool_detail::Piece;
        const size_t n_lists = j.lists.size(), D = devices.size();
        std::vector<std::vector<Piece>> per_list(n_lists);
        for (size_t li = 0; li < n_lists; li++) {
            size_t lo, hi;
            pool_detail::share_of(j.lists[li].n, D, d, lo, hi);
            for (size_t a = lo; a < hi; a += j.max_batch) per_list[li].push_back(Piece{li, a, std::min(hi, a + j.max_batch)});
        }
        std::vector<Piece> out;
        for (size_t k = 0;; k++) {
            bool any = false;
            for (size_t li = 0; li < n_lists; li++) if (k < per_list[li].size()) { out.push_back(per_list[li][k]); any = true; }
            if (!any) break;
        }
        return out;
    }

    // Member d's thread.  `live` persists across lists: the next list's jobs are submitted while the last jobs of the
    // previous one are still running; a list is finished for this member when all of ITS jobs have been retired.
    void member_loop(size_t d) {
        Member& me = *members[d];
        struct Live { bbs_job* job; bbs_pool_job* owner; };
        std::vector<Live> live;
        struct Open { bbs_pool_job* owner; size_t outstanding; bool all_submitted; int rc; };
        std::vector<Open> open;                       // lists this member still has jobs of
        auto find = [&](bbs_pool_job* o) -> Open* { for (auto& x : open) if (x.owner == o) return &x; return nullptr; };
        auto close_finished = [&]() {
            for (size_t k = 0; k < open.size();) {
                if (open[k].all_submitted && open[k].outstanding == 0) { open[k].owner->member_done(open[k].rc); open.erase(open.begin() + (long)k); }
                else k++;
            }
        };
        auto retire_one = [&]() {
            std::vector<bbs_job*> js;
            for (auto& l : live) js.push_back(l.job);
            size_t k = 0;
            int r = bbs_jobs_wait_any(js.data(), js.size(), &k);
            if (r == BBS_E_STATE && k >= live.size()) { k = 0; r = bbs_job_wait(live[0].job); }     // (cannot happen: every live job was run)
            if (k >= live.size()) k = 0;
            Open* o = find(live[k].owner);
            if (o) { if (r && !o->rc) o->rc = r; if (o->outstanding) o->outstanding--; }
            bbs_job_free(live[k].job);
            live.erase(live.begin() + (long)k);
            close_finished();
        };
        for (;;) {
            bbs_pool_job* next = nullptr;
            {
                std::unique_lock<std::mutex> lk(me.mu);
                if (me.todo.empty() && live.empty()) me.cv.wait(lk, [&]() { return me.stop || !me.todo.empty(); });
                if (!me.todo.empty()) { next = me.todo.front(); me.todo.pop_front(); }
                else if (me.stop && live.empty()) return;
            }
            if (!next) { if (!live.empty()) retire_one(); continue; }      // nothing new: drain
            open.push_back(Open{next, 0, false, BBS_OK});
            for (const pool_detail::Piece& pc : pieces_of(*next, d)) {
                Open* o = find(next);
                if (o->rc) break;                                          // this member failed on this list: submit no more of it
                while (live.size() >= (size_t)std::max(1, inflight.load())) retire_one();
                const bbs_pv_list& L = next->lists[pc.li];
                const size_t rec = 6 * bbs_fp_bytes(L.curve) + 128;
                bbs_job* job = nullptr;
                // sub-ranges of ragged sections are plain pointer arithmetic: item k's elements are data[off[k] .. off[k + 1]),
                // and the staging code rebases the offsets it is given (runtime.hpp RaggedIn)
                const int r = bbs_core_proof_verify_submit(
                    ctx[curve_slot(L.curve)][d], pc.hi - pc.lo, L.proofs_fixed + pc.lo * rec,
                    L.commitments, L.commit_off ? L.commit_off + pc.lo : nullptr, L.disclosed_msgs, L.dmsg_off ? L.dmsg_off + pc.lo : nullptr,
                    L.disclosed_idx, L.didx_off ? L.didx_off + pc.lo : nullptr, L.headers, L.hdr_off ? L.hdr_off + pc.lo : nullptr,
                    L.ph, L.ph_off ? L.ph_off + pc.lo : nullptr, next->dst[pc.li] + pc.lo, &job);
                o = find(next);
                if (r) { if (!o->rc) o->rc = r; break; }
                o->outstanding++;
                live.push_back(Live{job, next});
            }
            if (Open* o = find(next)) o->all_submitted = true;
            close_finished();
        }
    }
    void start_members() {
        if (!members.empty()) return;
        for (size_t d = 0; d < devices.size(); d++) members.emplace_back(new Member());
        for (size_t d = 0; d < devices.size(); d++) members[d]->th = std::thread([this, d]() { member_loop(d); });
    }
    ~bbs_pool() {
        for (auto& m : members) { { std::lock_guard<std::mutex> g(m->mu); m->stop = true; } m->cv.notify_all(); }
        for (auto& m : members) if (m->th.joinable()) m->th.join();
        for (auto& v : ctx) for (bbs_ctx* c : v) if (c) bbs_ctx_destroy(c);
    }
};

inline int pool_proof_verify_submit(bbs_pool* p, const bbs_pv_list* lists, size_t n_lists, size_t max_batch, bbs_pool_job** out) {
    if (!p || !out || (n_lists && !lists)) return BBS_E_ARG;
    if (max_batch == 0) max_batch = 4096;
    std::lock_guard<std::mutex> g(p->mu);
    // arguments first: nothing is queued unless the whole call is well-formed
    for (size_t li = 0; li < n_lists; li++) {
        const bbs_pv_list& L = lists[li];
        const int s = bbs_pool::curve_slot(L.curve);
        if (s < 0 || (L.n && (!L.status || !L.proofs_fixed || !L.commit_off || !L.dmsg_off || !L.didx_off))) return BBS_E_ARG;
        if (L.n && !(pool_detail::ragged_ok(L.commit_off, L.commitments, L.n) && pool_detail::ragged_ok(L.dmsg_off, L.disclosed_msgs, L.n) &&
                     pool_detail::ragged_ok(L.didx_off, L.disclosed_idx, L.n) && pool_detail::ragged_ok(L.hdr_off, L.headers, L.n) &&
                     pool_detail::ragged_ok(L.ph_off, L.ph, L.n))) return BBS_E_ARG;
        if (L.n && p->ctx[s].empty()) return BBS_E_STATE;          // no generators / key for this curve yet
    }
    auto job = std::unique_ptr<bbs_pool_job>(new bbs_pool_job());
    job->pool = p;
    job->lists.assign(lists, lists + n_lists);
    job->max_batch = max_batch;
    job->staged.resize(n_lists);
    job->dst.assign(n_lists, nullptr);
    for (size_t li = 0; li < n_lists; li++) {
        if (lists[li].global_index) { job->staged[li].assign(lists[li].n, (int8_t)0); job->dst[li] = job->staged[li].data(); }
        else job->dst[li] = lists[li].status;
    }
    p->start_members();
    job->members_left = p->members.size();
    p->lists_in_flight.fetch_add(1);
    for (auto& m : p->members) {
        { std::lock_guard<std::mutex> g2(m->mu); m->todo.push_back(job.get()); }
        m->cv.notify_all();
    }
    *out = job.release();
    return BBS_OK;
}
// blocks until every member has retired its jobs of this list; scatters the sections that carry a global index
inline int pool_job_wait(bbs_pool_job* j) {
    if (!j) return BBS_E_ARG;
    // under the job's lock until the statuses are where the caller reads them: a second thread waiting on the same job returns
    // only after the first has finished scattering (the members are done with the job by then, nobody else wants the lock)
    std::unique_lock<std::mutex> lk(j->mu);
    j->cv.wait(lk, [&]() { return j->members_left == 0; });
    if (j->scattered) return j->rc;
    j->scattered = true;
    j->pool->lists_in_flight.fetch_sub(1);
    if (j->rc) return j->rc;                                        // (statuses of the jobs that did deliver are in place; the list failed)
    for (size_t li = 0; li < j->lists.size(); li++)
        if (j->lists[li].global_index)
            for (size_t k = 0; k < j->lists[li].n; k++) j->lists[li].status[j->lists[li].global_index[k]] = j->staged[li][k];
    return BBS_OK;
}
inline void pool_job_free(bbs_pool_job* j) {
    if (!j) return;
    (void)pool_job_wait(j);                                         // the members hold pointers into the job until they are done with it
    delete j;
}
