// An issuer whose signatures and proofs carry ANY number of messages.
//
// The reference's public functions choose their generators by the item's own length on every call:
// create_generators(messages.len() + 1) in sign / verify / proof_gen (src/sign.rs:44-49, src/verify.rs:30-35,
// src/proof_gen.rs:91-96) and create_generators(proof.commitments.len() + disclosed_indexes.len() + 1) in proof_verify
// (src/proof_verify.rs:40-43).  A bbs_ctx holds the device-resident tables of ONE generator set, so a batch whose items
// differ in length needs several.  bbs_issuer keeps one context per message count it has seen -- created on first use:
// hash-to-curve of the generators on the host, window tables and line tables on the device; at most max_contexts of them
// and max_table_bytes of tables stay resident, idle ones leave least-recently-used first -- and ROUTES the items of a
// call: items are grouped by their message count, every group goes through the context's one-call wire form
// (bbs_*_wire_submit: octet strings and raw messages, everything else on the device), all groups are in flight
// together, and the statuses / outputs are scattered back into the caller's order.
//
// Plain host code on top of the C ABI of include/bbs_sign_amd.h (no field arithmetic here).
#pragma once
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/bbs_sign_amd.h"

// one context of the issuer: the generator set of one message count
struct bbs_issuer_entry {
    size_t L = 0;
    bbs_ctx* ctx = nullptr;              // built on first use, under `mu`
    std::mutex mu;                       // serialises everything done to / submitted on this context (the bbs_ctx contract:
                                         // one call at a time per context, one submitting thread for the asynchronous forms)
    uint64_t config_epoch = 0;           // the issuer configuration (keys, modes) this context was last brought up to
    size_t table_bytes = 0;
    // guarded by the issuer's mutex:
    int pins = 0;                        // routed lists in flight on this context
    bool handed_out = false;             // bbs_issuer_context gave the raw context to the caller: never evicted
    uint64_t last_use = 0;
    bool dead = false;                   // evicted: a thread that still holds the entry must look it up again
    ~bbs_issuer_entry() { if (ctx) bbs_ctx_destroy(ctx); }
};

struct bbs_issuer {
    int curve = 0, device = 0;
    std::vector<uint8_t> api_id;
    // ---- configuration, guarded by `mu`; a context picks it up (by epoch) the next time it is used
    size_t max_messages = 1024;
    int window_bits = 0;                 // 0: by free device memory (bbs_ctx_set_window_bits(ctx, 0))
    bool pk_set = false, sk_set = false;
    std::vector<uint8_t> pk;             // affine record
    int pk_inf = 0;
    uint8_t sk[32] = {0};
    int latency_mode = 2, batch_verify = 0, in_subgroup = 0;
    uint64_t epoch = 1;                  // bumped by every configuration change
    // ---- resident contexts: bounded in number and in table bytes; idle ones leave least-recently-used first.  The message
    // count of a proof comes from the length of an untrusted octet string: without the bound a caller sending one proof of
    // every length up to max_messages would make the device hold that many table sets.
    size_t max_contexts = 64;
    size_t max_table_bytes = 0;          // 0: half of the device memory free when the first context is created
    size_t table_bytes = 0;
    uint64_t tick = 0;
    std::map<size_t, std::shared_ptr<bbs_issuer_entry>> by_count;   // message count -> context
    std::mutex mu;
    ~bbs_issuer() {
        by_count.clear();
        volatile uint8_t* s = sk;
        for (int k = 0; k < 32; k++) s[k] = 0;
    }
    struct Config {
        bool pk_set, sk_set; std::vector<uint8_t> pk; int pk_inf; uint8_t sk[32]; int latency_mode, batch_verify, in_subgroup, window_bits;
        uint64_t epoch;
        ~Config() { volatile uint8_t* s = sk; for (int k = 0; k < 32; k++) s[k] = 0; }
    };
    void snapshot(Config& c) {           // `mu` held
        c.pk_set = pk_set; c.sk_set = sk_set; c.pk = pk; c.pk_inf = pk_inf; std::memcpy(c.sk, sk, 32);
        c.latency_mode = latency_mode; c.batch_verify = batch_verify; c.in_subgroup = in_subgroup; c.window_bits = window_bits; c.epoch = epoch;
    }
    bool configuration_locked() {        // `mu` held: a routed list is in flight
        for (auto& kv : by_count) if (kv.second->pins > 0) return true;
        return false;
    }
    // idle contexts, least recently used first, until `need_slots` more fit and the byte budget holds; the victims are
    // destroyed by the caller AFTER releasing `mu` (bbs_ctx_destroy synchronises the context's stream)
    void evict(size_t need_slots, std::vector<std::shared_ptr<bbs_issuer_entry>>& victims, const bbs_issuer_entry* keep) {
        auto over = [&]() { return by_count.size() + need_slots > max_contexts || (max_table_bytes && table_bytes > max_table_bytes); };
        while (over()) {
            std::map<size_t, std::shared_ptr<bbs_issuer_entry>>::iterator lru = by_count.end();
            for (auto it = by_count.begin(); it != by_count.end(); ++it) {
                if (it->second->pins > 0 || it->second->handed_out || it->second.get() == keep) continue;
                if (lru == by_count.end() || it->second->last_use < lru->second->last_use) lru = it;
            }
            if (lru == by_count.end()) return;               // everything left is in use
            lru->second->dead = true;
            table_bytes -= std::min(table_bytes, lru->second->table_bytes);
            victims.push_back(lru->second);
            by_count.erase(lru);
        }
    }
    // Pins the entry of message count L (created empty if new; BBS_E_NOMEM if the context limit is reached and nothing is
    // idle) and returns the configuration to bring it up to.
    int acquire(size_t L, std::shared_ptr<bbs_issuer_entry>& out, Config& cfg, bool forever = false, bool* newly_handed_out = nullptr) {
        std::vector<std::shared_ptr<bbs_issuer_entry>> victims;
        int rc = BBS_OK;
        {
            std::lock_guard<std::mutex> g(mu);
            if (L > max_messages) return BBS_E_ARG;
            auto it = by_count.find(L);
            if (it == by_count.end()) {
                evict(1, victims, nullptr);
                if (by_count.size() + 1 > max_contexts) rc = BBS_E_NOMEM;
                else {
                    auto e = std::make_shared<bbs_issuer_entry>();
                    e->L = L;
                    it = by_count.emplace(L, e).first;
                }
            }
            if (!rc) {
                out = it->second;
                out->pins += 1;
                if (forever) { if (newly_handed_out) *newly_handed_out = !out->handed_out; out->handed_out = true; }
                out->last_use = ++tick;
                snapshot(cfg);
            }
        }
        victims.clear();                                     // destroyed here, outside the lock
        return rc;
    }
    void release(const std::shared_ptr<bbs_issuer_entry>& e) {
        if (!e) return;
        std::lock_guard<std::mutex> g(mu);
        if (e->pins > 0) e->pins--;
    }
    // `e->mu` held: build the context if this is its first use, bring it up to the configuration
    int prepare(bbs_issuer_entry* e, const Config& cfg) {
        int rc = BBS_OK;
        const bool fresh = e->ctx == nullptr;
        if (fresh) {
            bbs_ctx* c = nullptr;
            if ((rc = bbs_ctx_create(curve, device, &c))) return rc;
            {                                                // first context of the issuer: the byte budget
                const size_t fr = bbs_device_free_bytes(device);
                std::lock_guard<std::mutex> g(mu);           // (bbs_issuer_set_budget writes it under the same lock)
                if (max_table_bytes == 0) max_table_bytes = std::max<size_t>(fr / 2, (size_t)64 << 20);
            }
            const size_t fpb = bbs_fp_bytes(curve);
            std::vector<uint8_t> gens((e->L + 1) * 2 * fpb);
            rc = bbs_create_generators(curve, e->L + 1, api_id.data(), api_id.size(), gens.data());
            if (!rc) rc = bbs_ctx_set_window_bits(c, cfg.window_bits);
            if (!rc) rc = bbs_ctx_set_generators(c, gens.data(), e->L + 1, api_id.data(), api_id.size());
            if (rc) { bbs_ctx_destroy(c); return rc; }
            e->ctx = c;
            e->config_epoch = 0;
        }
        if (e->config_epoch != cfg.epoch) {
            bbs_ctx* c = e->ctx;
            if (cfg.sk_set) rc = bbs_ctx_set_secret_key(c, cfg.sk);
            else if (cfg.pk_set) rc = bbs_ctx_set_public_key(c, cfg.pk.data(), cfg.pk_inf);
            if (!rc) rc = bbs_ctx_set_latency_mode(c, cfg.latency_mode);
            if (!rc) rc = bbs_ctx_set_batch_verification(c, cfg.batch_verify, nullptr);
            if (!rc) rc = bbs_ctx_set_points_in_subgroup(c, cfg.in_subgroup);
            if (rc) return rc;
            e->config_epoch = cfg.epoch;
        }
        if (fresh) {
            // account for the new tables (measured now that the keys are in: everything the context holds on the device);
            // make room among the idle contexts if the budget is exceeded
            std::vector<std::shared_ptr<bbs_issuer_entry>> victims;
            e->table_bytes = bbs_ctx_table_bytes(e->ctx);
            {
                std::lock_guard<std::mutex> g(mu);
                table_bytes += e->table_bytes;
                evict(0, victims, e);
            }
            victims.clear();
        }
        return BBS_OK;
    }
    // Runs fn(ctx) on the context of message count L with that context locked (one call at a time per context).  On success
    // the entry stays pinned and is handed to the caller in `pinned` (released when the routed list has been waited for);
    // on failure nothing stays pinned.
    int with_context(size_t L, std::shared_ptr<bbs_issuer_entry>& pinned, const std::function<int(bbs_ctx*)>& fn) {
        for (int attempt = 0; attempt < 4; attempt++) {
            std::shared_ptr<bbs_issuer_entry> e;
            Config cfg;
            int rc = acquire(L, e, cfg);
            if (rc) return rc;
            bool built = false;
            {
                std::lock_guard<std::mutex> g(e->mu);
                bool dead;
                { std::lock_guard<std::mutex> g2(mu); dead = e->dead; }
                if (dead) { release(e); continue; }          // evicted between the look-up and the lock (it was idle then)
                rc = prepare(e.get(), cfg);
                if (!rc) rc = fn(e->ctx);
                built = e->ctx != nullptr;                   // read while the entry is locked (another pinned thread may be building it)
            }
            if (rc) {
                release(e);
                if (!built) {                                // never built: do not keep the empty slot
                    std::lock_guard<std::mutex> g(mu);
                    auto it = by_count.find(L);
                    if (it != by_count.end() && it->second == e && e->pins == 0) { e->dead = true; by_count.erase(it); }
                }
                return rc;
            }
            pinned = e;
            return BBS_OK;
        }
        return BBS_E_STATE;
    }
    // bbs_issuer_context: the raw context, for warm-up and for callers that drive it themselves; it is never evicted
    int context(size_t L, bbs_ctx** out) {
        std::shared_ptr<bbs_issuer_entry> e;
        Config cfg;
        bool newly = false;
        int rc = acquire(L, e, cfg, true, &newly);
        if (rc) return rc;
        std::lock_guard<std::mutex> g(e->mu);
        rc = prepare(e.get(), cfg);
        {
            std::lock_guard<std::mutex> g2(mu);
            if (e->pins > 0) e->pins--;                      // (the pin only covered the build; handed_out keeps the context resident)
            if (rc) {
                if (newly) e->handed_out = false;            // an EARLIER caller may hold this context: only this call's own mark is taken back
                if (!e->ctx) { auto it = by_count.find(L); if (it != by_count.end() && it->second == e && e->pins == 0) { e->dead = true; by_count.erase(it); } }
                return rc;
            }
        }
        *out = e->ctx;
        return BBS_OK;
    }
};

// Contexts that bbs_issuer_context handed out are driven by the caller directly, so nothing "uses" them through the issuer
// and the lazy, by-epoch update of prepare() would never reach them: after a key rotation such a context would go on
// verifying (or signing) with the OLD key.  The configuration calls therefore bring them up to the new epoch at once.
// Called WITHOUT the issuer's lock (prepare takes the entry's lock first, then the issuer's).
inline int issuer_refresh_handed_out(bbs_issuer* is) {
    std::vector<std::shared_ptr<bbs_issuer_entry>> held;
    {
        std::lock_guard<std::mutex> g(is->mu);
        for (auto& kv : is->by_count) if (kv.second->handed_out) held.push_back(kv.second);
    }
    int rc = BBS_OK;
    for (auto& e : held) {
        std::lock_guard<std::mutex> g(e->mu);
        bbs_issuer::Config cfg;
        { std::lock_guard<std::mutex> g2(is->mu); if (e->dead) continue; is->snapshot(cfg); }
        if (!e->ctx) continue;
        const int r = is->prepare(e.get(), cfg);
        if (r && !rc) rc = r;
    }
    return rc;
}

namespace issuer_detail {

// a ragged section of the caller's batch: item i owns [off[i], off[i + 1]) elements of `elem` bytes (off = NULL: all empty)
struct Ragged {
    const uint8_t* data; const uint64_t* off; size_t elem;
    uint64_t count(size_t i) const { return off ? off[i + 1] - off[i] : 0; }
    bool sane(size_t n) const {
        if (!off) return true;
        for (size_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return false;
        const uint64_t total = off[n] - off[0];
        return total <= ((uint64_t)1 << 36) && (total == 0 || data != nullptr);
    }
};
// the same section restricted to a list of items, packed
struct Packed {
    std::vector<uint8_t> data;
    std::vector<uint64_t> off;
    void gather(const Ragged& r, const std::vector<size_t>& items) {
        off.assign(items.size() + 1, 0);
        uint64_t tot = 0;
        for (size_t k = 0; k < items.size(); k++) { tot += r.count(items[k]); off[k + 1] = tot; }
        data.resize((size_t)tot * r.elem + 8);
        for (size_t k = 0; k < items.size(); k++) {
            const uint64_t c = r.count(items[k]);
            if (c) std::memcpy(data.data() + (size_t)off[k] * r.elem, r.data + (size_t)r.off[items[k]] * r.elem, (size_t)c * r.elem);
        }
    }
};
// messages: two levels (item -> messages -> bytes)
struct PackedMsgs {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> byte_off, item_off;
    void gather(const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off, const std::vector<size_t>& items) {
        item_off.assign(items.size() + 1, 0);
        byte_off.assign(1, 0);
        bytes.clear();
        for (size_t k = 0; k < items.size(); k++) {
            const size_t i = items[k];
            for (uint64_t t = msg_item_off[i]; t < msg_item_off[i + 1]; t++) {
                const uint64_t b0 = msg_byte_off[t], b1 = msg_byte_off[t + 1];
                bytes.insert(bytes.end(), msg_bytes + b0, msg_bytes + b1);
                byte_off.push_back(bytes.size());
            }
            item_off[k + 1] = byte_off.size() - 1;
        }
        bytes.resize(bytes.size() + 8);
    }
};
inline bool msgs_sane(size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off) {
    if (!n) return true;
    if (!msg_item_off) return false;
    for (size_t i = 0; i < n; i++) if (msg_item_off[i + 1] < msg_item_off[i]) return false;
    const uint64_t m0 = msg_item_off[0], m1 = msg_item_off[n];
    if (m1 - m0 > ((uint64_t)1 << 32)) return false;
    if (m1 == m0) return true;
    if (!msg_byte_off) return false;
    for (uint64_t t = m0; t < m1; t++) if (msg_byte_off[t + 1] < msg_byte_off[t]) return false;
    const uint64_t bytes = msg_byte_off[m1] - msg_byte_off[m0];
    return bytes <= ((uint64_t)1 << 36) && (bytes == 0 || msg_bytes != nullptr);
}

// one group of a routed call: the items of one message count, their packed inputs, the job and where its results go
struct Group {
    size_t L = 0;
    std::vector<size_t> items;
    Packed oct, di, rnd, hdr, ph;
    PackedMsgs msgs;
    std::vector<int8_t> status;
    std::vector<uint8_t> out;            // produced octet strings
    std::vector<uint64_t> out_off;
    bbs_job* job = nullptr;
    std::shared_ptr<bbs_issuer_entry> entry;      // pinned while the group's job is in flight
};
}  // namespace issuer_detail

// a routed call in flight: the groups (packed inputs, their jobs, their result buffers) and how to scatter the results
// into the caller's buffers once every group has been waited for
struct bbs_issuer_job {
    bbs_issuer* issuer = nullptr;
    std::map<size_t, issuer_detail::Group> groups;
    std::function<void(std::map<size_t, issuer_detail::Group>&)> scatter;
    bool delivered = false;
};

namespace issuer_detail {
// (bbs_job_free waits for the job's streams: the context is idle for this list from then on)
inline void free_jobs(bbs_issuer* is, std::map<size_t, Group>& groups) {
    for (auto& kv : groups) {
        if (kv.second.job) { bbs_job_free(kv.second.job); kv.second.job = nullptr; }
        if (kv.second.entry) { is->release(kv.second.entry); kv.second.entry.reset(); }
    }
}
inline int wait_all(bbs_issuer* is, std::map<size_t, Group>& groups) {
    int rc = BBS_OK;
    for (auto& kv : groups) {
        if (!kv.second.job) continue;
        const int r = bbs_job_wait(kv.second.job);
        if (r && !rc) rc = r;
    }
    free_jobs(is, groups);
    return rc;
}
// One group of a routed call: fn(ctx) submits the group's job on the context of its message count.  A context that cannot
// be set up for lack of device memory (or because every resident context is busy and the limit is reached) fails THIS
// group's items with BBS_ST_NO_RESOURCES -- the other groups of the call are served; any other failure aborts the call.
inline int submit_group(bbs_issuer* is, Group& g, const std::function<int(bbs_ctx*)>& fn) {
    const int rc = is->with_context(g.L, g.entry, fn);
    if (rc == BBS_E_NOMEM) {
        g.status.assign(g.items.size(), (int8_t)BBS_ST_NO_RESOURCES);
        if (g.job) { bbs_job_free(g.job); g.job = nullptr; }
        return BBS_OK;
    }
    return rc;
}

}  // namespace issuer_detail
