// An issuer whose signatures and proofs carry ANY number of messages.
//
// The reference's public functions choose their generators by the item's own length on every call:
// create_generators(messages.len() + 1) in sign / verify / proof_gen (src/sign.rs:44-49, src/verify.rs:30-35,
// src/proof_gen.rs:91-96) and create_generators(proof.commitments.len() + disclosed_indexes.len() + 1) in proof_verify
// (src/proof_verify.rs:40-43).  A bbs_ctx holds the device-resident tables of ONE generator set, so a batch whose items
// differ in length needs several.  bbs_issuer keeps one context per message count it has seen -- created on first use:
// hash-to-curve of the generators on the host, window tables and line tables on the device -- and ROUTES the items of a
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

struct bbs_issuer {
    int curve = 0, device = 0;
    std::vector<uint8_t> api_id;
    size_t max_messages = 1024;
    int window_bits = 0;                 // 0: by free device memory (bbs_ctx_set_window_bits(ctx, 0))
    bool pk_set = false, sk_set = false;
    std::vector<uint8_t> pk;             // affine record
    int pk_inf = 0;
    uint8_t sk[32] = {0};
    int latency_mode = 2, batch_verify = 0, in_subgroup = 0;
    std::map<size_t, bbs_ctx*> by_count; // message count -> context
    std::mutex mu;
    ~bbs_issuer() {
        for (auto& kv : by_count) bbs_ctx_destroy(kv.second);
        volatile uint8_t* s = sk;
        for (int k = 0; k < 32; k++) s[k] = 0;
    }
    // the context of message count L (created and set up on first use)
    int context(size_t L, bbs_ctx** out) {
        std::lock_guard<std::mutex> g(mu);
        auto it = by_count.find(L);
        if (it != by_count.end()) { *out = it->second; return BBS_OK; }
        if (L > max_messages) return BBS_E_ARG;
        bbs_ctx* c = nullptr;
        int rc = bbs_ctx_create(curve, device, &c);
        if (rc) return rc;
        const size_t fpb = bbs_fp_bytes(curve);
        std::vector<uint8_t> gens((L + 1) * 2 * fpb);
        rc = bbs_create_generators(curve, L + 1, api_id.data(), api_id.size(), gens.data());
        if (!rc) rc = bbs_ctx_set_window_bits(c, window_bits);
        if (!rc) rc = bbs_ctx_set_generators(c, gens.data(), L + 1, api_id.data(), api_id.size());
        if (!rc && sk_set) rc = bbs_ctx_set_secret_key(c, sk);
        else if (!rc && pk_set) rc = bbs_ctx_set_public_key(c, pk.data(), pk_inf);
        if (!rc) rc = bbs_ctx_set_latency_mode(c, latency_mode);
        if (!rc && batch_verify) rc = bbs_ctx_set_batch_verification(c, 1, nullptr);
        if (!rc && in_subgroup) rc = bbs_ctx_set_points_in_subgroup(c, 1);
        if (rc) { bbs_ctx_destroy(c); return rc; }
        by_count[L] = c;
        *out = c;
        return BBS_OK;
    }
};

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
};
}  // namespace issuer_detail

// a routed call in flight: the groups (packed inputs, their jobs, their result buffers) and how to scatter the results
// into the caller's buffers once every group has been waited for
struct bbs_issuer_job {
    std::map<size_t, issuer_detail::Group> groups;
    std::function<void(std::map<size_t, issuer_detail::Group>&)> scatter;
    bool delivered = false;
};

namespace issuer_detail {
inline void free_jobs(std::map<size_t, Group>& groups) {
    for (auto& kv : groups) if (kv.second.job) { bbs_job_free(kv.second.job); kv.second.job = nullptr; }
}
inline int wait_all(std::map<size_t, Group>& groups) {
    int rc = BBS_OK;
    for (auto& kv : groups) {
        if (!kv.second.job) continue;
        const int r = bbs_job_wait(kv.second.job);
        if (r && !rc) rc = r;
    }
    free_jobs(groups);
    return rc;
}

}  // namespace issuer_detail
