/*
 * The call sequence of bindings/rust/src/lib.rs (GpuIssuer::new / set_secret_key / set_public_key / sign / verify /
 * proof_gen / proof_verify_submit / PendingVerify::wait / Drop), argument for argument, from a plain-C client of
 * include/bbs_sign_amd.h -- so that the ABI the Rust shim binds is exercised by something that is neither Python
 * (ctypes) nor C++.  No Rust toolchain exists in this image; the step numbers are the ones quoted in the shim.
 *
 * What the shim's callers see is checked: statuses of valid items, of tampered items (Ok(false)), of the reference's
 * Err variants (src/proof_verify.rs:139-150, src/proof_gen.rs:133-143, src/sign.rs:77-79), the round trip
 * sign -> verify -> proof_gen -> proof_verify, two submitted batches in flight, and that the buffers handed to
 * bbs_core_proof_verify_submit may be reused as soon as it returns.
 *
 * Build (tests/test_ffi_sequence.py): gcc -std=c99 -I include tests/cpp/ffi_sequence.c -l:<library> ; run: ./a.out
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bbs_sign_amd.h"

#define CHECK(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)

enum { L = 3, N = 4 };                 /* messages per item, items per batch */

/* what run_curve leaves for run_pool: the three valid proofs of the curve with everything bbs_core_proof_verify_submit takes */
static struct {
    uint8_t gens[(L + 1) * 2 * 48], pk[4 * 48], pf[3 * (6 * 48 + 128)], cm[N * L * 32], dm[4 * 32], hdr[16], ph[4];
    uint64_t cmo[4], dmo[4], di[4], dio[4], ho[4], po[4];
    char api_id[128];
} saved[2];

static const char* SUITE_ID[2] = {"BBS_BLS12381G1_XMD:SHA-256_SSWU_RO_", "BBS_QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_"};

/* the shim's msg_to_scalars(): every message of every item in one bbs_hash_to_scalar_batch call */
static void msg_to_scalars(bbs_ctx* ctx, const char* api_id, const char* const* msgs, size_t count, uint8_t* out) {
    uint8_t flat[1024];
    uint64_t off[64];
    char dst[160];
    size_t at = 0;
    off[0] = 0;
    for (size_t k = 0; k < count; k++) { memcpy(flat + at, msgs[k], strlen(msgs[k])); at += strlen(msgs[k]); off[k + 1] = at; }
    snprintf(dst, sizeof dst, "%sMAP_MSG_TO_SCALAR_AS_HASH_", api_id);
    CHECK(bbs_hash_to_scalar_batch(ctx, count, flat, off, (const uint8_t*)dst, strlen(dst), out) == BBS_OK);           /* step 10 */
}

static int run_curve(int curve) {
    const size_t fpb = bbs_fp_bytes(curve);                                                                             /* step 1 */
    const size_t sig_rec = 2 * fpb + 32, pf_rec = 6 * fpb + 128;
    char api_id[128];
    snprintf(api_id, sizeof api_id, "%sH2G_HM2S_", SUITE_ID[curve]);
    const size_t alen = strlen(api_id);

    uint8_t* gens = calloc(L + 1, 2 * fpb);
    CHECK(bbs_create_generators(curve, L + 1, (const uint8_t*)api_id, alen, gens) == BBS_OK);                            /* step 2 */
    bbs_ctx* ctx = NULL;
    CHECK(bbs_ctx_create(curve, 0, &ctx) == BBS_OK && ctx);                                                              /* step 3 */
    CHECK(bbs_ctx_set_window_bits(ctx, 5) == BBS_OK);                                                                    /* step 4 */
    CHECK(bbs_ctx_set_generators(ctx, gens, L + 1, (const uint8_t*)api_id, alen) == BBS_OK);                             /* step 5 */
    CHECK(bbs_ctx_set_points_in_subgroup(ctx, 1) == BBS_OK);                                                             /* step 6 */

    uint8_t ikm[32], sk[32];
    memset(ikm, 7, sizeof ikm);
    CHECK(bbs_key_gen(curve, ikm, 32, NULL, 0, (const uint8_t*)"BBS-SIG-KEYGEN-SALT-", 20, sk) == BBS_OK);               /* step 7 */
    CHECK(bbs_ctx_set_secret_key(ctx, sk) == BBS_OK);                                                                    /* step 8 */
    uint8_t pk[4 * 48];
    int pk_inf = -1;
    CHECK(bbs_ctx_get_public_key(ctx, pk, &pk_inf) == BBS_OK && pk_inf == 0);                                            /* step 9 */
    /* a verifier-side context: public key only (GpuIssuer::set_public_key) */
    bbs_ctx* vctx = NULL;
    CHECK(bbs_ctx_create(curve, 0, &vctx) == BBS_OK);
    CHECK(bbs_ctx_set_window_bits(vctx, 5) == BBS_OK);
    CHECK(bbs_ctx_set_generators(vctx, gens, L + 1, (const uint8_t*)api_id, alen) == BBS_OK);
    CHECK(bbs_ctx_set_points_in_subgroup(vctx, 1) == BBS_OK);
    CHECK(bbs_ctx_set_public_key(vctx, pk, 0) == BBS_OK);

    /* ---- SecretKey::sign for N items; the last item has L - 1 messages (SignatureError) -------------------------- */
    const char* raw[N * L] = {"message1", "message2", "msg3", "a", "", "ccc", "x1", "x2", "x3", "only-two", "here", NULL};
    const size_t n_msgs = N * L - 1;
    uint8_t scal[N * L * 32];
    msg_to_scalars(ctx, api_id, raw, n_msgs, scal);
    uint64_t mo[N + 1] = {0, L, 2 * L, 3 * L, 4 * L - 1};
    const uint8_t hdr_bytes[] = "hdr0hdr-two";
    uint64_t ho[N + 1] = {0, 4, 4, 11, 11};                 /* item 1 and 3: empty header */
    uint8_t sigs[N * (2 * 48 + 32)];
    int8_t st[N];
    memset(st, 99, sizeof st);
    CHECK(bbs_core_sign_batch(ctx, N, scal, mo, hdr_bytes, ho, sigs, st) == BBS_OK);                                     /* step 11 */
    CHECK(st[0] == 1 && st[1] == 1 && st[2] == 1 && st[3] == BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH);
    {   /* the same batch through the asynchronous form: records and statuses arrive at bbs_job_wait */
        uint8_t sigs2[N * (2 * 48 + 32)];
        int8_t st2[N];
        bbs_job* sj = NULL;
        memset(sigs2, 0xEE, sizeof sigs2); memset(st2, 99, sizeof st2);
        CHECK(bbs_core_sign_submit(ctx, N, scal, mo, hdr_bytes, ho, sigs2, st2, &sj) == BBS_OK && sj);
        CHECK(bbs_job_wait(sj) == BBS_OK);
        bbs_job_free(sj);
        CHECK(memcmp(st, st2, sizeof st) == 0 && memcmp(sigs, sigs2, N * sig_rec) == 0);
    }

    /* ---- PublicKey::verify: items 0..2; item 1's first message altered -> Ok(false) ------------------------------- */
    uint8_t scal_v[3 * L * 32];
    memcpy(scal_v, scal, sizeof scal_v);
    scal_v[L * 32] ^= 1;
    memset(st, 99, sizeof st);
    CHECK(bbs_core_verify_batch(vctx, 3, sigs, scal_v, mo, hdr_bytes, ho, st) == BBS_OK);                                /* step 12 */
    CHECK(st[0] == 1 && st[1] == 0 && st[2] == 1);

    /* ---- proof_gen: disclosed {0, 2}, {1}, {} ; item 3 (copy of item 0) discloses index L -> InvalidDisclosedIndex - */
    uint8_t sig4[N * (2 * 48 + 32)], scal4[N * L * 32];
    memcpy(sig4, sigs, 3 * sig_rec); memcpy(sig4 + 3 * sig_rec, sigs, sig_rec);
    memcpy(scal4, scal, 3 * L * 32); memcpy(scal4 + 3 * L * 32, scal, L * 32);
    uint64_t mo4[N + 1] = {0, L, 2 * L, 3 * L, 4 * L};
    uint64_t di[] = {0, 2, 1, L};
    uint64_t dio[N + 1] = {0, 2, 3, 3, 4};
    /* random scalars: what calculate_random_scalars does -- 48 random bytes through FromOkm (step 13) */
    uint8_t rnd[N * (5 + L) * 32];
    uint64_t ro[N + 1];
    size_t nr = 0;
    uint32_t lcg = 12345u + (uint32_t)curve;
    ro[0] = 0;
    for (int i = 0; i < N; i++) {
        const size_t r = (size_t)(dio[i + 1] - dio[i]);
        for (size_t k = 0; k < 5 + L - r; k++) {
            uint8_t okm[48];
            for (int b = 0; b < 48; b++) { lcg = lcg * 1664525u + 1013904223u; okm[b] = (uint8_t)(lcg >> 24); }
            CHECK(bbs_scalar_from_okm(curve, okm, rnd + 32 * nr) == BBS_OK);                                             /* step 13 */
            nr++;
        }
        ro[i + 1] = nr;
    }
    const uint8_t ph_bytes[] = "ph";
    uint64_t po[N + 1] = {0, 2, 2, 2, 2};
    uint64_t ho4[N + 1] = {0, 4, 4, 11, 15};
    const uint8_t hdr4[] = "hdr0hdr-twohdr0";
    uint8_t* pf = calloc(N, pf_rec);
    uint8_t cm[N * L * 32];
    uint64_t cmo[N + 1];
    memset(st, 99, sizeof st);
    CHECK(bbs_core_proof_gen_batch(vctx, N, sig4, scal4, mo4, di, dio, rnd, ro, hdr4, ho4, ph_bytes, po, pf, cm, cmo, st) == BBS_OK);   /* step 14 */
    CHECK(st[0] == 1 && st[1] == 1 && st[2] == 1 && st[3] == BBS_ST_INVALID_DISCLOSED_INDEX);
    CHECK(cmo[0] == 0 && cmo[1] == 1 && cmo[2] == 3 && cmo[3] == 6 && cmo[4] == 6);
    {   /* asynchronous form: the same proofs, byte for byte (the random scalars are inputs) */
        uint8_t* pf2 = calloc(N, pf_rec);
        uint8_t cm2[N * L * 32];
        uint64_t cmo2[N + 1];
        int8_t st2[N];
        bbs_job* pj = NULL;
        memset(st2, 99, sizeof st2);
        CHECK(bbs_core_proof_gen_submit(vctx, N, sig4, scal4, mo4, di, dio, rnd, ro, hdr4, ho4, ph_bytes, po, pf2, cm2, cmo2, st2, &pj) == BBS_OK && pj);
        CHECK(bbs_job_wait(pj) == BBS_OK);
        bbs_job_free(pj);
        CHECK(memcmp(st, st2, sizeof st) == 0 && memcmp(cmo, cmo2, sizeof cmo) == 0);
        CHECK(memcmp(pf, pf2, N * pf_rec) == 0 && memcmp(cm, cm2, (size_t)cmo[N] * 32) == 0);
        free(pf2);
    }

    /* ---- proof_verify: two batches submitted before either is waited for ------------------------------------------ */
    /* batch A: items 0..2 as generated.  batch B: item 0 with a commitment altered (Ok(false)), item 1 with an index
     * >= l (InvalidDisclosedIndex), item 2 with one disclosed message too many (InvalidIndicesAndMessagesLength) */
    uint8_t dm[4 * 32];
    memcpy(dm, scal + 0 * 32, 32); memcpy(dm + 32, scal + 2 * 32, 32); memcpy(dm + 64, scal + (L + 1) * 32, 32);
    uint64_t dmo[4] = {0, 2, 3, 3};
    uint64_t dio3[4] = {0, 2, 3, 3};
    int8_t stA[3], stB[3];
    memset(stA, 99, 3); memset(stB, 99, 3);
    bbs_job *jobA = NULL, *jobB = NULL;
    CHECK(bbs_core_proof_verify_submit(vctx, 3, pf, cm, cmo, dm, dmo, di, dio3, hdr4, ho4, ph_bytes, po, stA, &jobA) == BBS_OK && jobA);   /* step 15 */
    uint8_t* cmB = malloc(sizeof cm);
    memcpy(cmB, cm, sizeof cm);
    cmB[0] ^= 1;
    uint64_t diB[] = {0, 2, L + 4};
    memcpy(dm + 96, scal, 32);
    uint64_t dmoB[4] = {0, 2, 3, 4};
    CHECK(bbs_core_proof_verify_submit(vctx, 3, pf, cmB, cmo, dm, dmoB, diB, dio3, hdr4, ho4, ph_bytes, po, stB, &jobB) == BBS_OK && jobB);
    memset(cmB, 0xEE, sizeof cm);                    /* inputs were staged by the call: the caller may reuse its buffers */
    free(cmB);
    CHECK(bbs_job_wait(jobA) == BBS_OK);                                                                                 /* step 16 */
    CHECK(bbs_job_wait(jobB) == BBS_OK);
    bbs_job_free(jobA);                                                                                                  /* step 17 */
    bbs_job_free(jobB);
    CHECK(stA[0] == 1 && stA[1] == 1 && stA[2] == 1);
    CHECK(stB[0] == 0 && stB[1] == BBS_ST_INVALID_DISCLOSED_INDEX && stB[2] == BBS_ST_INVALID_INDICES_AND_MESSAGES_LENGTH);
    /* ---- a serving loop: four batches in flight, retired in COMPLETION order (bbs_jobs_wait_any), every slot refilled
     * once; a retired slot is NULL until it is refilled.  Slots 0 / 2 hold batch A, slots 1 / 3 batch B. -------------- */
    {
        bbs_job* fl[4] = {NULL, NULL, NULL, NULL};
        int8_t stq[4][3];
        int refills = 4, retired = 0;
        uint64_t diB2[] = {0, 2, L + 4};
        uint64_t dmoB2[4] = {0, 2, 3, 4};
        memset(stq, 99, sizeof stq);
        size_t none = 7;
        CHECK(bbs_jobs_wait_any(fl, 4, &none) == BBS_E_STATE);            /* nothing has been run */
        for (int k = 0; k < 4; k++) {
            if (k & 1) CHECK(bbs_core_proof_verify_submit(vctx, 3, pf, cm, cmo, dm, dmoB2, diB2, dio3, hdr4, ho4, ph_bytes, po, stq[k], &fl[k]) == BBS_OK);
            else CHECK(bbs_core_proof_verify_submit(vctx, 3, pf, cm, cmo, dm, dmo, di, dio3, hdr4, ho4, ph_bytes, po, stq[k], &fl[k]) == BBS_OK);
        }
        while (fl[0] || fl[1] || fl[2] || fl[3]) {
            size_t k = 9;
            CHECK(bbs_jobs_wait_any(fl, 4, &k) == BBS_OK && k < 4 && fl[k]);
            CHECK(bbs_job_poll(fl[k]) == 1);
            if (k & 1) CHECK(stq[k][0] == 1 && stq[k][1] == BBS_ST_INVALID_DISCLOSED_INDEX && stq[k][2] == BBS_ST_INVALID_INDICES_AND_MESSAGES_LENGTH);
            else CHECK(stq[k][0] == 1 && stq[k][1] == 1 && stq[k][2] == 1);
            bbs_job_free(fl[k]);
            fl[k] = NULL;
            retired++;
            if (refills > 0) {
                refills--;
                memset(stq[k], 99, 3);
                if (k & 1) CHECK(bbs_core_proof_verify_submit(vctx, 3, pf, cm, cmo, dm, dmoB2, diB2, dio3, hdr4, ho4, ph_bytes, po, stq[k], &fl[k]) == BBS_OK);
                else CHECK(bbs_core_proof_verify_submit(vctx, 3, pf, cm, cmo, dm, dmo, di, dio3, hdr4, ho4, ph_bytes, po, stq[k], &fl[k]) == BBS_OK);
            }
        }
        CHECK(retired == 8);
    }
    /* the synchronous form returns the same */
    int8_t stC[3];
    CHECK(bbs_core_proof_verify_batch(vctx, 3, pf, cm, cmo, dm, dmo, di, dio3, hdr4, ho4, ph_bytes, po, stC) == BBS_OK);
    CHECK(stC[0] == 1 && stC[1] == 1 && stC[2] == 1);
    /* a different presentation header must fail */
    const uint8_t ph2[] = "pX";
    CHECK(bbs_core_proof_verify_batch(vctx, 3, pf, cm, cmo, dm, dmo, di, dio3, hdr4, ho4, ph2, po, stC) == BBS_OK);
    CHECK(stC[0] == 0 && stC[1] == 1 && stC[2] == 1);     /* only item 0 carries a presentation header */

    /* ---- the same round trip through the WIRE forms: raw messages and octet strings only -- what a binding needs when it
     * does not want to convert field elements at all (the reference's public sign / verify / proof_gen / proof_verify) ---- */
    {
        uint8_t mbytes[256];
        uint64_t mbo[3 * L + 1], mio[4] = {0, L, 2 * L, 3 * L};
        size_t at = 0;
        mbo[0] = 0;
        for (int k = 0; k < 3 * L; k++) { memcpy(mbytes + at, raw[k], strlen(raw[k])); at += strlen(raw[k]); mbo[k + 1] = at; }
        const size_t so_len = fpb + 32;
        uint8_t so[3 * (48 + 32)];
        int8_t sw[3];
        CHECK(bbs_sign_wire_batch(ctx, 3, mbytes, mbo, mio, hdr_bytes, ho, so, sw) == BBS_OK);
        CHECK(sw[0] == 1 && sw[1] == 1 && sw[2] == 1);
        for (int i = 0; i < 3; i++) {                    /* the same signatures as the core call above, as octets */
            uint8_t want[48 + 32];
            CHECK(bbs_signature_to_octets(curve, sigs + (size_t)i * sig_rec, want) == BBS_OK);
            CHECK(memcmp(want, so + (size_t)i * so_len, so_len) == 0);
        }
        CHECK(bbs_verify_wire_batch(vctx, 3, so, mbytes, mbo, mio, hdr_bytes, ho, sw) == BBS_OK);
        CHECK(sw[0] == 1 && sw[1] == 1 && sw[2] == 1);
        mbytes[0] ^= 1;                                  /* first message of item 0 altered */
        CHECK(bbs_verify_wire_batch(vctx, 3, so, mbytes, mbo, mio, hdr_bytes, ho, sw) == BBS_OK);
        CHECK(sw[0] == 0 && sw[1] == 1 && sw[2] == 1);
        mbytes[0] ^= 1;
        /* proof_gen: the first three items of the core call above (same disclosed sets, same random scalars) */
        uint64_t dio_w[4] = {dio[0], dio[1], dio[2], dio[3]}, ro_w[4] = {ro[0], ro[1], ro[2], ro[3]};
        uint64_t ho_w[4] = {ho4[0], ho4[1], ho4[2], ho4[3]}, po_w[4] = {po[0], po[1], po[2], po[3]};
        uint8_t pocts[3 * (3 * 48 + 32 * (4 + L))];
        uint64_t poff[4];
        CHECK(bbs_proof_gen_wire_batch(vctx, 3, so, mbytes, mbo, mio, di, dio_w, rnd, ro_w, hdr4, ho_w, ph_bytes, po_w, pocts, poff, sw) == BBS_OK);
        CHECK(sw[0] == 1 && sw[1] == 1 && sw[2] == 1);
        for (int i = 0; i < 3; i++) {                    /* the same proofs, as octets */
            uint8_t want[3 * 48 + 32 * (4 + L)];
            const size_t nc = (size_t)(cmo[i + 1] - cmo[i]), wl = 3 * fpb + 32 * (4 + nc);
            CHECK(bbs_proof_to_octets(curve, pf + (size_t)i * pf_rec, cm + 32 * cmo[i], nc, want) == BBS_OK);
            CHECK(wl == poff[i + 1] - poff[i] && memcmp(want, pocts + poff[i], wl) == 0);
        }
        /* proof_verify: disclosed messages as raw bytes -- items disclose {0, 2}, {1}, {} */
        uint8_t dmb[64];
        uint64_t dmbo[4], dmio[4] = {0, 2, 3, 3};
        const char* dsel[3] = {raw[0], raw[2], raw[L + 1]};
        at = 0; dmbo[0] = 0;
        for (int k = 0; k < 3; k++) { memcpy(dmb + at, dsel[k], strlen(dsel[k])); at += strlen(dsel[k]); dmbo[k + 1] = at; }
        CHECK(bbs_proof_verify_wire_batch(vctx, 3, pocts, poff, dmb, dmbo, dmio, di, dio_w, hdr4, ho_w, ph_bytes, po_w, sw) == BBS_OK);
        CHECK(sw[0] == 1 && sw[1] == 1 && sw[2] == 1);
        pocts[poff[1] + 3 * fpb + 5] ^= 1;               /* e^ of item 1 altered */
        CHECK(bbs_proof_verify_wire_batch(vctx, 3, pocts, poff, dmb, dmbo, dmio, di, dio_w, hdr4, ho_w, ph_bytes, po_w, sw) == BBS_OK);
        CHECK(sw[0] == 1 && sw[1] == 0 && sw[2] == 1);
        pocts[poff[1] + 3 * fpb + 5] ^= 1;
        /* ---- the same three proofs PLUS one of another length through bbs_issuer_*: the library picks the generators by the
         * item's own message count (commitments + disclosed indexes, src/proof_verify.rs:40-43) ---- */
        bbs_issuer* issuer = NULL;
        CHECK(bbs_issuer_create(curve, 0, (const uint8_t*)api_id, alen, &issuer) == BBS_OK && issuer);
        CHECK(bbs_issuer_set_limits(issuer, 16, 5) == BBS_OK);
        CHECK(bbs_issuer_set_public_key(issuer, pk, 0) == BBS_OK);
        /* a fourth item of ONE message: signed and proven through the issuer's own one-message context */
        bbs_ctx* c1 = NULL;
        CHECK(bbs_issuer_context(issuer, 1, &c1) == BBS_OK && c1);
        CHECK(bbs_ctx_set_secret_key(c1, sk) == BBS_OK);
        const uint8_t one_msg[5] = {'h', 'e', 'l', 'l', 'o'};
        const uint64_t one_bo[2] = {0, 5}, one_io[2] = {0, 1}, none_off[2] = {0, 0}, one_dio[2] = {0, 1}, one_ro[2] = {0, 5};
        const uint64_t one_di[1] = {0};
        uint8_t so1[48 + 32], p1[3 * 48 + 32 * 5];
        uint64_t p1off[2];
        int8_t s1 = 0;
        CHECK(bbs_sign_wire_batch(c1, 1, one_msg, one_bo, one_io, NULL, none_off, so1, &s1) == BBS_OK && s1 == 1);
        CHECK(bbs_proof_gen_wire_batch(c1, 1, so1, one_msg, one_bo, one_io, one_di, one_dio, rnd, one_ro, NULL, none_off, NULL, none_off,
                                       p1, p1off, &s1) == BBS_OK && s1 == 1);
        /* the list: items 0 .. 2 (L messages each) and the one-message proof */
        uint8_t all_oct[sizeof(pocts) + sizeof(p1)], all_mb[128], all_hdr[64], all_ph[64];
        uint64_t all_oo[5], all_mbo[8], all_mio[5], all_di[8], all_dio[5], all_ho[5], all_po[5];
        memcpy(all_oct, pocts, poff[3]); memcpy(all_oct + poff[3], p1, p1off[1]);
        for (int i = 0; i <= 3; i++) all_oo[i] = poff[i];
        all_oo[4] = poff[3] + p1off[1];
        memcpy(all_mb, dmb, dmbo[3]); memcpy(all_mb + dmbo[3], one_msg, 5);
        for (int i = 0; i <= 3; i++) all_mbo[i] = dmbo[i];
        all_mbo[4] = dmbo[3] + 5;
        for (int i = 0; i <= 3; i++) all_mio[i] = dmio[i];
        all_mio[4] = dmio[3] + 1;
        for (uint64_t k = 0; k < dio_w[3]; k++) all_di[k] = di[k];
        all_di[dio_w[3]] = 0;
        for (int i = 0; i <= 3; i++) { all_dio[i] = dio_w[i]; all_ho[i] = ho_w[i]; all_po[i] = po_w[i]; }
        all_dio[4] = dio_w[3] + 1; all_ho[4] = ho_w[3]; all_po[4] = po_w[3];
        memcpy(all_hdr, hdr4, ho_w[3]); memcpy(all_ph, ph_bytes, po_w[3]);
        int8_t s4[4];
        CHECK(bbs_issuer_proof_verify(issuer, 4, all_oct, all_oo, all_mb, all_mbo, all_mio, all_di, all_dio, all_hdr, all_ho, all_ph, all_po, s4) == BBS_OK);
        CHECK(s4[0] == 1 && s4[1] == 1 && s4[2] == 1 && s4[3] == 1);
        all_mb[all_mbo[3]] ^= 1;                         /* the one-message item's message altered */
        CHECK(bbs_issuer_proof_verify(issuer, 4, all_oct, all_oo, all_mb, all_mbo, all_mio, all_di, all_dio, all_hdr, all_ho, all_ph, all_po, s4) == BBS_OK);
        CHECK(s4[0] == 1 && s4[1] == 1 && s4[2] == 1 && s4[3] == 0);
        bbs_issuer_destroy(issuer);
    }

    /* kept for run_pool */
    memcpy(saved[curve].gens, gens, (L + 1) * 2 * fpb); memcpy(saved[curve].pk, pk, 4 * fpb); memcpy(saved[curve].pf, pf, 3 * pf_rec);
    memcpy(saved[curve].cm, cm, sizeof cm); memcpy(saved[curve].dm, dm, sizeof dm); memcpy(saved[curve].hdr, hdr4, 15); memcpy(saved[curve].ph, ph_bytes, 2);
    for (int i = 0; i < 4; i++) { saved[curve].cmo[i] = cmo[i]; saved[curve].dmo[i] = dmo[i]; saved[curve].dio[i] = dio3[i]; saved[curve].ho[i] = ho4[i]; saved[curve].po[i] = po[i]; }
    for (int i = 0; i < 3; i++) saved[curve].di[i] = di[i];
    snprintf(saved[curve].api_id, sizeof saved[curve].api_id, "%s", api_id);

    bbs_ctx_destroy(vctx);                                                                                               /* step 18 */
    bbs_ctx_destroy(ctx);
    free(gens); free(pf);
    printf("curve %d: ffi sequence ok\n", curve);
    return 0;
}

/* GpuPool of the shim: a MIXED list (item i of curve i & 1 ... here BLS12-381 at the even positions) over a pool of two members,
 * both on device 0 -- two context sets on one GPU, what a one-GPU box can show of the multi-GPU fan-out behind the ABI. */
static void run_pool(void) {
    int devs[2] = {0, 0};
    bbs_pool* pool = NULL;
    CHECK(bbs_pool_create(devs, 2, &pool) == BBS_OK && pool && bbs_pool_device_count(pool) == 2);                        /* step 19 */
    for (int c = 0; c < 2; c++) {
        CHECK(bbs_pool_set_window_bits(pool, c, 5) == BBS_OK);                                                           /* step 20 */
        CHECK(bbs_pool_set_generators(pool, c, saved[c].gens, L + 1, (const uint8_t*)saved[c].api_id, strlen(saved[c].api_id)) == BBS_OK);
        CHECK(bbs_pool_set_public_key(pool, c, saved[c].pk, 0) == BBS_OK);
    }
    CHECK(bbs_pool_set_inflight(pool, 3) == BBS_OK);
    int qt = 0, qp = 0, qd = 0;
    size_t qs = 0;
    CHECK(bbs_runtime_queue_budget(0, &qt, &qp, &qd, &qs) == BBS_OK && qt >= 0 && qd == (qt > qp ? qt - qp : 0));
    /* the list: BLS12-381 proofs at positions 0, 2, 4, BN254 proofs at 1, 3, 5 */
    uint64_t gi[2][3] = {{0, 2, 4}, {1, 3, 5}};
    int8_t status[6];
    bbs_pv_list lists[2];
    for (int c = 0; c < 2; c++) {
        memset(&lists[c], 0, sizeof lists[c]);
        lists[c].curve = c; lists[c].n = 3; lists[c].proofs_fixed = saved[c].pf;
        lists[c].commitments = saved[c].cm; lists[c].commit_off = saved[c].cmo; lists[c].disclosed_msgs = saved[c].dm; lists[c].dmsg_off = saved[c].dmo;
        lists[c].disclosed_idx = saved[c].di; lists[c].didx_off = saved[c].dio; lists[c].headers = saved[c].hdr; lists[c].hdr_off = saved[c].ho;
        lists[c].ph = saved[c].ph; lists[c].ph_off = saved[c].po; lists[c].global_index = gi[c]; lists[c].status = status;
    }
    memset(status, 99, sizeof status);
    CHECK(bbs_pool_proof_verify(pool, lists, 2, 2) == BBS_OK);          /* shares of 2 + 1 items per member and curve */     /* step 21 */
    for (int i = 0; i < 6; i++) CHECK(status[i] == 1);
    /* one commitment of the BN254 item at list position 3 altered; two lists in flight, the second one clean again */
    uint8_t cm_bad[N * L * 32];
    memcpy(cm_bad, saved[1].cm, sizeof cm_bad);
    cm_bad[32 * saved[1].cmo[1]] ^= 1;
    int8_t st_bad[6], st_ok[6];
    bbs_pv_list bad[2] = {lists[0], lists[1]}, ok[2] = {lists[0], lists[1]};
    bad[1].commitments = cm_bad; bad[0].status = st_bad; bad[1].status = st_bad; ok[0].status = st_ok; ok[1].status = st_ok;
    memset(st_bad, 99, 6); memset(st_ok, 99, 6);
    bbs_pool_job *j1 = NULL, *j2 = NULL;
    CHECK(bbs_pool_proof_verify_submit(pool, bad, 2, 0, &j1) == BBS_OK && j1);                                           /* step 22 */
    CHECK(bbs_pool_proof_verify_submit(pool, ok, 2, 1, &j2) == BBS_OK && j2);
    CHECK(bbs_pool_set_window_bits(pool, 0, 5) == BBS_E_STATE);          /* no reconfiguration while lists are in flight */
    CHECK(bbs_pool_job_wait(j1) == BBS_OK && bbs_pool_job_wait(j2) == BBS_OK);                                           /* step 23 */
    bbs_pool_job_free(j1); bbs_pool_job_free(j2);
    for (int i = 0; i < 6; i++) CHECK(st_bad[i] == (i == 3 ? 0 : 1) && st_ok[i] == 1);
    bbs_ctx* c1 = NULL;
    CHECK(bbs_pool_context(pool, BBS_CURVE_BN254, 1, &c1) == BBS_OK && c1);
    CHECK(bbs_pool_context(pool, BBS_CURVE_BN254, 2, &c1) == BBS_E_ARG);
    bbs_pool_destroy(pool);                                                                                              /* step 24 */
    printf("pool: mixed list over two members ok (queue budget: %d queues, pool %d, kernel frame %zu B per lane)\n", qt, qp, qs);
}

int main(void) {
    printf("%s\n", bbs_version());
    run_curve(BBS_CURVE_BLS12_381);
    run_curve(BBS_CURVE_BN254);
    run_pool();
    printf("all checks passed\n");
    return 0;
}
